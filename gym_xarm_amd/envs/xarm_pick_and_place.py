"""Single-environment view with the reference's exact call surface
(/root/reference/gym_xarm/envs/xarm_pick_and_place.py:16): `XarmPickAndPlace(config)`,
`step(action) -> (obs dict of numpy arrays, reward, done, info)`, `reset()`, `compute_reward`,
`seed`, `close`.  It is a thin numpy shell around a 1-env XarmPickAndPlaceVecEnv, i.e. it runs
the same HIP kernels — there is no PyBullet and no CPU path behind it."""
import numpy as np
import torch

from ..vec_env import XarmPickAndPlaceVecEnv


class XarmPickAndPlace:
    def __init__(self, config=None, device=None, seed=0):
        self._vec = XarmPickAndPlaceVecEnv(1, config=config, device=device, seed=seed, auto_reset=False)
        self.config = self._vec.config
        self.action_space = self._vec.action_space
        self.observation_space = self._vec.observation_space
        self._max_episode_steps = self._vec._max_episode_steps
        self.distance_threshold = self._vec.distance_threshold
        self.metadata = self._vec.metadata
        self.num_steps = 0
        self.goal = None

    def _np_obs(self, d):
        return {k: v[0].detach().cpu().numpy().copy() for k, v in d.items()}

    def reset(self):
        self.num_steps = 0
        obs = self._np_obs(self._vec.reset())
        self.goal = obs["desired_goal"].reshape(1, 3).copy()
        return obs

    def step(self, action):
        action = np.asarray(action, dtype=np.float32)
        assert action.shape == (4,), 'action shape error'
        self.num_steps += 1
        obs, rew, done, info = self._vec.step(torch.from_numpy(action)[None])
        o = self._np_obs(obs)
        return o, float(rew[0].item()), bool(done[0].item()), {
            "is_success": np.array([float(info["is_success"][0].item())], dtype=np.float32)}

    def compute_reward(self, achieved_goal, goal, info=None):
        ag = np.asarray(achieved_goal, dtype=np.float32)
        out = self._vec.compute_reward(ag, np.asarray(goal, dtype=np.float32)).cpu().numpy()
        return out if ag.ndim > 1 else np.float32(out)

    def seed(self, seed=None):
        return self._vec.seed(seed)

    def render(self, mode="rgb_array", **kw):
        return self._vec.render(mode)

    def close(self):
        self._vec.close()
