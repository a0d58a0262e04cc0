"""Single-environment view with the reference's call surface (/root/reference/gym_xarm/envs/xarm_reach.py:9):
`XarmReachEnv(config)` with config keys 'reward_type' and 'GUI', numpy in/out, info = {'is_success',
'future_length'}.  A 1-env XarmReachVecEnv (HIP kernels) sits behind it."""
import numpy as np
import torch

from ..vec_env import XarmReachVecEnv


class XarmReachEnv:
    def __init__(self, config=None, device=None, seed=0):
        self._vec = XarmReachVecEnv(1, config=config, device=device, seed=seed, auto_reset=False)
        self.reward_type = self._vec.config["reward_type"]
        self.action_space = self._vec.action_space
        self.observation_space = self._vec.observation_space
        self._max_episode_steps = self._vec._max_episode_steps
        self.distance_threshold = self._vec.distance_threshold
        self.num_steps = 0
        self.goal = None

    def _np_obs(self, d):
        return {k: v[0].detach().cpu().numpy().copy() for k, v in d.items()}

    def reset(self):
        self.num_steps = 0
        obs = self._np_obs(self._vec.reset())
        self.goal = obs["desired_goal"].copy()
        return obs

    def step(self, action):
        action = np.asarray(action, dtype=np.float32)
        assert action.shape == (4,), 'action shape error'
        self.num_steps += 1
        obs, rew, done, info = self._vec.step(torch.from_numpy(action)[None])
        return self._np_obs(obs), float(rew[0].item()), bool(done[0].item()), {
            "is_success": np.float32(info["is_success"][0].item()),
            "future_length": int(info["future_length"][0].item())}

    def compute_reward(self, achieved_goal, goal, info=None):
        ag = np.asarray(achieved_goal, dtype=np.float32)
        out = self._vec.compute_reward(ag, np.asarray(goal, dtype=np.float32)).cpu().numpy()
        return out if ag.ndim > 1 else np.float32(out)

    def seed(self, seed=None):
        return self._vec.seed(seed)

    def render(self):
        raise NotImplementedError("rendering is outside the HIP hot path")

    def close(self):
        self._vec.close()
