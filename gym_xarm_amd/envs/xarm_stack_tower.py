"""Single-environment view with the reference's call surface (/root/reference/gym_xarm/envs/xarm_stack_tower.py:13):
`XarmStackTowerEnv()` takes no config, numpy in/out, step() never sets done (:111).  A 1-env XarmStackTowerVecEnv
(HIP kernels, two lanes = two arms) sits behind it."""
import numpy as np
import torch

from ..vec_env import XarmStackTowerVecEnv


class XarmStackTowerEnv:
    def __init__(self, config=None, device=None, seed=0):
        self._vec = XarmStackTowerVecEnv(1, config=config, device=device, seed=seed, auto_reset=False)
        self.action_space = self._vec.action_space
        self.observation_space = self._vec.observation_space
        self._max_episode_steps = self._vec._max_episode_steps
        self.num_obj = 3
        self.distance_threshold = self._vec.distance_threshold
        self.reward_type = self._vec.config["reward_type"]
        self.goal = None

    def _np_obs(self, d):
        return {k: v[0].detach().cpu().numpy().copy() for k, v in d.items()}

    def reset(self):
        obs = self._np_obs(self._vec.reset())
        self.goal = obs["desired_goal"].copy()
        return obs

    def step(self, action):
        action = np.asarray(action, dtype=np.float32)
        assert action.shape == (8,), 'action shape error'
        obs, rew, done, info = self._vec.step(torch.from_numpy(action)[None])
        # `done = False` in the reference (:111); the registry's TimeLimit is what ends an episode
        return self._np_obs(obs), float(rew[0].item()), False, {"is_success": float(info["is_success"][0].item())}

    def compute_reward(self, achieved_goal, goal, info=None):
        ag = np.asarray(achieved_goal, dtype=np.float32)
        out = self._vec.compute_reward(ag, np.asarray(goal, dtype=np.float32)).cpu().numpy()
        return out if ag.ndim > 1 else float(out)

    def seed(self, seed=None):
        return self._vec.seed(seed)

    def close(self):
        self._vec.close()
