"""Single-environment view with the reference's call surface (/root/reference/gym_xarm/envs/xarm_handover.py:19):
`XarmHandover(config)` with config keys GUI / num_obj (1 or 2) / same_side_rate / goal_shape / use_stand (test.py:9-15),
numpy in/out.  A 1-env XarmHandoverVecEnv (HIP kernels, two lanes = two arms) sits behind it."""
import numpy as np
import torch

from ..vec_env import XarmHandoverVecEnv


class XarmHandover:
    def __init__(self, config=None, device=None, seed=0):
        self._vec = XarmHandoverVecEnv(1, config=config, device=device, seed=seed, auto_reset=False)
        self.config = self._vec.config
        self.action_space = self._vec.action_space
        self.observation_space = self._vec.observation_space
        self._max_episode_steps = self._vec._max_episode_steps
        self.distance_threshold = self._vec.distance_threshold
        self.reward_type = "sparse"
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
        return self._np_obs(obs), float(rew[0].item()), bool(done[0].item()), {"is_success": float(info["is_success"][0].item())}

    def compute_reward(self, achieved_goal, goal, info=None):
        ag = np.asarray(achieved_goal, dtype=np.float32)
        out = self._vec.compute_reward(ag, np.asarray(goal, dtype=np.float32)).cpu().numpy()
        return float(out.reshape(-1)[0]) if out.size == 1 else out      # `if len(rew) == 1: return rew[0]` (:180-183)

    def seed(self, seed=None):
        return self._vec.seed(seed)

    def close(self):
        self._vec.close()
