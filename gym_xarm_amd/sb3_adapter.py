"""Stable-Baselines3-shaped view of the batched envs: the surface `benchmark/train.py:74-79` drives
(`make_vec_env(env_id, n_envs, seed, vec_env_cls=DummyVecEnv, monitor_dir)` -> `VecNormalize` -> `A2C`).

SB3's VecEnv contract (stable_baselines3.common.vec_env.base_vec_env.VecEnv, SB3 1.x - the era of the reference):
numpy arrays in and out; `reset() -> obs`; `step_async(actions)` / `step_wait() -> (obs, rewards, dones, infos)` with
`obs` a dict of `[n_envs, dim]` arrays for Dict spaces, `rewards` float32 `[n_envs]`, `dones` bool `[n_envs]` and `infos` a
LIST of per-env dicts; an env that finishes is reset inside the call, its row of `obs` is the first observation of the next
episode and `infos[i]['terminal_observation']` holds the last one (for Dict spaces: the dict), `infos[i]['TimeLimit.truncated']`
says whether the time limit, not the task, ended it; `env_method / get_attr / set_attr / seed / env_is_wrapped / close`.

PARITY UNPINNED for this surface: stable_baselines3 is not importable in the build image and cannot be fetched, so the class
below is written to the documented interface and exercised by a stub consumer that stores references across steps
(tests/test_sb3_adapter.py) - it has never met the real `A2C.learn`.  Every array handed out is a fresh host copy: the
torch VecEnv underneath reuses its device buffers call after call (gym_xarm_amd/vec_env.py), a learner that keeps
references (rollout buffers do) must not see them change.

Also here: `VecExtractDictObs` (the reference's own wrapper, benchmark/train.py:49-63) and the flat-observation 'NoGoal'
variant the reference trains on (`XarmPDHandoverNoGoal-v1`, benchmark/train.py:67, README.md:38: observation 29 wide - the
shape its saved vec_normalize.pkl documents - with the staged dense reward of train.py:66)."""
import numpy as np
import torch

from .spaces import Box


def _np(t):
    return t.detach().cpu().numpy().copy()


class SB3VecEnv:
    """numpy / list-of-dict adapter around a gym_xarm_amd torch VecEnv (dict observations)."""

    def __init__(self, venv):
        self.venv = venv
        self.num_envs = venv.num_envs
        self.observation_space = venv.observation_space
        self.action_space = venv.action_space
        self.metadata = getattr(venv, "metadata", {})
        self._last_dg = None      # desired goals of the running episodes: the goal half of a terminal observation
        self._actions = None
        self.reset_infos = [{} for _ in range(self.num_envs)]

    # ------------------------------------------------------------------ VecEnv API
    def reset(self):
        obs = {k: _np(v) for k, v in self.venv.reset().items()}
        self._last_dg = obs["desired_goal"].copy()
        return obs

    def step_async(self, actions):
        a = np.asarray(actions, dtype=np.float32)
        assert a.shape == (self.num_envs,) + self.action_space.shape, "action shape error"
        self._actions = torch.from_numpy(np.ascontiguousarray(a))

    def step_wait(self):
        obs_t, rew_t, done_t, info_t = self.venv.step(self._actions)
        obs = {k: _np(v) for k, v in obs_t.items()}
        rew = _np(rew_t).astype(np.float32)
        done = _np(done_t).astype(bool)
        succ = _np(info_t["is_success"]).astype(np.float32)
        trunc = _np(info_t["TimeLimit.truncated"]).astype(bool)
        # 65 536 dicts per call: plain Python floats from one tolist() per key (a numpy scalar conversion per env is ~5x slower)
        infos = [{"is_success": x} for x in succ.tolist()]
        extra = {k: _np(v) for k, v in info_t.items() if k not in ("is_success", "TimeLimit.truncated", "terminal_observation") and torch.is_tensor(v)}
        for k, v in extra.items():
            if v.ndim == 1:
                for d, x in zip(infos, v.tolist()):
                    d[k] = x
            else:
                for d, x in zip(infos, v):
                    d[k] = x
        idx = np.nonzero(done)[0]
        if idx.size:
            term = _np(info_t["terminal_observation"][torch.from_numpy(idx).to(info_t["terminal_observation"].device)])
            ag = term[:, self.venv.AG_SLICE]
            for j, i in enumerate(idx):
                infos[i]["terminal_observation"] = {"observation": term[j], "achieved_goal": ag[j].copy(),
                                                    "desired_goal": self._last_dg[i].copy()}
                infos[i]["TimeLimit.truncated"] = bool(trunc[i])
        self._last_dg = obs["desired_goal"].copy()
        return obs, rew, done, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.venv.close()

    def seed(self, seed=None):
        s = self.venv.seed(seed)[0]
        return [s + i if s is not None else None for i in range(self.num_envs)]

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        return [indices] if isinstance(indices, int) else list(indices)

    def get_attr(self, attr_name, indices=None):
        return [getattr(self.venv, attr_name)] * len(self._indices(indices))

    def set_attr(self, attr_name, value, indices=None):
        setattr(self.venv, attr_name, value)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        """one call on the batched env, its result repeated per requested index - e.g. SB3's HerReplayBuffer asks
        env_method('compute_reward', achieved, desired, infos, indices=[0]) and takes element 0"""
        args = [a for a in method_args]
        if method_name == "compute_reward":
            ag, g = np.asarray(args[0], dtype=np.float32), np.asarray(args[1], dtype=np.float32)
            out = _np(self.venv.compute_reward(torch.from_numpy(ag), torch.from_numpy(g))).astype(np.float32)
            res = out if ag.ndim > 1 else float(out)
        else:
            res = getattr(self.venv, method_name)(*args, **method_kwargs)
        return [res] * len(self._indices(indices))

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * len(self._indices(indices))

    def get_images(self):
        raise NotImplementedError("rendering is outside the HIP hot path (SURVEY.md 2 #20)")

    def render(self, mode="human"):
        raise NotImplementedError("rendering is outside the HIP hot path (SURVEY.md 2 #20)")

    @property
    def unwrapped(self):
        return self


class VecExtractDictObs:
    """benchmark/train.py:49-63, same name and semantics: present one key of the dict observation as the observation."""

    def __init__(self, venv, key):
        self.venv, self.key = venv, key
        self.num_envs = venv.num_envs
        self.observation_space = venv.observation_space.spaces[key]
        self.action_space = venv.action_space

    def reset(self):
        return self.venv.reset()[self.key]

    def step_async(self, actions):
        self.venv.step_async(actions)

    def step_wait(self):
        obs, reward, done, info = self.venv.step_wait()
        if isinstance(info, list):      # SB3 shape: a flat terminal observation for a flat observation space
            for d in info:
                if "terminal_observation" in d and isinstance(d["terminal_observation"], dict):
                    d["terminal_observation"] = d["terminal_observation"][self.key]
        return obs[self.key], reward, done, info

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def __getattr__(self, name):
        return getattr(self.venv, name)


def make_vec_env(env_id, n_envs=1, seed=None, vec_env_cls=None, monitor_dir=None, env_kwargs=None, **unused):
    """counterpart of stable_baselines3.common.env_util.make_vec_env as benchmark/train.py:74 calls it: ONE batched HIP env
    of n_envs instead of n_envs Python envs under DummyVecEnv (`vec_env_cls` is accepted and ignored).  monitor_dir is not
    served here - the on-device driver writes the Monitor CSV (gym_xarm_amd/train.py EpisodeMonitor)."""
    import gym_xarm_amd
    kw = dict(env_kwargs or {})
    venv = gym_xarm_amd.make(env_id, num_envs=n_envs, seed=0 if seed is None else seed, **kw)
    if getattr(venv, "flat_observation", False):
        return VecExtractDictObs(SB3VecEnv(venv.env), "observation")
    return SB3VecEnv(venv)


class FlatObsVecEnv:
    """torch-side 'NoGoal' view for the on-device driver (gym_xarm_amd/train.py): `observation` alone, as a tensor [E, obs_dim],
    with a Box observation space - what `XarmPDHandoverNoGoal-v1` (benchmark/train.py:67) presents to A2C's MlpPolicy.
    Tensors are the env's persistent buffers (no copy; the driver consumes them before the next step)."""

    flat_observation = True

    def __init__(self, env):
        self.env = env
        self.num_envs, self.device = env.num_envs, env.device
        self.obs_dim, self.goal_dim, self.act_dim = env.obs_dim, 0, env.act_dim
        self.observation_space = Box(-np.inf, np.inf, shape=(env.obs_dim,), dtype=np.float32)
        self.action_space = env.action_space
        self._max_episode_steps = env._max_episode_steps

    def reset(self, mask=None):
        return self.env.reset(mask)["observation"]

    def step(self, actions):
        obs, rew, done, info = self.env.step(actions)
        return obs["observation"], rew, done, info

    def step_async(self, actions):
        self.env.step_async(actions)

    def step_wait(self):
        obs, rew, done, info = self.env.step_wait()
        return obs["observation"], rew, done, info

    def __getattr__(self, name):
        return getattr(self.env, name)


# benchmark/train.py:66-67: reward_type "dense" on the NoGoal id; the remaining keys are test.py's with one stick
NOGOAL_HANDOVER_CONFIG = {"GUI": False, "num_obj": 1, "same_side_rate": 0.5, "goal_shape": "ground", "use_stand": False,
                          "reward_type": "dense"}


def make_handover_nogoal_vec(num_envs, config=None, **kwargs):
    from .vec_env import XarmHandoverVecEnv
    cfg = dict(NOGOAL_HANDOVER_CONFIG)
    cfg.update(config or {})
    return FlatObsVecEnv(XarmHandoverVecEnv(num_envs, config=cfg, **kwargs))


class XarmHandoverNoGoal:
    """single-env numpy view of the NoGoal id: flat 29-wide observation, staged dense reward"""

    def __init__(self, config=None, device=None, seed=0):
        from .envs.xarm_handover import XarmHandover
        cfg = dict(NOGOAL_HANDOVER_CONFIG)
        cfg.update(config or {})
        self._env = XarmHandover(cfg, device=device, seed=seed)
        self.action_space = self._env.action_space
        self.observation_space = self._env.observation_space.spaces["observation"]
        self._max_episode_steps = self._env._max_episode_steps
        self.reward_type = cfg["reward_type"]

    def reset(self):
        return self._env.reset()["observation"]

    def step(self, action):
        obs, r, d, info = self._env.step(action)
        return obs["observation"], r, d, info

    def seed(self, seed=None):
        return self._env.seed(seed)

    def close(self):
        self._env.close()
