"""Batched XarmPickAndPlace on one MI355X: torch tensors in, torch tensors out, every env step is
one call into libxarm_hip.so (include/xarm_hip.h).

Mirrors the reference class `XarmPickAndPlace(gym.GoalEnv)`
(/root/reference/gym_xarm/envs/xarm_pick_and_place.py:16) — same `config` keys (:17-20,68,164,
261,272,276), same spaces (:95-100), same step/reset/compute_reward semantics — widened to E
environments behind the Stable-Baselines3 VecEnv call surface the reference's train script drives
(benchmark/train.py:74-79: make_vec_env -> VecNormalize -> A2C).
"""
import ctypes as C

import numpy as np
import torch

from . import _native
from .spaces import Box, Dict

CONFIG_DEFAULTS = {
    # benchmark configuration, cf. xarm_pick_and_place.py:353-360
    "GUI": False, "num_obj": 1, "reward_type": "sparse", "init_grasp_rate": 0.0,
    "goal_ground_rate": 0.0, "goal_shape": "air",
}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class XarmPickAndPlaceVecEnv:
    """E independent XarmPickAndPlace environments stepped by hand-written HIP kernels."""

    metadata = {"render.modes": ["rgb_array"], "video.frames_per_second": 30}
    ENV_KIND = _native.ENV_PICK_AND_PLACE
    AG_SLICE = slice(8, 11)   # achieved_goal = object position inside `observation` (xarm_pick_and_place.py:228-246)

    def _check_config(self, config):
        cfg = dict(CONFIG_DEFAULTS)
        cfg.update(config or {})
        if cfg["num_obj"] != 1:
            # not runnable in the reference either: step() -> _is_success (xarm_pick_and_place.py:114, :289-291) subtracts
            # self.goal of shape (N, 3) from the flat achieved_goal of shape (3N,), a NumPy broadcast error for N > 1
            raise NotImplementedError("XarmPickAndPlace supports num_obj == 1 only (with num_obj > 1 the reference's own "
                                      "step() raises a broadcast ValueError in _is_success, xarm_pick_and_place.py:289-291)")
        if cfg["reward_type"] not in _native.REWARD_TYPES:
            # 'dense_diff_o2g' / 'incremental' raise in the reference itself (:179, :302-308)
            raise NotImplementedError("reward_type %r" % (cfg["reward_type"],))
        if cfg["goal_shape"] not in _native.GOAL_SHAPES:
            raise NotImplementedError("goal_shape %r" % (cfg["goal_shape"],))
        return cfg

    def _native_config(self):
        return _native.XarmConfig(self.num_envs, self._env_id_offset, self._seed, self.ENV_KIND,
                                  int(self.config["num_obj"]), _native.REWARD_TYPES[self.config["reward_type"]],
                                  _native.GOAL_SHAPES[self.config["goal_shape"]], float(self.config["init_grasp_rate"]),
                                  float(self.config["goal_ground_rate"]), int(self._auto_reset),
                                  self.device.index if self.device.index is not None else torch.cuda.current_device(), 0.0,
                                  int(self._reset_coop_limit), int(self._step_coop_limit), 0)

    def __init__(self, num_envs, config=None, device=None, seed=0, env_id_offset=0, auto_reset=True, reset_coop_limit=0,
                 step_coop_limit=0):
        cfg = self._check_config(config)
        self.config = cfg
        if cfg.get("GUI"):
            raise NotImplementedError("GUI / rendering is outside the HIP hot path (SURVEY.md 2 #20)")
        if not torch.cuda.is_available():
            raise _native.XarmNativeError("gym_xarm_amd needs a HIP device (torch.cuda.is_available() is False); "
                                          "there is no CPU fallback")
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        if self.device.type != "cuda":
            raise _native.XarmNativeError("device must be a cuda (HIP) device, got %s" % self.device)
        self._L = _native.load()
        self.num_envs = int(num_envs)
        self._seed = int(seed)
        self._env_id_offset = int(env_id_offset)
        # auto_reset: True = the reference's VecEnv semantics (reset inside the step call in which the episode ends),
        # False = never, "lazy" = the six reset ticks are spread over the env's next six step calls (include/xarm_hip.h)
        # PickAndPlace: resets of at most this many envs per call run on the cooperative (16 lanes per env) kernel;
        # 0 = the library default, < 0 = always the one-env-per-lane kernel (include/xarm_hip.h)
        self._reset_coop_limit = int(reset_coop_limit)
        # same for the step kernel: batches of at most this many envs step on the cooperative kernel (0 = default)
        self._step_coop_limit = int(step_coop_limit)
        self._lazy = auto_reset == "lazy"
        self._auto_reset = 2 if self._lazy else int(bool(auto_reset))
        self._h = C.c_void_p(0)
        self._create()
        d = _native.XarmDims()
        _native.check(self._L, self._h, self._L.xarm_dims(self._h, C.byref(d)), "xarm_dims")
        self.obs_dim, self.goal_dim, self.act_dim, self.state_dim = d.obs_dim, d.goal_dim, d.act_dim, d.state_dim
        self._max_episode_steps = d.max_episode_steps
        self.n_substeps = d.n_substeps          # internal substeps (Handover: stepSimulation calls) per env step
        self.distance_threshold = 0.05
        E, dev = self.num_envs, self.device
        f32 = torch.float32
        self._obs = torch.zeros(E, self.obs_dim, device=dev, dtype=f32)
        self._ag = torch.zeros(E, self.goal_dim, device=dev, dtype=f32)
        self._dg = torch.zeros(E, self.goal_dim, device=dev, dtype=f32)
        self._rew = torch.zeros(E, device=dev, dtype=f32)
        self._done = torch.zeros(E, device=dev, dtype=torch.uint8)
        self._succ = torch.zeros(E, device=dev, dtype=torch.uint8)
        self._term = torch.zeros(E, self.obs_dim, device=dev, dtype=f32)
        self._actions = None
        # single-env spaces, xarm_pick_and_place.py:95-100
        self.action_space = Box(-1.0, 1.0, shape=(self.act_dim,), dtype=np.float32)
        self.observation_space = Dict(dict(
            desired_goal=Box(-np.inf, np.inf, shape=(self.goal_dim,), dtype=np.float32),
            achieved_goal=Box(-np.inf, np.inf, shape=(self.goal_dim,), dtype=np.float32),
            observation=Box(-np.inf, np.inf, shape=(self.obs_dim,), dtype=np.float32),
        ))

    # ------------------------------------------------------------------ native plumbing
    def _create(self):
        c = self._native_config()
        h = C.c_void_p(0)
        rc = self._L.xarm_create(C.byref(c), C.byref(h))
        _native.check(self._L, None, rc, "xarm_create")
        self._h = h

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _obs_dict(self):
        return {"observation": self._obs, "achieved_goal": self._ag, "desired_goal": self._dg}

    # ------------------------------------------------------------------ gym / VecEnv surface
    def reset(self, mask=None):
        """reset() of every env (or of the envs with mask != 0); returns the dict of [E, .] tensors."""
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
            if m.shape != (self.num_envs,):
                raise ValueError("mask must have shape (%d,)" % self.num_envs)
        rc = self._L.xarm_reset(self._h, _ptr(m), _ptr(self._obs), _ptr(self._ag), _ptr(self._dg), self._stream())
        _native.check(self._L, self._h, rc, "xarm_reset")
        return self._obs_dict()

    def step_async(self, actions):
        a = torch.as_tensor(actions, device=self.device, dtype=torch.float32)
        # same contract as the reference's `assert action.shape == (4,)` (:200), per env
        assert a.shape == (self.num_envs, self.act_dim), "action shape error"
        self._actions = a.contiguous()

    def step_wait(self):
        rc = self._L.xarm_step(self._h, _ptr(self._actions), _ptr(self._obs), _ptr(self._ag), _ptr(self._dg),
                               _ptr(self._rew), _ptr(self._done), _ptr(self._succ), _ptr(self._term), self._stream())
        _native.check(self._L, self._h, rc, "xarm_step")
        if self._lazy:
            phase = self._done
            done = (phase == 1).to(torch.uint8)
            info = {"is_success": self._succ, "terminal_observation": self._obs, "resetting": phase == 2,
                    "TimeLimit.truncated": (phase == 1) & (self._succ == 0)}
            return self._obs_dict(), self._rew, done, info
        info = {"is_success": self._succ, "terminal_observation": self._term,
                "TimeLimit.truncated": (self._done != 0) & (self._succ == 0)}
        self._extra_info(info)
        return self._obs_dict(), self._rew, self._done, info

    def _extra_info(self, info):
        pass

    @property
    def max_episode_steps(self):
        return self._max_episode_steps

    @property
    def action_dim(self):
        return self.act_dim

    def achieved_goal_of(self, observation):
        """achieved_goal carried inside an `observation` row - needed for info['terminal_observation'], which
        holds the last observation of a finished episode but not its goal dict"""
        return observation[..., self.AG_SLICE]

    def episode_steps(self):
        """int32 [E]: steps taken in the current episode of every env"""
        out = torch.empty(self.num_envs, device=self.device, dtype=torch.int32)
        _native.check(self._L, self._h, self._L.xarm_episode_steps(self._h, _ptr(out), self._stream()), "xarm_episode_steps")
        return out

    def set_episode_steps(self, steps):
        """Overwrite the per-env step counter (int [E]) - e.g. to desynchronise the episodes of a freshly reset
        batch so that time-limit resets arrive at their steady-state rate instead of in one burst."""
        s = self.get_state()
        s[:, self.state_dim - 2] = torch.as_tensor(steps, device=self.device).to(torch.float32)
        self.set_state(s)

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def compute_reward(self, achieved_goal, goal, info=None):
        """Batched compute_reward (HER relabelling), xarm_pick_and_place.py:155-177."""
        ag = torch.as_tensor(achieved_goal, device=self.device, dtype=torch.float32).contiguous()
        g = torch.as_tensor(goal, device=self.device, dtype=torch.float32).contiguous()
        assert ag.shape == g.shape and ag.shape[-1] == self.goal_dim
        out = torch.empty(ag.shape[:-1], device=self.device, dtype=torch.float32)
        n = out.numel()
        rc = self._L.xarm_compute_reward(self._h, _ptr(ag), _ptr(g), n, _ptr(out), self._stream())
        _native.check(self._L, self._h, rc, "xarm_compute_reward")
        return out

    def get_state(self):
        s = torch.empty(self.num_envs, self.state_dim, device=self.device, dtype=torch.float32)
        _native.check(self._L, self._h, self._L.xarm_get_state(self._h, _ptr(s), self._stream()), "xarm_get_state")
        return s

    def set_state(self, state):
        s = torch.as_tensor(state, device=self.device, dtype=torch.float32).contiguous()
        assert s.shape == (self.num_envs, self.state_dim)
        _native.check(self._L, self._h, self._L.xarm_set_state(self._h, _ptr(s), self._stream()), "xarm_set_state")
        torch.cuda.current_stream(self.device).synchronize()  # `s` may be a temporary

    def debug_substeps(self, q_target, n):
        """Test hook: n internal substeps toward joint targets [E,9] (include/xarm_hip.h)."""
        qt = torch.as_tensor(q_target, device=self.device, dtype=torch.float32).contiguous()
        assert qt.shape == (self.num_envs, 9)
        _native.check(self._L, self._h, self._L.xarm_debug_substeps(self._h, _ptr(qt), int(n), self._stream()),
                      "xarm_debug_substeps")
        torch.cuda.current_stream(self.device).synchronize()

    def save_state(self, path):
        """Snapshot of the full simulator state (the reference never saves env state).
        File = safetensors with one float32 tensor [E, state_dim] + metadata."""
        from safetensors.torch import save_file
        save_file({"state": self.get_state().cpu()}, path,
                  metadata={"env_kind": str(self.ENV_KIND), "num_envs": str(self.num_envs), "seed": str(self._seed),
                            "env_id_offset": str(self._env_id_offset), "state_dim": str(self.state_dim)})

    def load_state(self, path):
        from safetensors import safe_open
        with safe_open(path, framework="pt") as f:
            meta = f.metadata()
            if int(meta["env_kind"]) != self.ENV_KIND or int(meta["state_dim"]) != self.state_dim or int(meta["num_envs"]) != self.num_envs:
                raise ValueError("snapshot %s does not match this environment (%s)" % (path, meta))
            self.set_state(f.get_tensor("state"))

    @property
    def goal(self):
        return self._dg

    def seed(self, seed=None):
        """Re-key the per-env counter RNG: re-creates the native handle (state is re-initialised)."""
        if seed is not None and int(seed) != self._seed:
            self._seed = int(seed)
            self._L.xarm_destroy(self._h)
            self._create()
        return [self._seed]

    def env_method(self, name, *args, **kwargs):
        return getattr(self, name)(*args, **kwargs)

    def get_attr(self, name, indices=None):
        n = self.num_envs if indices is None else len(indices)
        return [getattr(self, name)] * n

    def set_attr(self, name, value, indices=None):
        setattr(self, name, value)

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self.num_envs

    def render(self, mode="rgb_array", **kw):
        raise NotImplementedError("rendering is outside the HIP hot path (SURVEY.md 2 #20)")

    def timing_enable(self, on=True):
        _native.check(self._L, self._h, self._L.xarm_timing_enable(self._h, int(on)), "xarm_timing_enable")

    def timing_read(self):
        ms, n = C.c_double(0), C.c_int64(0)
        _native.check(self._L, self._h, self._L.xarm_timing_read(self._h, C.byref(ms), C.byref(n)), "xarm_timing_read")
        return ms.value, n.value

    def timing_read_reset(self):
        """(total ms, calls) of the reset kernels launched inside step() since timing_enable"""
        ms, n = C.c_double(0), C.c_int64(0)
        _native.check(self._L, self._h, self._L.xarm_timing_read_reset(self._h, C.byref(ms), C.byref(n)), "xarm_timing_read_reset")
        return ms.value, n.value

    def kernel_limits(self):
        """(reset_coop_limit, step_coop_limit) in force: at most that many envs run on the cooperative kernels"""
        r, s = C.c_int32(0), C.c_int32(0)
        _native.check(self._L, self._h, self._L.xarm_kernel_limits(self._h, C.byref(r), C.byref(s)), "xarm_kernel_limits")
        return r.value, s.value

    def pipeline_info(self):
        """which launches a step() call is made of and the solver constants the library was built with (include/xarm_hip.h
        xarm_pipeline_info): dict(fast_pipeline, reset_overlap, eject_coop_cap, solver_iterations)"""
        v = [C.c_int32(0) for _ in range(4)]
        _native.check(self._L, self._h, self._L.xarm_pipeline_info(self._h, *[C.byref(x) for x in v]), "xarm_pipeline_info")
        return dict(fast_pipeline=bool(v[0].value), reset_overlap=bool(v[1].value), eject_coop_cap=v[2].value, solver_iterations=v[3].value)

    def stage_info(self):
        """the staged step of Handover (one stick) and PickAndPlace (include/xarm_hip.h xarm_stage_info): the first tick / substep of each
        fast stage and, last, their number per step - [0, 5, 10, 15] by default for a pipelined handle, [0, 15] unstaged or small, [0, 0] for
        the other env kinds"""
        n, t = C.c_int32(0), (C.c_int32 * 6)()
        _native.check(self._L, self._h, self._L.xarm_stage_info(self._h, C.byref(n), t), "xarm_stage_info")
        return [int(t[k]) for k in range(n.value + 1)]

    def debug_counts(self):
        """(episodes that ended in the step kernels, envs handed off by the fast kernel) of the last step() call (development hook)"""
        v = [C.c_int32(0) for _ in range(2)]
        _native.check(self._L, self._h, self._L.xarm_debug_counts(self._h, *[C.byref(x) for x in v], self._stream()), "xarm_debug_counts")
        return tuple(x.value for x in v)

    def library(self):
        """(xarm_version(), path of the loaded libxarm_hip.so)"""
        return self._L.xarm_version().decode(), _native.loaded_path()

    def class_keys(self):
        """uint8 [E] (StackTower): the row-set class of every env's last substep - what the step kernel groups the envs by
        (include/xarm_hip.h xarm_class_keys)"""
        out = torch.empty(self.num_envs, device=self.device, dtype=torch.uint8)
        _native.check(self._L, self._h, self._L.xarm_class_keys(self._h, _ptr(out), self._stream()), "xarm_class_keys")
        return out

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.xarm_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


REACH_CONFIG_DEFAULTS = {"reward_type": "sparse", "GUI": False}   # the two keys xarm_reach.py reads (:25,38)


class XarmReachVecEnv(XarmPickAndPlaceVecEnv):
    """E independent XarmReach-v0 environments (/root/reference/gym_xarm/envs/xarm_reach.py:9): contact-free
    reach with the xArm gripper, obs 8 / goal 3 / action 4, 25 steps per episode, rewards sparse / dense /
    dense_diff (:107-116), info['future_length'] (:90)."""

    ENV_KIND = _native.ENV_REACH
    AG_SLICE = slice(0, 3)    # achieved_goal = gripper position (xarm_reach.py:154-161)

    def _check_config(self, config):
        cfg = dict(REACH_CONFIG_DEFAULTS)
        cfg.update(config or {})
        if cfg["reward_type"] not in _native.REACH_REWARD_TYPES:
            raise NotImplementedError("reward_type %r" % (cfg["reward_type"],))
        return cfg

    def _native_config(self):
        return _native.XarmConfig(self.num_envs, self._env_id_offset, self._seed, self.ENV_KIND, 0,
                                  _native.REACH_REWARD_TYPES[self.config["reward_type"]], 0, 0.0, 0.0, int(self._auto_reset),
                                  self.device.index if self.device.index is not None else torch.cuda.current_device(), 0.0,
                                  int(self._reset_coop_limit), int(self._step_coop_limit), 0)

    def _extra_info(self, info):
        # with auto-reset, finished envs already report the fresh episode's counter
        info["future_length"] = self._max_episode_steps - self.episode_steps()
        info["TimeLimit.truncated"] = self._done != 0   # Reach ends only by the step count (:93)

    def debug_substeps(self, q_target, n):
        raise NotImplementedError


# test.py:9-15 of the reference drives XarmHandover-v0 with exactly these keys
HANDOVER_CONFIG_DEFAULTS = {"GUI": False, "num_obj": 1, "same_side_rate": 0.5, "goal_shape": "ground", "use_stand": False}


class XarmHandoverVecEnv(XarmPickAndPlaceVecEnv):
    """E independent XarmHandover-v0 environments (/root/reference/gym_xarm/envs/xarm_handover.py:19): two xArm7 +
    Panda-gripper arms, config['num_obj'] = 1 or 2 sticks, two tables with a gap; obs 13 N + 16 = 29 / 42 (:325-329),
    achieved / desired goal 3 N, action 8 (:118), sparse reward -sum_i [d_i > 0.05] (:177-183) or, for one stick, the
    staged dense reward (:184-199); done = success (every stick within 0.05 of its goal, :395-402) or 100 steps (:138 +
    registry); config['use_stand'] (:391-392): a static stand under every goal.  num_obj = 2 is the reference's
    test.py configuration (test.py:9-15): the second stick / goal are rejection-sampled (:357-360, :375-379)."""

    ENV_KIND = _native.ENV_HANDOVER
    AG_SLICE = slice(0, 3)    # achieved_goal = object position(s), first in the observation (xarm_handover.py:325-336)

    def _check_config(self, config):
        cfg = dict(HANDOVER_CONFIG_DEFAULTS)
        cfg.update(config or {})
        if cfg["num_obj"] not in (1, 2):
            raise NotImplementedError("XarmHandover supports num_obj 1 (BASELINE config 5) or 2 (the reference's test.py)")
        # the reference hard-wires reward_type = 'sparse' (xarm_handover.py:40); its staged 'dense' branch (:184-199) is
        # offered as an opt-in config key (the undefined `d` of its last stage = object-to-goal distance)
        cfg.setdefault("reward_type", "sparse")
        if cfg["reward_type"] not in ("sparse", "dense"):
            raise NotImplementedError("XarmHandover reward_type %r" % (cfg["reward_type"],))
        if cfg["num_obj"] == 2 and cfg["reward_type"] == "dense":
            # not runnable in the reference either: grip_pos_1 (3,) - achieved_goal (6,) is a NumPy broadcast error (:187)
            raise NotImplementedError("XarmHandover reward_type 'dense' with num_obj == 2 raises in the reference itself "
                                      "(xarm_handover.py:187-188: a (3,) grip position minus the (6,) achieved_goal)")
        self.AG_SLICE = slice(0, 3 * cfg["num_obj"])
        return cfg

    def _native_config(self):
        return _native.XarmConfig(self.num_envs, self._env_id_offset, self._seed, self.ENV_KIND, int(self.config["num_obj"]),
                                  _native.REWARD_TYPES[self.config["reward_type"]],
                                  1 if self.config["goal_shape"] == "ground" else 0, 0.0, 0.0, int(self._auto_reset),
                                  self.device.index if self.device.index is not None else torch.cuda.current_device(),
                                  float(self.config["same_side_rate"]), int(self._reset_coop_limit), int(self._step_coop_limit),
                                  int(bool(self.config["use_stand"])))

    def debug_substeps(self, q_target, n):
        raise NotImplementedError


# the reference's XarmStackTowerEnv takes no config (xarm_stack_tower.py:14); reward_type is an attribute (:28)
STACK_CONFIG_DEFAULTS = {"GUI": False, "num_obj": 3, "reward_type": "sparse"}


class XarmStackTowerVecEnv(XarmPickAndPlaceVecEnv):
    """E independent XarmPDStackTower-v0 environments (/root/reference/gym_xarm/envs/xarm_stack_tower.py:13): two
    xArm7 + Panda-gripper arms and three 5 cm cubes on one table; obs 55 (:190-199), action 8 (:86), goal 9 =
    tower positions (:212-219), reward -(|ag - g| > 0.09) or -d (:124-129), 50 steps (:43).  step() of the reference
    never sets done (:111); the VecEnv reports the 50-step limit as done with TimeLimit.truncated."""

    ENV_KIND = _native.ENV_STACK_TOWER
    AG_SLICE = slice(0, 9)    # achieved_goal = the three cube positions, first in the observation (:190)

    def _check_config(self, config):
        cfg = dict(STACK_CONFIG_DEFAULTS)
        cfg.update(config or {})
        if cfg["num_obj"] != 3:
            raise NotImplementedError("XarmStackTower has num_obj == 3 (xarm_stack_tower.py:19)")
        if cfg["reward_type"] not in ("sparse", "dense"):
            raise NotImplementedError("reward_type %r" % (cfg["reward_type"],))
        return cfg

    def _native_config(self):
        return _native.XarmConfig(self.num_envs, self._env_id_offset, self._seed, self.ENV_KIND, 3,
                                  0 if self.config["reward_type"] == "sparse" else 1, 0, 0.0, 0.0, int(self._auto_reset),
                                  self.device.index if self.device.index is not None else torch.cuda.current_device(), 0.0, 0)

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.distance_threshold = 0.03 * 3   # :20-21

    def _extra_info(self, info):
        info["TimeLimit.truncated"] = self._done != 0   # only the step limit ends an episode

    def debug_substeps(self, q_target, n):
        raise NotImplementedError

