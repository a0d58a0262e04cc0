"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module;
the product package gym_xarm_amd never does (tests/test_boundary.py greps for it).
The model comes from gym_xarm_amd/model/xarm7_pd.json at run time, i.e. through a different
path than the constexpr header the HIP kernels are compiled with.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
MODEL_JSON = os.path.join(ROOT, "gym_xarm_amd", "model", "xarm7_pd.json")
LIB_PATH = os.path.join(HERE, "libxarm_oracle.so")

MAXL, MAXD, NPAD = 16, 13, 2
STATE_DIM, OBS_DIM, GOAL_DIM, ACT_DIM = 54, 24, 3, 4

_d = C.c_double
_i = C.c_int32


class XoModel(C.Structure):
    _fields_ = [
        ("n_links", _i), ("parent", _i * MAXL), ("jtype", _i * MAXL),
        ("eef_link", _i), ("hand_link", _i), ("finger_link", _i * 2),
        ("org_p", (_d * 3) * MAXL), ("org_rpy", (_d * 3) * MAXL), ("axis", (_d * 3) * MAXL),
        ("lower", _d * MAXL), ("upper", _d * MAXL), ("damping", _d * MAXL),
        ("mass", _d * MAXL), ("com", (_d * 3) * MAXL), ("inertia", (_d * 6) * MAXL),
        ("pad_radius", _d), ("pad_center_left", (_d * 3) * NPAD),
        ("gravity", _d), ("contact_erp", _d), ("contact_margin", _d), ("solver_margin", _d), ("warmstart", _d),
        ("motor_kp", _d), ("motor_kd", _d), ("arm_motor_force", _d),
        ("gear_erp", _d), ("gear_max_force", _d), ("global_erp", _d),
        ("finger_contact_stiffness", _d), ("finger_contact_damping", _d), ("object_contact_damping", _d),
        ("lin_damping", _d), ("ang_damping", _d), ("ik_lambda", _d), ("ik_residual", _d),
        ("ik_max_dtheta", _d), ("limit_window", _d),
        ("mu_object", _d), ("mu_table", _d), ("mu_finger", _d), ("mu_finger_grasp", _d),
        ("num_iterations", _i), ("_pad0", _i),
        ("table_half_x", _d), ("table_half_y", _d), ("table_top_z", _d),
        ("time_step", _d), ("action_dt", _d), ("max_vel", _d), ("max_gripper_vel", _d),
        ("pos_low", _d * 3), ("pos_high", _d * 3), ("goal_low", _d * 3), ("goal_high", _d * 3),
        ("obj_low", _d * 2), ("obj_high", _d * 2),
        ("gripper_low", _d), ("gripper_high", _d), ("height_offset", _d),
        ("start_gripper_pos", _d * 3), ("reset_finger_target", _d),
        ("finger_motor_force", _d), ("distance_threshold", _d),
        ("obj_half", _d * 3), ("obj_mass", _d), ("eef2grip", _d * 3),
        ("n_substeps", _i), ("reset_ticks", _i), ("max_episode_steps", _i), ("_pad1", _i),
    ]


class XoReachCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("env_id_offset", C.c_int64), ("reward_type", _i), ("driver_link", _i),
                ("time_step", _d), ("action_dt", _d), ("max_vel", _d), ("max_gripper_vel", _d),
                ("pos_low", _d * 3), ("pos_high", _d * 3), ("goal_low", _d * 3), ("goal_high", _d * 3),
                ("motor_force", _d), ("distance_threshold", _d), ("joint_init_pos", _d * MAXD),
                ("n_substeps", _i), ("max_episode_steps", _i)]


REACH_JSON = os.path.join(ROOT, "gym_xarm_amd", "model", "xarm7_reach.json")
REACH_REWARD_TYPES = {"sparse": 0, "dense": 1, "dense_diff": 2}
REACH_STATE_DIM, REACH_OBS_DIM = 45, 8


class XoHoCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("env_id_offset", C.c_int64), ("same_side_rate", _d), ("goal_shape", _i), ("n_ticks", _i),
                ("time_step", _d), ("action_dt", _d), ("max_vel", _d), ("max_gripper_vel", _d),
                ("pos_low", (_d * 3) * 2), ("pos_high", (_d * 3) * 2), ("goal_low", _d * 3), ("goal_high", _d * 3),
                ("obj_low", _d * 2), ("obj_high", _d * 2), ("gripper_low", _d), ("gripper_high", _d), ("height_offset", _d),
                ("eff_init_pos", (_d * 3) * 2), ("joint_init_pos", _d * 9), ("base_pos", (_d * 3) * 2), ("base_yaw", _d * 2),
                ("finger_motor_force", _d), ("distance_threshold", _d), ("obj_half", _d * 3), ("eef2grip", _d * 3),
                ("table_x_min", _d), ("table_x_max", _d), ("table_half_y", _d), ("ground_z", _d),
                ("reset_ticks", _i), ("max_episode_steps", _i), ("reward_type", _i), ("use_stand", _i),
                ("stand_half", _d * 3), ("stand_below_goal", _d),
                ("spawn_min_dy", _d), ("goal_min_dy", _d), ("goal_min_obj_dist", _d), ("sample_max_tries", _i), ("_pad2", _i)]


HO_STATE_DIM, HO_OBS_DIM, HO_ACT_DIM = 76, 29, 8
HO2_STATE_DIM, HO2_OBS_DIM, HO2_GOAL_DIM = 100, 42, 6


class XoStCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("env_id_offset", C.c_int64), ("reward_type", _i), ("n_substeps", _i),
                ("max_episode_steps", _i), ("reserved", _i),
                ("time_step", _d), ("action_dt", _d), ("max_vel", _d), ("max_gripper_vel", _d),
                ("pos_low", (_d * 3) * 2), ("pos_high", (_d * 3) * 2), ("goal_low", _d * 2), ("goal_high", _d * 2),
                ("obj_low", _d * 2), ("obj_high", _d * 2), ("gripper_low", _d), ("gripper_high", _d), ("height_offset", _d),
                ("joint_init_pos", _d * 9), ("base_pos", (_d * 3) * 2), ("base_yaw", _d * 2),
                ("finger_motor_force", _d), ("distance_threshold", _d), ("cube_half", _d), ("cube_mass", _d)]


ST_STATE_DIM, ST_OBS_DIM, ST_ACT_DIM, ST_GOAL_DIM = 136, 55, 8, 9
# state offsets (oracle/xarm_oracle_stack.inc.c)
ST_Q, ST_QD, ST_QT, ST_BP, ST_BQ, ST_BV, ST_BW, ST_GOAL, ST_LT, ST_LP, ST_STEPS, ST_EPISODE = 0, 18, 36, 54, 63, 75, 84, 93, 102, 126, 134, 135


class XoPnpCfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("env_id_offset", C.c_int64), ("init_grasp_rate", _d),
                ("goal_ground_rate", _d), ("goal_shape", _i), ("reward_type", _i)]


JTYPE = {"fixed": 0, "revolute": 1, "prismatic": 2}
REWARD_TYPES = {"sparse": 0, "dense_o2g": 1, "dense": 2}
GOAL_SHAPES = {"air": 0, "ground": 1}


def load_model_json(path=MODEL_JSON):
    with open(path) as f:
        return json.load(f)


def build_model(js=None):
    js = js or load_model_json()
    m = XoModel()
    links = js["links"]
    m.n_links = len(links)
    for i, l in enumerate(links):
        m.parent[i] = l["parent"]
        m.jtype[i] = JTYPE[l["joint"]]
        for k in range(3):
            m.org_p[i][k] = l["origin_xyz"][k]
            m.org_rpy[i][k] = l["origin_rpy"][k]
            m.axis[i][k] = l["axis"][k]
            m.com[i][k] = l["com"][k]
        for k in range(6):
            m.inertia[i][k] = l["inertia"][k]
        m.lower[i], m.upper[i], m.damping[i], m.mass[i] = l["lower"], l["upper"], l["damping"], l["mass"]
    m.eef_link, m.hand_link = js["eef_link"], js["hand_link"]
    if "finger_links" in js:
        m.finger_link[0], m.finger_link[1] = js["finger_links"]
        m.pad_radius = js["pads"]["radius"]
        for j in range(NPAD):
            for k in range(3):
                m.pad_center_left[j][k] = js["pads"]["centers_left"][j][k]
    for k, v in js["solver"].items():
        if not k.startswith("_"):
            setattr(m, k, v)
    if "pick_and_place" not in js:
        return m
    m.table_half_x, m.table_half_y, m.table_top_z = js["table"]["half_x"], js["table"]["half_y"], js["table"]["top_z"]
    p = js["pick_and_place"]
    for name in ("time_step", "action_dt", "max_vel", "max_gripper_vel", "gripper_low", "gripper_high",
                 "height_offset", "reset_finger_target", "finger_motor_force", "distance_threshold",
                 "obj_mass", "n_substeps", "reset_ticks", "max_episode_steps"):
        setattr(m, name, p[name])
    for name, n in (("pos_low", 3), ("pos_high", 3), ("goal_low", 3), ("goal_high", 3), ("obj_low", 2),
                    ("obj_high", 2), ("start_gripper_pos", 3), ("obj_half", 3), ("eef2grip", 3)):
        arr = getattr(m, name)
        for k in range(n):
            arr[k] = p[name][k]
    return m


def build_lib(force=False):
    src = os.path.join(HERE, "xarm_oracle.c")
    hdr = os.path.join(HERE, "xarm_oracle.h")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", HERE, "-s", "-B", "libxarm_oracle.so"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build_lib()
        L = C.CDLL(LIB_PATH)
        dp, u8p = C.POINTER(_d), C.POINTER(C.c_uint8)
        mp, cp = C.POINTER(XoModel), C.POINTER(XoPnpCfg)
        L.xo_pnp_init.argtypes = [mp, cp, C.c_int64, dp]
        L.xo_pnp_reset.argtypes = [mp, cp, C.c_int64, dp, u8p, dp, dp, dp]
        L.xo_pnp_step.argtypes = [mp, cp, C.c_int64, dp, dp, dp, dp, dp, dp, u8p, u8p]
        L.xo_pnp_compute_reward.argtypes = [mp, _i, C.c_int64, dp, dp, dp]
        L.xo_pnp_substeps.argtypes = [mp, C.c_int64, dp, dp, C.c_int32]
        L.xo_pnp_dense_reward.argtypes = [mp, _i, dp, dp, dp]
        L.xo_pnp_dense_reward.restype = _d
        hp = C.POINTER(XoHoCfg)
        L.xo_ho_init.argtypes = [mp, hp, C.c_int64, dp]
        L.xo_ho_reset.argtypes = [mp, hp, C.c_int64, dp, u8p, dp, dp, dp]
        L.xo_ho_step.argtypes = [mp, hp, C.c_int64, dp, dp, dp, dp, dp, dp, u8p, u8p]
        L.xo_ho_compute_reward.argtypes = [hp, C.c_int64, dp, dp, dp]
        L.xo_ho2_init.argtypes = [mp, hp, C.c_int64, dp]
        L.xo_ho2_reset.argtypes = [mp, hp, C.c_int64, dp, u8p, dp, dp, dp]
        L.xo_ho2_step.argtypes = [mp, hp, C.c_int64, dp, dp, dp, dp, dp, dp, u8p, u8p]
        L.xo_ho2_compute_reward.argtypes = [hp, C.c_int64, dp, dp, dp]
        sp = C.POINTER(XoStCfg)
        L.xo_st_init.argtypes = [mp, sp, C.c_int64, dp]
        L.xo_st_reset.argtypes = [mp, sp, C.c_int64, dp, u8p, dp, dp, dp]
        L.xo_st_step.argtypes = [mp, sp, C.c_int64, dp, dp, dp, dp, dp, dp, u8p, u8p]
        L.xo_st_compute_reward.argtypes = [sp, _i, C.c_int64, dp, dp, dp]
        rp = C.POINTER(XoReachCfg)
        L.xo_reach_init.argtypes = [mp, rp, C.c_int64, dp]
        L.xo_reach_reset.argtypes = [mp, rp, C.c_int64, dp, u8p, dp, dp, dp]
        L.xo_reach_step.argtypes = [mp, rp, C.c_int64, dp, dp, dp, dp, dp, dp, u8p, u8p, C.POINTER(C.c_int32)]
        L.xo_reach_compute_reward.argtypes = [rp, _i, C.c_int64, dp, dp, dp]
        L.xo_fk.argtypes = [mp, dp, dp, dp]
        L.xo_ik.argtypes = [mp, dp, dp, _i, dp]
        L.xo_mass_matrix_inv.argtypes = [mp, dp, dp]
        L.xo_forward_dynamics.argtypes = [mp, dp, dp, dp, dp]
        L.xo_philox.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        L.xo_philox.restype = None
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(_d))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


class OraclePnP:
    """Batched CPU PickAndPlace (num_obj = 1), float64, same call surface as the C-ABI."""

    def __init__(self, num_envs, seed=0, env_id_offset=0, init_grasp_rate=0.0, goal_ground_rate=0.0,
                 goal_shape="air", reward_type="sparse", model_json=None):
        self.L = lib()
        self.m = build_model(model_json)
        self.cfg = XoPnpCfg(seed, env_id_offset, init_grasp_rate, goal_ground_rate,
                            GOAL_SHAPES[goal_shape], REWARD_TYPES[reward_type])
        self.E = int(num_envs)
        self.state = np.zeros((self.E, STATE_DIM))
        self.L.xo_pnp_init(self.m, self.cfg, self.E, _p(self.state))

    def _bufs(self):
        return (np.zeros((self.E, OBS_DIM)), np.zeros((self.E, GOAL_DIM)), np.zeros((self.E, GOAL_DIM)))

    def reset(self, mask=None):
        obs, ag, dg = self._bufs()
        mk = None if mask is None else _u8(np.ascontiguousarray(mask, dtype=np.uint8))
        self.L.xo_pnp_reset(self.m, self.cfg, self.E, _p(self.state), mk, _p(obs), _p(ag), _p(dg))
        return obs, ag, dg

    def step(self, actions):
        actions = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.E, ACT_DIM)
        obs, ag, dg = self._bufs()
        rew = np.zeros(self.E)
        done = np.zeros(self.E, dtype=np.uint8)
        succ = np.zeros(self.E, dtype=np.uint8)
        self.L.xo_pnp_step(self.m, self.cfg, self.E, _p(self.state), _p(actions), _p(obs), _p(ag), _p(dg),
                           _p(rew), _u8(done), _u8(succ))
        return obs, ag, dg, rew, done, succ

    def compute_reward(self, ag, g, reward_type=None):
        ag = np.ascontiguousarray(ag, dtype=np.float64).reshape(-1, 3)
        g = np.ascontiguousarray(g, dtype=np.float64).reshape(-1, 3)
        out = np.zeros(ag.shape[0])
        rt = self.cfg.reward_type if reward_type is None else REWARD_TYPES[reward_type]
        self.L.xo_pnp_compute_reward(self.m, rt, ag.shape[0], _p(ag), _p(g), _p(out))
        return out

    def debug_substeps(self, q_target, n):
        """n internal substeps toward joint targets [E, 9] (counterpart of the C ABI's xarm_debug_substeps)"""
        qt = np.ascontiguousarray(q_target, dtype=np.float64).reshape(self.E, 9)
        self.L.xo_pnp_substeps(self.m, self.E, _p(self.state), _p(qt), int(n))

    def get_state(self):
        return self.state.copy()

    def set_state(self, s):
        self.state[...] = np.asarray(s, dtype=np.float64).reshape(self.E, STATE_DIM)


class OracleReach:
    """Batched CPU XarmReach-v0 (xarm_reach.py), float64."""

    def __init__(self, num_envs, seed=0, env_id_offset=0, reward_type="sparse"):
        self.L = lib()
        js = load_model_json(REACH_JSON)
        self.m = build_model(js)
        r = js["reach"]
        c = XoReachCfg()
        c.seed, c.env_id_offset, c.reward_type, c.driver_link = seed, env_id_offset, REACH_REWARD_TYPES[reward_type], js["driver_link"]
        for k in ("time_step", "action_dt", "max_vel", "max_gripper_vel", "motor_force", "distance_threshold",
                  "n_substeps", "max_episode_steps"):
            setattr(c, k, r[k])
        for k in ("pos_low", "pos_high", "goal_low", "goal_high"):
            for i in range(3):
                getattr(c, k)[i] = r[k][i]
        for i in range(MAXD):
            c.joint_init_pos[i] = r["joint_init_pos"][i]
        self.cfg = c
        self.E = int(num_envs)
        self.state = np.zeros((self.E, REACH_STATE_DIM))
        self.L.xo_reach_init(self.m, self.cfg, self.E, _p(self.state))

    def _bufs(self):
        return np.zeros((self.E, REACH_OBS_DIM)), np.zeros((self.E, 3)), np.zeros((self.E, 3))

    def reset(self, mask=None):
        obs, ag, dg = self._bufs()
        mk = None if mask is None else _u8(np.ascontiguousarray(mask, dtype=np.uint8))
        self.L.xo_reach_reset(self.m, self.cfg, self.E, _p(self.state), mk, _p(obs), _p(ag), _p(dg))
        return obs, ag, dg

    def step(self, actions):
        actions = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.E, ACT_DIM)
        obs, ag, dg = self._bufs()
        rew, done, succ = np.zeros(self.E), np.zeros(self.E, np.uint8), np.zeros(self.E, np.uint8)
        fut = np.zeros(self.E, np.int32)
        self.L.xo_reach_step(self.m, self.cfg, self.E, _p(self.state), _p(actions), _p(obs), _p(ag), _p(dg), _p(rew),
                             _u8(done), _u8(succ), fut.ctypes.data_as(C.POINTER(C.c_int32)))
        return obs, ag, dg, rew, done, succ, fut

    def compute_reward(self, ag, g, reward_type=None):
        ag = np.ascontiguousarray(ag, dtype=np.float64).reshape(-1, 3)
        g = np.ascontiguousarray(g, dtype=np.float64).reshape(-1, 3)
        out = np.zeros(ag.shape[0])
        rt = self.cfg.reward_type if reward_type is None else REACH_REWARD_TYPES[reward_type]
        if self.L.xo_reach_compute_reward(self.cfg, rt, ag.shape[0], _p(ag), _p(g), _p(out)) != 0:
            raise ValueError("reward_type is stateful")
        return out

    def get_state(self):
        return self.state.copy()

    def set_state(self, s):
        self.state[...] = np.asarray(s, dtype=np.float64).reshape(self.E, REACH_STATE_DIM)


class OracleHandover:
    """Batched CPU XarmHandover-v0 (xarm_handover.py), float64.  num_obj = 1 (BASELINE config 5) or 2 (the reference's
    test.py configuration; sparse reward; use_stand: one stand per goal, :391-392)."""

    def __init__(self, num_envs, seed=0, env_id_offset=0, same_side_rate=0.5, goal_shape="ground", reward_type="sparse", use_stand=False,
                 num_obj=1):
        self.L = lib()
        assert num_obj in (1, 2)
        if num_obj == 2 and reward_type != "sparse":
            raise ValueError("num_obj = 2: sparse reward only (the reference's dense branch raises, xarm_handover.py:187-188)")
        self.num_obj = num_obj
        self.state_dim, self.obs_dim, self.goal_dim = (HO_STATE_DIM, HO_OBS_DIM, 3) if num_obj == 1 else (HO2_STATE_DIM, HO2_OBS_DIM, HO2_GOAL_DIM)
        self._fn = {k: getattr(self.L, ("xo_ho_" if num_obj == 1 else "xo_ho2_") + k) for k in ("init", "reset", "step", "compute_reward")}
        js = load_model_json()
        self.m = build_model(js)
        h = js["handover"]
        c = XoHoCfg()
        c.reward_type = {"sparse": 0, "dense": 1}[reward_type]
        c.use_stand = int(bool(use_stand))
        c.stand_below_goal = h["stand_below_goal"]
        for i in range(3):
            c.stand_half[i] = h["stand_half"][i]
        c.seed, c.env_id_offset, c.same_side_rate = seed, env_id_offset, same_side_rate
        c.goal_shape = 1 if goal_shape == "ground" else 0
        for k in ("n_ticks", "time_step", "action_dt", "max_vel", "max_gripper_vel", "gripper_low", "gripper_high", "height_offset",
                  "finger_motor_force", "distance_threshold", "table_x_min", "table_x_max", "table_half_y", "ground_z",
                  "reset_ticks", "max_episode_steps"):
            setattr(c, k, h[k])
        for k in ("pos_low", "pos_high", "eff_init_pos", "base_pos"):
            for a in range(2):
                for i in range(3):
                    getattr(c, k)[a][i] = h[k][a][i]
        for k, n in (("goal_low", 3), ("goal_high", 3), ("obj_low", 2), ("obj_high", 2), ("joint_init_pos", 9), ("base_yaw", 2),
                     ("obj_half", 3), ("eef2grip", 3)):
            for i in range(n):
                getattr(c, k)[i] = h[k][i]
        for k in ("spawn_min_dy", "goal_min_dy", "goal_min_obj_dist", "sample_max_tries"):
            setattr(c, k, h[k])
        self.cfg = c
        self.E = int(num_envs)
        self.state = np.zeros((self.E, self.state_dim))
        self._fn["init"](self.m, self.cfg, self.E, _p(self.state))

    def _bufs(self):
        return np.zeros((self.E, self.obs_dim)), np.zeros((self.E, self.goal_dim)), np.zeros((self.E, self.goal_dim))

    def reset(self, mask=None):
        obs, ag, dg = self._bufs()
        mk = None if mask is None else _u8(np.ascontiguousarray(mask, dtype=np.uint8))
        self._fn["reset"](self.m, self.cfg, self.E, _p(self.state), mk, _p(obs), _p(ag), _p(dg))
        return obs, ag, dg

    def step(self, actions):
        actions = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.E, HO_ACT_DIM)
        obs, ag, dg = self._bufs()
        rew, done, succ = np.zeros(self.E), np.zeros(self.E, np.uint8), np.zeros(self.E, np.uint8)
        self._fn["step"](self.m, self.cfg, self.E, _p(self.state), _p(actions), _p(obs), _p(ag), _p(dg), _p(rew), _u8(done), _u8(succ))
        return obs, ag, dg, rew, done, succ

    def compute_reward(self, ag, g):
        ag = np.ascontiguousarray(ag, dtype=np.float64).reshape(-1, self.goal_dim)
        g = np.ascontiguousarray(g, dtype=np.float64).reshape(-1, self.goal_dim)
        out = np.zeros(ag.shape[0])
        self._fn["compute_reward"](self.cfg, ag.shape[0], _p(ag), _p(g), _p(out))
        return out

    def get_state(self):
        return self.state.copy()

    def set_state(self, s):
        self.state[...] = np.asarray(s, dtype=np.float64).reshape(self.E, self.state_dim)


class OracleStackTower:
    """Batched CPU XarmPDStackTower-v0 (xarm_stack_tower.py: two arms, three cubes), float64."""

    def __init__(self, num_envs, seed=0, env_id_offset=0, reward_type="sparse"):
        self.L = lib()
        js = load_model_json()
        self.m = build_model(js)
        h = js["stack_tower"]
        c = XoStCfg()
        c.seed, c.env_id_offset = seed, env_id_offset
        c.reward_type = 0 if reward_type == "sparse" else 1
        for k in ("n_substeps", "max_episode_steps", "time_step", "action_dt", "max_vel", "max_gripper_vel", "gripper_low",
                  "gripper_high", "height_offset", "finger_motor_force", "distance_threshold", "cube_half", "cube_mass"):
            setattr(c, k, h[k])
        for k in ("pos_low", "pos_high", "base_pos"):
            for a in range(2):
                for i in range(3):
                    getattr(c, k)[a][i] = h[k][a][i]
        for k, n in (("goal_low", 2), ("goal_high", 2), ("obj_low", 2), ("obj_high", 2), ("joint_init_pos", 9), ("base_yaw", 2)):
            for i in range(n):
                getattr(c, k)[i] = h[k][i]
        self.cfg = c
        self.E = int(num_envs)
        self.state = np.zeros((self.E, ST_STATE_DIM))
        self.L.xo_st_init(self.m, self.cfg, self.E, _p(self.state))

    def _bufs(self):
        return np.zeros((self.E, ST_OBS_DIM)), np.zeros((self.E, 9)), np.zeros((self.E, 9))

    def reset(self, mask=None):
        obs, ag, dg = self._bufs()
        mk = None if mask is None else _u8(np.ascontiguousarray(mask, dtype=np.uint8))
        self.L.xo_st_reset(self.m, self.cfg, self.E, _p(self.state), mk, _p(obs), _p(ag), _p(dg))
        return obs, ag, dg

    def step(self, actions):
        actions = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.E, ST_ACT_DIM)
        obs, ag, dg = self._bufs()
        rew, done, succ = np.zeros(self.E), np.zeros(self.E, np.uint8), np.zeros(self.E, np.uint8)
        self.L.xo_st_step(self.m, self.cfg, self.E, _p(self.state), _p(actions), _p(obs), _p(ag), _p(dg), _p(rew), _u8(done), _u8(succ))
        return obs, ag, dg, rew, done, succ

    def compute_reward(self, ag, g, reward_type="sparse"):
        ag = np.ascontiguousarray(ag, dtype=np.float64).reshape(-1, 9)
        g = np.ascontiguousarray(g, dtype=np.float64).reshape(-1, 9)
        out = np.zeros(ag.shape[0])
        self.L.xo_st_compute_reward(self.cfg, 0 if reward_type == "sparse" else 1, ag.shape[0], _p(ag), _p(g), _p(out))
        return out

    def get_state(self):
        return self.state.copy()

    def set_state(self, s):
        self.state[...] = np.asarray(s, dtype=np.float64).reshape(self.E, ST_STATE_DIM)


def box_box(pA, RA, hA, pB, RB, hB, margin):
    """-> (points [n,3], normal [3] from B to A, dist [n]) of the cube/cube manifold (xo_box_box)"""
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (pA, RA, hA, pB, RB, hB)]
    pts, nrm, dist = np.zeros((4, 3)), np.zeros(3), np.zeros(4)
    L = lib()
    L.xo_box_box.restype = C.c_int
    L.xo_box_box.argtypes = [C.c_void_p] * 6 + [C.c_double] + [C.c_void_p] * 3
    n = L.xo_box_box(*[x.ctypes.data for x in a], float(margin), pts.ctypes.data, nrm.ctypes.data, dist.ctypes.data)
    return pts[:n].copy(), nrm, dist[:n].copy()


def dense_reward(if_grasp, hand_com, ag, g, model=None):
    m = model or build_model()
    a = [np.ascontiguousarray(x, dtype=np.float64) for x in (hand_com, ag, g)]
    return float(lib().xo_pnp_dense_reward(m, int(if_grasp), _p(a[0]), _p(a[1]), _p(a[2])))


def fk(q, model=None):
    m = model or build_model()
    q = _padq(q)
    pos = np.zeros((m.n_links, 3))
    rot = np.zeros((m.n_links, 9))
    lib().xo_fk(m, _p(q), _p(pos), _p(rot))
    return pos, rot.reshape(-1, 3, 3)


def num_dofs(m):
    return sum(1 for i in range(m.n_links) if m.jtype[i] != 0)


def _padq(q):
    out = np.zeros(MAXD)
    q = np.asarray(q, dtype=np.float64)
    out[:q.shape[0]] = q
    return out


def ik(q, target, max_iter=15, model=None):
    m = model or build_model()
    t = np.ascontiguousarray(target, dtype=np.float64)
    out = np.zeros(MAXD)
    lib().xo_ik(m, _p(_padq(q)), _p(t), max_iter, _p(out))
    return out[:num_dofs(m)]


def mass_matrix_inv(q, model=None):
    m = model or build_model()
    nd = num_dofs(m)
    out = np.zeros((nd, nd))
    lib().xo_mass_matrix_inv(m, _p(_padq(q)), _p(out))
    return out


def forward_dynamics(q, qd, tau, model=None):
    m = model or build_model()
    a = [_padq(x) for x in (q, qd, tau)]
    out = np.zeros(MAXD)
    lib().xo_forward_dynamics(m, _p(a[0]), _p(a[1]), _p(a[2]), _p(out))
    return out[:num_dofs(m)]


def philox(seed, c0, c1, c2, c3):
    out = (C.c_uint32 * 4)()
    lib().xo_philox(seed, c0, c1, c2, c3, out)
    return [int(x) for x in out]


def run_demo(env, get_state, set_state, substeps, n_sub=15):
    """Literal replay of XarmPickAndPlace._run_demo (xarm_pick_and_place.py:310-349), the reference's only grasp sequence,
    on `env` (the oracle or the HIP env through the callables): after reset() the object is teleported to [0.4, 0, 0.025]
    (:312), then 10 ticks with the arm motors on IK([0.4, 0, 0.125]) (:314-318), 3 ticks with the finger motors on 0.02 and
    the friction toggled from the finger contacts (:322-330; the arm motors keep their last targets), 10 ticks with the arm
    on IK([0.3, 0, 0.3]) and the same toggle (:336-346).  IK runs every tick from the current pose (maxNumIterations = 15)
    - here through the oracle's xo_ik for both users, the step-level tests cover the device's own IK.  The demo's finger
    motors are left at the default force; the model's finger force (1000, :210-211) is used.  During the first 10 ticks
    the finger motors still hold the reset's 0.02 target (:256-257).  Returns the states after each of the 23 ticks."""
    st = np.array(get_state(), dtype=np.float64)
    E = st.shape[0]
    st[:, 18:21] = [0.4, 0.0, 0.025]
    st[:, 21:25] = [0, 0, 0, 1]
    st[:, 25:31] = 0
    st[:, 34:50] = 0
    set_state(st)
    out = []
    qt = st[:, :9].copy()
    for phase, n_ticks, target in ((0, 10, (0.4, 0.0, 0.125)), (1, 3, None), (2, 10, (0.3, 0.0, 0.3))):
        for _ in range(n_ticks):
            st = np.array(get_state(), dtype=np.float64)
            if target is not None:
                for e in range(E):
                    qt[e, :7] = ik(st[e, :9], target)[:7]
            qt[:, 7:9] = 0.02
            if phase >= 1:
                st[:, 51] = st[:, 50]          # lateralFriction 100 / 1 from the current finger contacts (:325-330, :340-345)
                set_state(st)
            substeps(qt, n_sub)
            out.append(np.array(get_state(), dtype=np.float64))
    return np.stack(out)
