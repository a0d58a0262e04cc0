"""ctypes binding of include/xarm_hip.h (libxarm_hip.so).  There is NO fallback: if the HIP
library is missing or cannot be loaded, importing the product path raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libxarm_hip.so")

XARM_OK = 0
ENV_PICK_AND_PLACE = 0
ENV_REACH = 1
ENV_HANDOVER = 2
ENV_STACK_TOWER = 3
REACH_REWARD_TYPES = {"sparse": 0, "dense": 1, "dense_diff": 2}
REWARD_TYPES = {"sparse": 0, "dense_o2g": 1, "dense": 2}
GOAL_SHAPES = {"air": 0, "ground": 1}
RESET_COOP_LIMIT_DEFAULT = STEP_COOP_LIMIT_DEFAULT = 8192   # include/xarm_hip.h XARM_*_COOP_LIMIT_DEFAULT

EXPORTS = ["xarm_create", "xarm_destroy", "xarm_dims", "xarm_reset", "xarm_step", "xarm_compute_reward",
           "xarm_get_state", "xarm_set_state", "xarm_episode_steps", "xarm_debug_substeps", "xarm_timing_enable", "xarm_timing_read", "xarm_timing_read_reset", "xarm_kernel_limits", "xarm_pipeline_info", "xarm_stage_info", "xarm_debug_counts", "xarm_class_keys", "xarm_last_error",
           "xarm_version"]


class XarmConfig(C.Structure):
    _fields_ = [("num_envs", C.c_int64), ("env_id_offset", C.c_int64), ("seed", C.c_uint64),
                ("env_kind", C.c_int32), ("num_obj", C.c_int32), ("reward_type", C.c_int32),
                ("goal_shape", C.c_int32), ("init_grasp_rate", C.c_float), ("goal_ground_rate", C.c_float),
                ("auto_reset", C.c_int32), ("device", C.c_int32), ("same_side_rate", C.c_float), ("reset_coop_limit", C.c_int32),
                ("step_coop_limit", C.c_int32), ("use_stand", C.c_int32)]


class XarmDims(C.Structure):
    _fields_ = [("obs_dim", C.c_int32), ("goal_dim", C.c_int32), ("act_dim", C.c_int32),
                ("state_dim", C.c_int32), ("max_episode_steps", C.c_int32), ("n_substeps", C.c_int32)]


class XarmNativeError(RuntimeError):
    pass


_lib = None


def load(path=None):
    """Load libxarm_hip.so and declare every entry point of include/xarm_hip.h."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # XARM_HIP_LIB: a development variant of the library (tools/coop_split.sh, tools/variant_time.sh) loaded from its own
    # path; xarm_version() tells a timing variant from the product build.  Still no fallback: a missing file raises.
    p = path or os.environ.get("XARM_HIP_LIB") or LIB_PATH
    if not os.path.exists(p):
        raise XarmNativeError(
            "HIP extension %s not found: build it with `python -m gym_xarm_amd.build` (hipcc, gfx950). "
            "gym_xarm_amd has no CPU fallback." % p)
    L = C.CDLL(p)
    vp, fp, u8p = C.c_void_p, C.c_void_p, C.c_void_p  # device pointers travel as integers
    L.xarm_create.argtypes = [C.POINTER(XarmConfig), C.POINTER(C.c_void_p)]
    L.xarm_destroy.argtypes = [vp]
    L.xarm_dims.argtypes = [vp, C.POINTER(XarmDims)]
    L.xarm_reset.argtypes = [vp, u8p, fp, fp, fp, vp]
    L.xarm_step.argtypes = [vp, fp, fp, fp, fp, fp, u8p, u8p, fp, vp]
    L.xarm_compute_reward.argtypes = [vp, fp, fp, C.c_int64, fp, vp]
    L.xarm_get_state.argtypes = [vp, fp, vp]
    L.xarm_set_state.argtypes = [vp, fp, vp]
    L.xarm_episode_steps.argtypes = [vp, fp, vp]
    L.xarm_debug_substeps.argtypes = [vp, fp, C.c_int32, vp]
    L.xarm_timing_enable.argtypes = [vp, C.c_int32]
    L.xarm_timing_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.xarm_timing_read_reset.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.xarm_kernel_limits.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.xarm_pipeline_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.xarm_stage_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.xarm_debug_counts.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), vp]
    L.xarm_class_keys.argtypes = [vp, vp, vp]
    L.xarm_last_error.argtypes = [vp]
    L.xarm_last_error.restype = C.c_char_p
    L.xarm_version.argtypes = []
    L.xarm_version.restype = C.c_char_p
    for name in EXPORTS:
        if name not in ("xarm_last_error", "xarm_version"):
            getattr(L, name).restype = C.c_int
    if path is None:
        _lib = L
    return L


def loaded_path():
    """the file the product path bound: csrc/libxarm_hip.so, or a development variant named by XARM_HIP_LIB"""
    return os.environ.get("XARM_HIP_LIB") or LIB_PATH


def check(L, handle, rc, what):
    if rc != XARM_OK:
        msg = L.xarm_last_error(handle)
        raise XarmNativeError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
