"""The drop-in boundary without a GPU: the C-ABI library loads and exports exactly what
include/xarm_hip.h declares, the product never reaches into oracle/, the registry / spaces mirror
the reference, and the product fails loudly (no CPU fallback) when there is no HIP device."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "include", "xarm_hip.h")
LIB = os.path.join(ROOT, "gym_xarm_amd", "csrc", "libxarm_hip.so")


def declared_functions():
    src = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(xarm_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        from gym_xarm_amd import build
        build.build()
    return C.CDLL(LIB)


def test_library_exports_every_declared_symbol(lib):
    names = declared_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), "missing export %s" % n
    from gym_xarm_amd import _native
    assert sorted(_native.EXPORTS) == names


def test_abi_struct_layout_matches_header():
    from gym_xarm_amd import _native
    assert C.sizeof(_native.XarmConfig) == 72 and C.sizeof(_native.XarmDims) == 24
    src = open(HDR).read()
    fields = re.findall(r"^\s+(?:u?int\d+_t|float)\s+(\w+);", src[src.index("typedef struct xarm_config"):src.index("} xarm_config;")], flags=re.M)
    assert fields == [f[0] for f in _native.XarmConfig._fields_]


def test_library_contains_gfx950_code_object(lib):
    out = subprocess.run(["strings", "-n", "6", LIB], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_no_device_errors_are_reported_not_thrown(lib):
    from gym_xarm_amd import _native
    L = _native.load()
    bad = _native.XarmConfig(0, 0, 0, 0, 1, 0, 0, 0.0, 0.0, 1, 0, 0.0, 0)  # num_envs = 0
    h = C.c_void_p(0)
    assert L.xarm_create(C.byref(bad), C.byref(h)) == -1 and not h.value
    assert b"num_envs" in L.xarm_last_error(None)
    bad = _native.XarmConfig(4, 0, 0, 0, 2, 0, 0, 0.0, 0.0, 1, 0, 0.0, 0)  # num_obj = 2 unsupported
    assert L.xarm_create(C.byref(bad), C.byref(h)) == -1
    assert L.xarm_destroy(None) == 0 and L.xarm_step(None, *([None] * 9)) == -1


def test_pnp_num_obj_above_one_is_a_broadcast_error_in_the_reference():
    """XarmPickAndPlace.step -> _is_success (xarm_pick_and_place.py:114, :289-291) evaluates
    `np.linalg.norm(achieved_goal - self.goal, axis=-1)` with achieved_goal flat (3N,) (`_get_obs` :239) and self.goal
    (N, 3) (`_sample_goal` returns np.array(goal), :287): defined for N = 1 only.  The VecEnv therefore refuses
    num_obj > 1 for PickAndPlace instead of inventing a behaviour."""
    import numpy as np
    assert np.linalg.norm(np.zeros(3) - np.zeros((1, 3)), axis=-1).shape == (1,)
    for n in (2, 3, 4):
        with pytest.raises(ValueError):
            np.zeros(3 * n) - np.zeros((n, 3))
    from gym_xarm_amd.vec_env import XarmPickAndPlaceVecEnv
    with pytest.raises(NotImplementedError, match="broadcast"):
        XarmPickAndPlaceVecEnv._check_config(None, {"num_obj": 2})


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "gym_xarm_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dp, f)).read()
                code = "\n".join(l for l in text.split("\n") if not l.strip().startswith(("//", "#", "*", "/*")))
                assert not re.search(r"(from|import)\s+oracle|libxarm_oracle|xarm_oracle\.h|hostbuild", code), f
    # the development tools measure and generate for the product: those that need the oracle live under tests/tools/
    for dp, _, files in os.walk(os.path.join(ROOT, "tools")):
        for f in files:
            if f.endswith((".py", ".sh", ".cpp", ".hip")):
                code = open(os.path.join(dp, f)).read()
                assert not re.search(r"(from|import)\s+oracle\b|libxarm_oracle|xarm_oracle\.h", code), f


def test_registry_and_spaces_mirror_reference():
    import gym_xarm_amd as g
    assert set(g.registered_ids()) >= {"XarmPickAndPlace-v1", "XarmPDPickAndPlace-v0"}
    assert g.spec("XarmPickAndPlace-v1")["max_episode_steps"] == 50   # gym_xarm/__init__.py:18-22
    with pytest.raises(KeyError):
        g.spec("XarmNope-v0")
    from gym_xarm_amd.spaces import Box, Dict
    a = Box(-1.0, 1.0, shape=(4,), dtype=np.float32)
    assert a.contains(a.sample()) and not a.contains(np.full(4, 2, np.float32)) and not a.contains(np.zeros(3, np.float32))
    d = Dict(dict(observation=Box(-np.inf, np.inf, shape=(24,), dtype=np.float32)))
    assert d.contains({"observation": np.zeros(24, np.float32)})


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import gym_xarm_amd as g
    from gym_xarm_amd._native import XarmNativeError
    with pytest.raises(XarmNativeError, match="no CPU fallback"):
        g.make("XarmPDPickAndPlace-v0", num_envs=4)
    with pytest.raises(XarmNativeError):
        g.make("XarmPickAndPlace-v1", config={"GUI": False, "num_obj": 1, "reward_type": "sparse",
                                               "init_grasp_rate": 0.0, "goal_ground_rate": 0.0, "goal_shape": "air"})


def test_missing_extension_is_an_error(tmp_path):
    from gym_xarm_amd import _native
    with pytest.raises(_native.XarmNativeError, match="not found"):
        _native.load(str(tmp_path / "libxarm_hip.so"))


@pytest.mark.gpu
def test_gpu_plain_c_host_program_on_the_abi():
    """examples/abi_step_loop.c: a gcc-built plain-C program that binds include/xarm_hip.h directly (hipMalloc'd buffers, its
    own stream, no torch, no Python) resets and steps 2 048 envs with auto-reset and exits 0 with finite observations"""
    import subprocess
    from gym_xarm_amd import build as b
    try:
        exe = b.build_example(verbose=False)
    except Exception as e:
        pytest.skip("examples/abi_step_loop could not be built here (%s)" % e)
    out = subprocess.run([exe, "2048", "30"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "2048 envs x 30 steps" in out.stdout and "non-finite 0" in out.stdout and "obs_dim 24" in out.stdout
