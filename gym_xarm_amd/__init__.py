"""gym_xarm_amd - MI355X-native batched Xarm7 manipulation environments.

Registry mirroring /root/reference/gym_xarm/__init__.py:6-22 (`register(id, entry_point,
max_episode_steps)`) plus the IDs the reference's README/test.py name (README.md:43-48, test.py:8).
`make(id, num_envs=E, config=...)` returns the torch VecEnv; `make(id, config=...)` without
num_envs returns the single-env object with the reference's numpy call surface.
"""
from . import _native  # noqa: F401
from .vec_env import XarmPickAndPlaceVecEnv, XarmReachVecEnv, XarmHandoverVecEnv, XarmStackTowerVecEnv  # noqa: F401

__version__ = "0.1.0"

_REGISTRY = {}


def register(id, entry_point, max_episode_steps, vec_entry_point=None):
    _REGISTRY[id] = dict(entry_point=entry_point, max_episode_steps=max_episode_steps, vec_entry_point=vec_entry_point)


def registered_ids():
    return sorted(_REGISTRY)


def spec(id):
    if id not in _REGISTRY:
        raise KeyError("No registered env with id: %s (known: %s)" % (id, ", ".join(registered_ids())))
    return _REGISTRY[id]


def _resolve(path):
    mod, _, name = path.partition(":")
    import importlib
    return getattr(importlib.import_module(mod), name)


def make(id, num_envs=None, config=None, **kwargs):
    s = spec(id)
    if num_envs is None:
        return _resolve(s["entry_point"])(config, **kwargs)
    return _resolve(s["vec_entry_point"])(num_envs, config=config, **kwargs)


# registered in the reference (gym_xarm/__init__.py:18-22) and its BASELINE.json alias
for _id in ("XarmPickAndPlace-v1", "XarmPDPickAndPlace-v0"):
    register(_id, "gym_xarm_amd.envs:XarmPickAndPlace", 50, "gym_xarm_amd.vec_env:XarmPickAndPlaceVecEnv")
# gym_xarm/__init__.py:6-10
register("XarmReach-v0", "gym_xarm_amd.envs:XarmReachEnv", 25, "gym_xarm_amd.vec_env:XarmReachVecEnv")
# gym_xarm/__init__.py:12-16 and its README / BASELINE.json alias
for _id in ("XarmHandover-v0", "XarmPDHandover-v0"):
    register(_id, "gym_xarm_amd.envs:XarmHandover", 100, "gym_xarm_amd.vec_env:XarmHandoverVecEnv")
# the flat-observation id the reference trains on (benchmark/train.py:67, plot.py:5; README.md:38 spells it
# XarmPDHandoverDenseEnvNoGoal-v1): unregistered in the reference itself; observation 29 (what its saved vec_normalize.pkl
# documents), staged dense reward (train.py:66)
for _id in ("XarmPDHandoverNoGoal-v1", "XarmPDHandoverDenseEnvNoGoal-v1"):
    register(_id, "gym_xarm_amd.sb3_adapter:XarmHandoverNoGoal", 100, "gym_xarm_amd.sb3_adapter:make_handover_nogoal_vec")
# not in the reference's registry (the class exists, xarm_stack_tower.py:13, _max_episode_steps = 50 :43); BASELINE.json
# config 4 names it XarmPDStackTower-v0
for _id in ("XarmStackTower-v0", "XarmPDStackTower-v0"):
    register(_id, "gym_xarm_amd.envs:XarmStackTowerEnv", 50, "gym_xarm_amd.vec_env:XarmStackTowerVecEnv")


def register_with_gym():
    """Mirror the registry into a real `gym` / `gymnasium` when one is importable, so that `gym.make('Xarm*-v0',
    config=...)` resolves to these classes exactly as the reference's `gym_xarm/__init__.py:6-22` arranges it
    (entry_point + max_episode_steps).  Neither package is installed in the build image: then this is a no-op and
    `gym_xarm_amd.make` is the registry.  Returns the names of the packages that took the registrations."""
    took = []
    for name in ("gym", "gymnasium"):
        try:
            mod = __import__(name)
            reg = __import__(name + ".envs.registration", fromlist=["register"])
        except Exception:
            continue
        known = getattr(getattr(mod.envs, "registry", None), "keys", lambda: [])()
        for env_id, spec_ in _REGISTRY.items():
            if env_id in known:
                continue
            try:
                reg.register(id=env_id, entry_point=spec_["entry_point"], max_episode_steps=spec_["max_episode_steps"])
            except Exception:
                continue
        took.append(name)
    return took


_GYM_BACKENDS = register_with_gym()
