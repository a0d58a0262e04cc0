#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own NumPy code (run in the build container
only, where /root/reference exists; the fixtures are committed, the reference is never copied).

The reference modules import gym and pybullet at module level; neither is installed, so they are
loaded with inert stub modules in sys.modules and only the pure-NumPy methods are called, unbound,
on a SimpleNamespace `self` (SURVEY.md 8c 'What can be imported here'):
  XarmPickAndPlace.compute_reward   xarm_pick_and_place.py:155  (sparse, dense_o2g)
  XarmPickAndPlace._is_success      xarm_pick_and_place.py:289
  XarmPickAndPlace._subgoal_distances :293
The step's `done` expression (:117) is evaluated literally as written there.
"""
import importlib.util
import os
import sys
import types
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/gym_xarm/envs/xarm_pick_and_place.py"
OUT = os.path.join(ROOT, "tests", "golden")


def stub_modules():
    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            m = _Any(self.__name__ + "." + name)
            return m

        def __call__(self, *a, **k):
            return None

    gym = _Any("gym")
    gym.GoalEnv = object
    for name in ("gym", "gym.utils", "gym.spaces", "gym.wrappers", "gym.wrappers.monitoring", "pybullet", "pybullet_data"):
        mod = gym if name == "gym" else _Any(name)
        sys.modules[name] = mod
    sys.modules["gym"].error = _Any("gym.error")
    sys.modules["gym"].spaces = sys.modules["gym.spaces"]
    sys.modules["gym"].utils = sys.modules["gym.utils"]
    sys.modules["gym.utils"].seeding = _Any("gym.utils.seeding")
    sys.modules["gym.wrappers.monitoring"].video_recorder = _Any("video_recorder")


def main():
    stub_modules()
    spec = importlib.util.spec_from_file_location("ref_pnp", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cls = mod.XarmPickAndPlace
    rng = np.random.default_rng(20240607)
    n = 512
    g = rng.uniform([0.35, -0.25, 0.025], [0.45, 0.25, 0.27], size=(n, 3))
    # achieved goals: a mix of far points, near-threshold shells and exact hits
    direction = rng.normal(size=(n, 3))
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    radius = np.concatenate([rng.uniform(0, 0.3, n // 2), 0.05 + rng.uniform(-2e-3, 2e-3, n // 4),
                             rng.uniform(0, 0.05, n - n // 2 - n // 4)])
    ag = g + direction * radius[:, None]
    ag[:8] = g[:8]
    out = {"achieved_goal": ag, "goal": g}
    for rt in ("sparse", "dense_o2g"):
        self = SimpleNamespace(config={"reward_type": rt, "num_obj": 1}, distance_threshold=0.05)
        self._subgoal_distances = lambda a, b, s=self: cls._subgoal_distances(s, a, b)
        out["reward_" + rt] = np.asarray(cls.compute_reward(self, ag, g, {}))
        # single-row calls, as step() does (:116)
        out["reward_single_" + rt] = np.array([cls.compute_reward(self, ag[i], g[i], {}) for i in range(64)])
    succ, done = [], []
    for i in range(n):
        self = SimpleNamespace(goal=g[i].reshape(1, 3), distance_threshold=0.05, config={"num_obj": 1})
        s = cls._is_success(self, ag[i], g[i])
        succ.append(np.asarray(s).reshape(-1)[0])
        for num_steps in (1, 49, 50):
            # literal restatement of xarm_pick_and_place.py:117
            d = (np.linalg.norm(ag[i] - g[i].reshape(1, 3).flatten(), axis=-1) < 0.05) or num_steps == 50
            done.append(bool(d))
    out["is_success"] = np.array(succ, dtype=np.float32)
    out["done_steps_1_49_50"] = np.array(done, dtype=np.uint8).reshape(n, 3)
    # 'dense' (:166-175) reads the simulator: give the reference a scripted `p` whose answers we choose
    class FakeBullet:
        def __init__(self):
            self.grasp, self.hand = False, np.zeros(3)

        def getContactPoints(self, *a):
            return [(0,)] if self.grasp else []

        def getLinkState(self, body, link):
            return (tuple(self.hand),)
    fake = FakeBullet()
    mod.p = fake
    nd = 256
    hand = rng.uniform([0.3, -0.3, 0.15], [0.5, 0.3, 0.4], size=(nd, 3))
    grasp = rng.random(nd) < 0.5
    agd = hand - [0, 0, 0.067] + rng.normal(scale=0.03, size=(nd, 3))
    agd[:, 2] = np.where(rng.random(nd) < 0.5, rng.uniform(0.0, 0.05, nd), rng.uniform(0.05, 0.3, nd))
    gd = rng.uniform([0.35, -0.25, 0.025], [0.45, 0.25, 0.27], size=(nd, 3))
    dense = []
    for i in range(nd):
        fake.grasp, fake.hand = bool(grasp[i]), hand[i]
        self = SimpleNamespace(config={"reward_type": "dense", "num_obj": 1}, distance_threshold=0.05, xarm=0, legos=[1],
                               finger1_index=10, finger2_index=11, gripper_base_index=9, eef2grip_offset=[0, 0, 0.088 - 0.021])
        self._subgoal_distances = lambda a, b, s=self: cls._subgoal_distances(s, a, b)
        dense.append(float(cls.compute_reward(self, agd[i], gd[i], {})))
    out.update(dense_hand_com=hand, dense_if_grasp=grasp.astype(np.uint8), dense_achieved_goal=agd, dense_goal=gd,
               dense_reward=np.array(dense))
    os.makedirs(OUT, exist_ok=True)
    np.savez(os.path.join(OUT, "pnp_reward_reference.npz"), **out)
    print("wrote pnp_reward_reference.npz:", {k: v.shape for k, v in out.items()})


def reach():
    """XarmReachEnv.compute_reward (xarm_reach.py:107-116) incl. the stateful dense_diff, _is_success (:175-177)
    and the done / future_length expressions of step (:88-93), evaluated by the reference's own code."""
    stub_modules()
    spec = importlib.util.spec_from_file_location("ref_reach", "/root/reference/gym_xarm/envs/xarm_reach.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cls = mod.XarmReachEnv
    rng = np.random.default_rng(424242)
    n = 512
    g = rng.uniform([0.3, -0.25, 0.3], [0.5, 0.25, 0.4], size=(n, 3))
    direction = rng.normal(size=(n, 3))
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    radius = np.concatenate([rng.uniform(0, 0.4, n // 2), 0.05 + rng.uniform(-2e-3, 2e-3, n // 4), rng.uniform(0, 0.05, n - n // 2 - n // 4)])
    ag = g + direction * radius[:, None]
    out = {"achieved_goal": ag, "goal": g}
    for rt in ("sparse", "dense"):
        self = SimpleNamespace(reward_type=rt, distance_threshold=0.05)
        out["reward_" + rt] = np.asarray(cls.compute_reward(self, ag, g, {}))
    # dense_diff along a trajectory of one env: d_old is carried by the object
    self = SimpleNamespace(reward_type="dense_diff", distance_threshold=0.05, d_old=0.3)
    traj = g[0] + np.cumsum(rng.normal(scale=0.02, size=(64, 3)), axis=0)
    out["diff_traj"], out["diff_goal"], out["diff_d_old0"] = traj, g[0], np.float64(0.3)
    out["reward_dense_diff"] = np.array([cls.compute_reward(self, traj[i], g[0], {}) for i in range(64)])
    succ = []
    for i in range(n):
        self = SimpleNamespace(goal=g[i], distance_threshold=0.05)
        succ.append(cls._is_success(self, ag[i], g[i]))
    out["is_success"] = np.array(succ, dtype=np.float32)
    steps = np.arange(1, 27)
    out["steps"] = steps
    out["done"] = np.array([s == 25 for s in steps], dtype=np.uint8)          # :93
    out["future_length"] = np.array([25 - s for s in steps], dtype=np.int32)   # :90
    np.savez(os.path.join(OUT, "reach_reward_reference.npz"), **out)
    print("wrote reach_reward_reference.npz")


def handover():
    """XarmHandover.compute_reward sparse (:164-183, incl. the batch form and N = 2), _is_success (:395-402) and the
    done expression of step (:138) from the reference's own code."""
    stub_modules()
    sys.modules["pybullet_utils"] = sys.modules["pybullet"]
    sys.modules["pybullet_utils.bullet_client"] = sys.modules["pybullet"]
    sys.modules["pkgutil"] = __import__("pkgutil")
    spec = importlib.util.spec_from_file_location("ref_ho", "/root/reference/gym_xarm/envs/xarm_handover.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cls = mod.XarmHandover
    rng = np.random.default_rng(777)
    n = 512
    g = rng.uniform([-0.28, -0.18, 0.025], [0.28, 0.18, 0.2], size=(n, 3))
    direction = rng.normal(size=(n, 3))
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    radius = np.concatenate([rng.uniform(0, 0.4, n // 2), 0.05 + rng.uniform(-2e-3, 2e-3, n // 4), rng.uniform(0, 0.05, n - n // 2 - n // 4)])
    ag = g + direction * radius[:, None]
    self = SimpleNamespace(reward_type="sparse", distance_threshold=0.05, config={"num_obj": 1})
    out = {"achieved_goal": ag, "goal": g}
    out["reward_batch"] = np.asarray(cls.compute_reward(self, ag, g, {}))
    out["reward_single"] = np.array([cls.compute_reward(self, ag[i], g[i], {}) for i in range(64)])
    out["is_success"] = np.array([cls._is_success(self, ag[i], g[i]) for i in range(n)])
    steps = np.array([1, 99, 100])
    out["done"] = np.array([[(s == 100) or bool(out["is_success"][i]) for s in steps] for i in range(n)], dtype=np.uint8)  # :138 with TimeLimit(100)
    # the staged dense reward (:184-199): reads the simulator through self._p and the grasp flags of _set_action; the
    # reference's fourth branch (only arm 2 grasps) raises NameError (`d` undefined, :199) - recorded as such
    class FakeClient:
        def __init__(self):
            self.hand = {1: np.zeros(3), 2: np.zeros(3)}

        def getLinkState(self, body, link):
            return (tuple(self.hand[body]),)
    fake = FakeClient()
    nd = 384
    hand1 = rng.uniform([-0.35, -0.2, 0.1], [0.05, 0.2, 0.3], size=(nd, 3))
    hand2 = rng.uniform([-0.05, -0.2, 0.1], [0.35, 0.2, 0.3], size=(nd, 3))
    if1, if2 = rng.random(nd) < 0.5, rng.random(nd) < 0.4
    agd = np.where((rng.random(nd) < 0.5)[:, None], hand1 - [0, 0, 0.067], hand2 - [0, 0, 0.067]) + rng.normal(scale=0.04, size=(nd, 3))
    agd[:, 2] = np.where(rng.random(nd) < 0.5, rng.uniform(0.0, 0.05, nd), rng.uniform(0.05, 0.25, nd))
    gd = rng.uniform([-0.28, -0.18, 0.025], [0.28, 0.18, 0.2], size=(nd, 3))
    dense, raised = np.zeros(nd), np.zeros(nd, np.uint8)
    for i in range(nd):
        fake.hand[1], fake.hand[2] = hand1[i], hand2[i]
        self = SimpleNamespace(reward_type="dense", distance_threshold=0.05, config={"num_obj": 1}, _p=fake, xarm_1=1, xarm_2=2,
                               gripper_base_index=9, eef2grip_offset=[0, 0, 0.088 - 0.021], if_xarm1_grasp=bool(if1[i]), if_xarm2_grasp=bool(if2[i]))
        try:
            dense[i] = float(cls.compute_reward(self, agd[i], gd[i], {}))
        except NameError:
            raised[i] = 1
            dense[i] = np.nan
    assert raised.sum() == ((~if1) & if2).sum() and raised.sum() > 20
    out.update(dense_hand_com_1=hand1, dense_hand_com_2=hand2, dense_if_1=if1.astype(np.uint8), dense_if_2=if2.astype(np.uint8),
               dense_achieved_goal=agd, dense_goal=gd, dense_reward=dense, dense_reference_raises=raised)
    # num_obj = 2 (the reference's own test configuration, test.py:9-15): rows of 6 = two sticks; its own generator so that
    # the arrays above stay what they were
    rng2 = np.random.default_rng(778)
    n2 = 512
    g2 = rng2.uniform([-0.28, -0.18, 0.025] * 2, [0.28, 0.18, 0.2] * 2, size=(n2, 6))
    ag2 = g2.copy()
    for o in range(2):
        d = rng2.normal(size=(n2, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        r = np.concatenate([rng2.uniform(0, 0.3, n2 // 4), 0.05 + rng2.uniform(-2e-3, 2e-3, n2 // 4), rng2.uniform(0, 0.05, n2 - 2 * (n2 // 4))])
        ag2[:, 3 * o:3 * o + 3] += d * rng2.permutation(r)[:, None]
    self2 = SimpleNamespace(reward_type="sparse", distance_threshold=0.05, config={"num_obj": 2})
    out["n2_achieved_goal"], out["n2_goal"] = ag2, g2
    out["n2_reward_batch"] = np.asarray(cls.compute_reward(self2, ag2, g2, {}))                       # :177-183 batch form
    out["n2_reward_single"] = np.array([cls.compute_reward(self2, ag2[i], g2[i], {}) for i in range(96)])  # as step() calls it (:137)
    out["n2_is_success"] = np.array([cls._is_success(self2, ag2[i], g2[i]) for i in range(n2)])            # :395-402
    assert set(np.unique(out["n2_reward_batch"])) == {0.0, -1.0, -2.0} and 0 < out["n2_is_success"].mean() < 1
    # the dense branch cannot run with two sticks: (3,) grip position minus the (6,) achieved_goal (:187)
    fake.hand[1], fake.hand[2] = hand1[0], hand2[0]
    selfd = SimpleNamespace(reward_type="dense", distance_threshold=0.05, config={"num_obj": 2}, _p=fake, xarm_1=1, xarm_2=2,
                            gripper_base_index=9, eef2grip_offset=[0, 0, 0.088 - 0.021], if_xarm1_grasp=False, if_xarm2_grasp=False)
    try:
        cls.compute_reward(selfd, ag2[0], g2[0], {})
        out["n2_dense_raises"] = np.uint8(0)
    except ValueError:
        out["n2_dense_raises"] = np.uint8(1)
    assert out["n2_dense_raises"] == 1
    # the scripted controller XarmHandover.ezpolicy (:404-446): pure NumPy on the observation dict.  Rows cover every branch:
    # reach (arm 1 far / inside the 0.1 closing shell / inside the 0.05 grasp shell), lift (arm 1 holds, arm 2 far / near),
    # both hold, the thresholds themselves (0.05 and 0.1 +- 2e-3) and finger openings on both sides of the 0.25 test
    rng3 = np.random.default_rng(779)
    ne = 640
    eo = np.zeros((ne, 29))
    eo[:, 0:3] = rng3.uniform([-0.28, -0.18, 0.02], [0.28, 0.18, 0.25], size=(ne, 3))
    eo[:, 3:7] = [0, 0, 0, 1]
    eo[:, 7:13] = rng3.normal(scale=0.1, size=(ne, 6))
    shells = np.array([0.02, 0.045, 0.05, 0.055, 0.08, 0.1, 0.12, 0.3])
    for k, (lo, vel, fin) in enumerate(((13, 16, 19), (21, 24, 27))):
        d = rng3.normal(size=(ne, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        r = shells[rng3.integers(0, shells.size, ne)] + rng3.uniform(-2e-3, 2e-3, ne)
        eo[:, lo:lo + 3] = eo[:, 0:3] + d * np.abs(r)[:, None]
        eo[:, vel:vel + 3] = rng3.normal(scale=0.1, size=(ne, 3))
        eo[:, fin] = np.where(rng3.random(ne) < 0.9, rng3.uniform(0.0, 0.04, ne), rng3.uniform(0.2, 0.3, ne))
        eo[:, fin + 1] = rng3.normal(scale=0.01, size=ne)
    ea = np.array([np.asarray(cls.ezpolicy(SimpleNamespace(), {"observation": eo[i]}), dtype=np.float64) for i in range(ne)])
    n1, n2 = np.linalg.norm(eo[:, 0:3] - eo[:, 13:16], axis=1), np.linalg.norm(eo[:, 0:3] - eo[:, 21:24], axis=1)
    ig1, ig2 = (eo[:, 19] < 0.25) & (n1 < 0.05), (eo[:, 27] < 0.25) & (n2 < 0.05)
    assert (~ig1).sum() > 50 and (ig1 & ~ig2).sum() > 50 and (ig1 & ig2).sum() > 20 and ((eo[:, 19] > 0.25) & (n1 < 0.05)).sum() > 3
    out["ez_observation"], out["ez_action"] = eo, ea
    np.savez(os.path.join(OUT, "handover_reward_reference.npz"), **out)
    print("wrote handover_reward_reference.npz")


def stack_tower():
    """XarmStackTowerEnv.compute_reward (xarm_stack_tower.py:124-129: sparse -(d > 0.09) over the 9-vector, else -d)
    and _is_success (:221-223) from the reference's own code, plus the tower goal layout of _sample_goal (:212-219)
    evaluated with the reference's arithmetic (height_offset * (2 i + 1))."""
    stub_modules()
    spec = importlib.util.spec_from_file_location("ref_st", "/root/reference/gym_xarm/envs/xarm_stack_tower.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cls = mod.XarmStackTowerEnv
    rng = np.random.default_rng(31337)
    n = 512
    xy = rng.uniform([-0.3, -0.2], [0.3, 0.2], size=(n, 2))
    g = np.stack([np.concatenate((xy[i], [0.025 * (2 * k + 1)])) for i in range(n) for k in range(3)]).reshape(n, 9)
    direction = rng.normal(size=(n, 9))
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    radius = np.concatenate([rng.uniform(0, 0.5, n // 2), 0.09 + rng.uniform(-2e-3, 2e-3, n // 4), rng.uniform(0, 0.09, n - n // 2 - n // 4)])
    ag = g + direction * radius[:, None]
    out = {"achieved_goal": ag, "goal": g}
    for rt in ("sparse", "dense"):
        self = SimpleNamespace(reward_type=rt, distance_threshold=0.03 * 3)
        out["reward_" + rt] = np.asarray(cls.compute_reward(self, ag, g, {}))
        out["reward_single_" + rt] = np.array([cls.compute_reward(self, ag[i], g[i], {}) for i in range(64)])
    out["is_success"] = np.array([cls._is_success(SimpleNamespace(goal=g[i], distance_threshold=0.09), ag[i], g[i]) for i in range(n)],
                                 dtype=np.float32)
    np.savez(os.path.join(OUT, "stack_reward_reference.npz"), **out)
    print("wrote stack_reward_reference.npz")


if __name__ == "__main__":
    main()
    reach()
    handover()
    stack_tower()
