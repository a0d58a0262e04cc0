"""XarmPDStackTower-v0 oracle (oracle/xarm_oracle_stack.inc.c): reward arithmetic pinned by the reference's own
NumPy code, cube/cube manifold known answers, statics, and a scripted pick-and-stack (the behaviour the env exists
for).  Physics parity vs PyBullet is UNPINNED (PyBullet absent)."""
import numpy as np
import pytest

H = np.full(3, 0.025)
I3 = np.eye(3)


def rot(ax, a):
    c, s = np.cos(a), np.sin(a)
    return {0: np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), 1: np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
            2: np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[ax]


def test_stack_reward_matches_reference_numpy(oracle):
    d = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "stack_reward_reference.npz"))
    env = oracle.OracleStackTower(1)
    ag, g = d["achieved_goal"], d["goal"]
    assert np.array_equal(env.compute_reward(ag, g, "sparse"), d["reward_sparse"].astype(np.float64))   # xarm_stack_tower.py:124-127
    assert np.allclose(env.compute_reward(ag, g, "dense"), d["reward_dense"], rtol=0, atol=1e-15)      # :128-129
    assert np.array_equal(env.compute_reward(ag[:64], g[:64], "sparse"), d["reward_single_sparse"].astype(np.float64))
    # is_success is the complement of the sparse penalty except exactly on the threshold (:221-223 uses <, :126 uses >)
    succ = (np.linalg.norm(ag - g, axis=1) < 0.09).astype(np.float32)
    assert np.array_equal(succ, d["is_success"])


def test_box_box_known_answers(oracle):
    # aligned stack: the four bottom corners of the upper cube, on the shared plane, normal up (from B to A)
    p, n, d = oracle.box_box([0, 0, 0.075], I3, H, [0, 0, 0.025], I3, H, 0.005)
    assert len(p) == 4 and np.allclose(n, [0, 0, 1]) and np.allclose(d, 0, atol=1e-15)
    assert np.allclose(sorted(map(tuple, np.round(p, 12))), sorted([(x, y, 0.05) for x in (-.025, .025) for y in (-.025, .025)]))
    # 1 mm of penetration, upper cube yawed 45 degrees: the octagon is reduced to 4 points 90 degrees apart
    p, n, d = oracle.box_box([0, 0, 0.074], rot(2, np.pi / 4), H, [0, 0, 0.025], I3, H, 0.005)
    assert len(p) == 4 and np.allclose(n, [0, 0, 1]) and np.allclose(d, -0.001)
    ang = np.sort(np.arctan2(p[:, 1], p[:, 0]))
    assert np.allclose(np.diff(ang), np.pi / 2, atol=1e-9) and np.allclose(np.linalg.norm(p[:, :2], axis=1), np.hypot(0.025, 0.025 * (np.sqrt(2) - 1)))
    # side by side with an offset in y: the overlap rectangle of the two faces
    p, n, d = oracle.box_box([0.049, 0.01, 0.025], I3, H, [0, 0, 0.025], I3, H, 0.005)
    assert len(p) == 4 and np.allclose(n, [1, 0, 0]) and np.allclose(d, -0.001)
    assert np.isclose(p[:, 1].min(), -0.015) and np.isclose(p[:, 1].max(), 0.025)
    # crossed edges: one point, midway, normal along the common perpendicular
    p, n, d = oracle.box_box([0, 0, 0.0695], rot(0, np.pi / 4), H, [0, 0, 0], rot(1, np.pi / 4), H, 0.005)
    assert len(p) == 1 and np.allclose(n, [0, 0, 1]) and np.isclose(d[0], 0.0695 - 2 * 0.025 * np.sqrt(2))
    assert np.allclose(p[0], [0, 0, 0.03475])
    # speculative point inside the margin, nothing beyond it
    p, n, d = oracle.box_box([0, 0, 0.078], I3, H, [0, 0, 0.025], I3, H, 0.005)
    assert len(p) == 4 and np.allclose(d, 0.003)
    assert len(oracle.box_box([0, 0, 0.081], I3, H, [0, 0, 0.025], I3, H, 0.005)[0]) == 0


def test_box_box_random_poses_are_consistent(oracle):
    """swapping the boxes flips the normal; the reported distance equals the separation along the normal for every
    returned point (support-function check), and separated pairs beyond the margin give nothing"""
    rng = np.random.default_rng(5)

    def support(pc, R, h, d):     # furthest extent of the box along d
        return pc @ d + np.abs(R.T @ d) @ h

    hits = 0
    for _ in range(400):
        RA = rot(0, rng.uniform(-3, 3)) @ rot(1, rng.uniform(-3, 3)) @ rot(2, rng.uniform(-3, 3))
        RB = rot(2, rng.uniform(-3, 3)) @ rot(0, rng.uniform(-3, 3))
        pA, pB = rng.uniform(-0.04, 0.04, 3), rng.uniform(-0.04, 0.04, 3)
        p, n, d = oracle.box_box(pA, RA, H, pB, RB, H, 0.005)
        p2, n2, d2 = oracle.box_box(pB, RB, H, pA, RA, H, 0.005)
        assert len(p) == len(p2) or min(len(p), len(p2)) > 0      # same verdict either way round
        if len(p) == 0:
            continue
        hits += 1
        assert np.isclose(np.linalg.norm(n), 1) and np.allclose(n, -n2, atol=1e-9)
        # separation of the two boxes along n (n points from B to A): min over A minus max over B
        sep = -support(pA, RA, H, -n) - support(pB, RB, H, n)
        assert d.min() >= sep - 1e-9 and d.min() <= sep + 0.02, (d, sep)
    assert hits > 100


def _place(oracle, env, cubes):
    s = env.get_state()
    s[0, oracle.ST_BP:oracle.ST_BP + 9] = np.asarray(cubes, dtype=float).reshape(9)
    env.set_state(s)


def test_stack_statics_tower_stands_and_succeeds(oracle):
    env = oracle.OracleStackTower(1, seed=3)
    _place(oracle, env, [[0.1, 0.05, 0.025], [0.1, 0.05, 0.075], [0.1, 0.05, 0.125]])
    s = env.get_state()
    s[0, oracle.ST_GOAL:oracle.ST_GOAL + 9] = s[0, oracle.ST_BP:oracle.ST_BP + 9]
    env.set_state(s)
    for _ in range(20):
        obs, ag, dg, r, done, succ = env.step(np.zeros((1, 8)))
    assert np.allclose(ag.reshape(3, 3)[:, 2], [0.025, 0.075, 0.125], atol=2e-4), ag     # SURVEY 8c statics
    assert np.allclose(ag.reshape(3, 3)[:, :2], [0.1, 0.05], atol=5e-4)
    assert r[0] == 0 and succ[0] == 1 and done[0] == 0
    assert np.abs(obs[0, 21:39]).max() < 1e-3                                              # cubes at rest


def test_stack_overlapping_spawn_is_pushed_apart(oracle):
    env = oracle.OracleStackTower(1, seed=3)
    _place(oracle, env, [[0.0, 0.0, 0.025], [0.03, 0.0, 0.025], [0.2, 0.1, 0.025]])     # 20 mm of overlap in x
    for _ in range(30):
        obs, ag, *_ = env.step(np.zeros((1, 8)))
    c = ag.reshape(3, 3)
    assert abs(c[1, 0] - c[0, 0]) > 0.0495 and np.all(np.abs(c[:, 2] - 0.025) < 1e-3)
    assert np.all(np.isfinite(obs))


def test_stack_obs_layout_and_episode_logic(oracle):
    env = oracle.OracleStackTower(3, seed=11)
    obs, ag, dg = env.reset()
    assert obs.shape == (3, 55) and ag.shape == (3, 9)
    assert np.array_equal(obs[:, 0:9], ag)                                  # :190 obj_pos first
    assert np.allclose(obs[:, 9:21].reshape(3, 3, 4), [0, 0, 0, 1])         # identity quaternions after the respawn tick
    assert np.allclose(dg.reshape(3, 3, 3)[:, :, 2], [0.025, 0.075, 0.125]) and np.allclose(dg[:, 0:2], dg[:, 3:5])   # :212-219
    assert np.all(np.abs(dg[:, 0]) <= 0.3) and np.all(np.abs(dg[:, 1]) <= 0.2)
    assert np.all(np.abs(ag.reshape(3, 3, 3)[:, :, 0]) <= 0.3 + 1e-3)
    # arm 2 mirrors arm 1 (bases at -+0.6, second yawed by pi): same pose, mirrored position
    assert np.allclose(obs[:, 39:42] * [-1, -1, 1], obs[:, 47:50], atol=1e-9)
    rng = np.random.default_rng(0)
    for k in range(50):
        obs, ag, dg, r, done, succ = env.step(rng.uniform(-1, 1, (3, 8)))
        assert np.all(done == (k == 49))                                    # only the 50-step limit ends an episode
    # same (seed, env id, episode) -> same spawn, independent of the batch
    solo = oracle.OracleStackTower(1, seed=11, env_id_offset=2)
    assert np.array_equal(solo.get_state()[0, oracle.ST_BP:oracle.ST_BP + 9], oracle.OracleStackTower(3, seed=11).get_state()[2, oracle.ST_BP:oracle.ST_BP + 9])


def test_stack_scripted_pick_and_stack(oracle):
    """arm 1 picks cube 0 and puts it on cube 1 - grasp friction, cube/cube support and release all have to work"""
    env = oracle.OracleStackTower(1, seed=1)
    _place(oracle, env, [[-0.2, 0, 0.025], [0.0, 0.1, 0.025], [0.2, -0.1, 0.025]])
    obs = env.step(np.zeros((1, 8)))[0]

    def servo(xy, z, g, n):
        nonlocal obs
        for _ in range(n):
            hp = obs[0, 39:42]
            a = np.zeros((1, 8))
            a[0, 0:2] = np.clip((np.asarray(xy) - hp[:2]) / 0.0625, -1, 1)
            a[0, 2] = np.clip((z - hp[2]) / 0.0625, -1, 1)
            a[0, 3] = g
            obs = env.step(a)[0]

    servo([-0.2, 0], 0.25, 1, 12)
    servo([-0.2, 0], 0.085, 1, 12)
    servo([-0.2, 0], 0.085, -1, 6)
    servo([-0.2, 0], 0.25, -1, 10)
    assert obs[0, 2] > 0.15, "cube 0 was not lifted"
    servo([0.0, 0.1], 0.25, -1, 14)
    servo([0.0, 0.1], 0.139, -1, 10)
    servo([0.0, 0.1], 0.139, 1, 6)
    servo([0.0, 0.1], 0.3, 1, 8)
    c = obs[0, 0:9].reshape(3, 3)
    assert abs(c[0, 2] - 0.075) < 1e-3 and abs(c[1, 2] - 0.025) < 1e-3, c            # cube 0 rests on cube 1
    assert np.linalg.norm(c[0, :2] - c[1, :2]) < 0.02
