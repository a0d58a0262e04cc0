"""XarmPDStackTower-v0: the kernel core (gym_xarm_amd/csrc/xarm_stack_core.h) against the CPU oracle - host (g++)
instantiation in float64/float32 on the CPU, the HIP path through the C-ABI on the GPU.  Physics parity against
PyBullet is UNPINNED (PyBullet absent); the reward arithmetic is pinned by the reference's NumPy code
(tests/golden/stack_reward_reference.npz)."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CONT = np.r_[0:36, 54:93]          # q, qd of both arms, cube poses and velocities
POS = np.r_[0:18, 54:75]           # joint positions, cube positions and quaternions


def rot(ax, a):
    c, s = np.cos(a), np.sin(a)
    return {0: np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), 1: np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
            2: np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[ax]


def perturb(state, rng, eps=1e-6):
    s = np.array(state, dtype=np.float64, copy=True)
    s[:, CONT] += rng.uniform(-eps, eps, size=(s.shape[0], CONT.size))
    for o in range(3):
        q = s[:, 63 + 4 * o:67 + 4 * o]
        s[:, 63 + 4 * o:67 + 4 * o] = q / np.linalg.norm(q, axis=1, keepdims=True)
    return s


def oracle_step_with_sens(ora, state, actions, seed=0):
    rng = np.random.default_rng(seed)
    ora.set_state(state)
    out = ora.step(actions)
    nxt = ora.get_state()
    sens = np.zeros(state.shape[0])
    for _ in range(2):
        ora.set_state(perturb(state, rng))
        ora.step(actions)
        sens = np.maximum(sens, np.abs(ora.get_state()[:, CONT] - nxt[:, CONT]).max(axis=1))
    ora.set_state(nxt)
    return (nxt,) + tuple(out) + (sens,)


@pytest.fixture(scope="module")
def host():
    from conftest import HostCore
    return HostCore()


def test_cube_cube_host_matches_oracle(oracle, host):
    rng = np.random.default_rng(5)
    h3 = np.full(3, 0.025)
    hits = 0
    for i in range(600):
        if i % 3 == 0:   # yaw-only poses: the face/face configurations of cubes lying on a table or on each other
            RA, RB = rot(2, rng.uniform(-3, 3)), rot(2, rng.uniform(-3, 3))
        else:
            RA = rot(0, rng.uniform(-3, 3)) @ rot(1, rng.uniform(-3, 3)) @ rot(2, rng.uniform(-3, 3))
            RB = rot(2, rng.uniform(-3, 3)) @ rot(0, rng.uniform(-3, 3))
        pA, pB = rng.uniform(-0.04, 0.04, 3), rng.uniform(-0.04, 0.04, 3)
        p, n, d = oracle.box_box(pA, RA, h3, pB, RB, h3, 0.005)
        p2, n2, d2 = host.cube_cube(pA, RA, pB, RB, f32=0)
        assert len(p) == len(p2)
        if len(p):
            hits += 1
            assert np.allclose(p, p2, atol=1e-12) and np.allclose(n, n2, atol=1e-12) and np.allclose(d, d2, atol=1e-12)
            p3, n3, d3 = host.cube_cube(pA, RA, pB, RB, f32=1)
            if len(p3) == len(p) and np.allclose(n, n3, atol=1e-3):   # float32 may break a tie the other way
                assert abs(d3.min() - d.min()) < 1e-5
    assert hits > 400


def test_stack_host_f64_matches_oracle(oracle, host):
    E = 2
    ora = oracle.OracleStackTower(E, seed=4)
    st = host.st_init(E, f32=0, seed=4)
    assert np.array_equal(st, ora.get_state())
    st, obs, ag, dg = host.st_reset(st, f32=0, seed=4)
    o2, a2, d2 = ora.reset()
    assert np.abs(st - ora.get_state()).max() < 1e-12 and np.abs(obs - o2).max() < 1e-12 and np.array_equal(dg, d2)
    rng = np.random.default_rng(1)
    for _ in range(2):
        act = rng.uniform(-1.2, 1.2, (E, 8))
        st, obs, ag, dg, rew, done, succ = host.st_step(st, act, f32=0, seed=4)
        o2, a2, d2, r2, dn2, s2 = ora.step(act)
        assert np.abs(st - ora.get_state()).max() < 1e-10 and np.abs(obs - o2).max() < 1e-10
        assert np.array_equal(rew, r2) and np.array_equal(done, dn2) and np.array_equal(succ, s2)


def test_stack_host_f64_contact_scenarios(oracle, host):
    """overlapping spawn (cube/cube rows), a standing tower, and a cube inside a closing gripper"""
    ora = oracle.OracleStackTower(3, seed=2)
    s = ora.get_state()
    s[0, 54:63] = [0.0, 0.0, 0.025, 0.03, 0.004, 0.025, 0.2, 0.1, 0.025]
    s[1, 54:63] = [0.1, 0.05, 0.025, 0.1, 0.05, 0.075, 0.103, 0.048, 0.125]
    s[1, 67:71] = [0, 0, np.sin(0.3), np.cos(0.3)]                          # middle cube yawed
    ora.set_state(s)
    ora.step(np.zeros((3, 8)))
    # env 2: put cube 0 between the pads of arm 1
    obs = ora.step(np.zeros((3, 8)))[0]
    s = ora.get_state()
    s[2, 54:57] = obs[2, 39:42] - [0, 0, 0.067]
    s[2, 75:93] = 0
    ora.set_state(s)
    st = ora.get_state()
    act = np.zeros((3, 8))
    act[2, 3] = -1
    for _ in range(2):
        st, obs, ag, dg, rew, done, succ = host.st_step(st, act, f32=0, seed=2)
        o2 = ora.step(act)[0]
        assert np.abs(st - ora.get_state()).max() < 1e-9, np.abs(st - ora.get_state()).max(0).argmax()
    assert st[2, 126:130].max() > 0, "the gripper scenario produced no pad contact"
    assert abs(st[1, 56 + 6] - 0.125) < 2e-3          # the tower still stands


def test_stack_host_f64_both_arms_on_one_cube(oracle, host):
    """both grippers closed on the same cube: the finger phases of the two arms must be swept one after the other
    with the cube velocity handed over in between (the sequential path); elsewhere they commute and run at once"""
    q = np.array([-0.009068751632859924, -0.08153217279952825, 0.09299669711139864, 1.067692645248743,
                  0.0004018824370178429, 1.1524205092196147, -0.0004991403332530034, 0.04, 0.04])
    for _ in range(40):   # hand of arm 1 to the point midway between the bases; arm 2 (mirrored) then sits at the same point
        q = oracle.ik(q, [0.6, 0.0, 0.2], 15)[:9]
        q[7:] = 0.04
    ora = oracle.OracleStackTower(2, seed=5)
    s = ora.get_state()
    s[:, 0:9] = q; s[:, 9:18] = q
    s[:, 36:54] = s[:, 0:18]
    ora.set_state(s)
    obs = ora.step(np.zeros((2, 8)))[0]
    assert np.abs(obs[:, 39:42] - obs[:, 47:50]).max() < 2e-3          # the two hands coincide
    s = ora.get_state()
    s[0, 54:57] = obs[0, 39:42] - [0, 0, 0.067]                          # env 0: the cube between both pairs of pads
    s[1, 54:57] = [0.2, 0.1, 0.025]                                      # env 1: nothing to grasp
    s[:, 75:93] = 0
    ora.set_state(s)
    st = ora.get_state()
    act = np.zeros((2, 8))
    act[:, 3] = act[:, 7] = -1
    for _ in range(2):
        st, *_ = host.st_step(st, act, f32=0, seed=5)
        ora.step(act)
        assert np.abs(st - ora.get_state()).max() < 1e-9
    assert st[0, 126:130].max() > 0 and st[0, 130:134].max() > 0, "both arms must hold the cube"
    assert st[1, 126:134].max() == 0


def test_stack_host_f32_close_to_oracle(oracle, host):
    E = 4
    ora = oracle.OracleStackTower(E, seed=9)
    st = ora.get_state()
    rng = np.random.default_rng(3)
    act = rng.uniform(-1, 1, (E, 8))
    nxt, o2, a2, d2, r2, dn2, s2, sens = oracle_step_with_sens(ora, st, act)
    st32, obs, ag, dg, rew, done, succ = host.st_step(st, act, f32=1, seed=9)
    err = np.abs(st32[:, POS] - nxt[:, POS]).max(axis=1)
    assert np.all(err < 5e-4 + 300 * sens), (err, sens)
    assert np.array_equal(rew, r2)


# ------------------------------------------------------------------------------------------------ GPU
def _make(E, **kw):
    import gym_xarm_amd
    return gym_xarm_amd.make("XarmPDStackTower-v0", num_envs=E, **kw)


@pytest.mark.gpu
def test_stack_gpu_dims_and_reset_parity(oracle):
    import torch
    E = 128
    env = _make(E, seed=4, auto_reset=False)
    assert (env.obs_dim, env.goal_dim, env.act_dim, env.state_dim, env.max_episode_steps) == (55, 9, 8, 136, 50)
    assert env.action_space.shape == (8,) and env.observation_space["observation"].shape == (55,)
    ora = oracle.OracleStackTower(E, seed=4)
    assert np.abs(env.get_state().cpu().numpy() - ora.get_state()).max() < 1e-6
    obs = env.reset()
    o2, a2, d2 = ora.reset()
    sd = np.abs(env.get_state().cpu().numpy() - ora.get_state())
    assert sd[:, POS].max() < 5e-4, sd[:, POS].max()
    assert np.abs(obs["desired_goal"].cpu().numpy() - d2).max() < 1e-6
    assert np.abs(obs["achieved_goal"].cpu().numpy() - a2).max() < 5e-4
    assert torch.equal(obs["observation"][:, 0:9], obs["achieved_goal"])
    env.close()


@pytest.mark.gpu
def test_stack_gpu_step_parity_with_sensitivity(oracle, parity):
    import torch
    E = 256
    env = _make(E, seed=6, auto_reset=False)
    env.reset()
    ora = oracle.OracleStackTower(E, seed=6)
    rng = np.random.default_rng(2)
    # spread a few envs into contact-rich states: overlapping cubes, a tower, a cube under a gripper
    s = env.get_state().cpu().numpy().astype(np.float64)
    s[0:32, 57:60] = s[0:32, 54:57] + [0.03, 0.004, 0.0]
    s[32:64, 57:60] = s[32:64, 54:57] + [0.002, -0.001, 0.05]
    s[32:64, 60:63] = s[32:64, 54:57] + [0.0, 0.002, 0.10]
    env.set_state(torch.tensor(s, dtype=torch.float32))
    worst = {}
    for k in range(6):
        st = env.get_state().cpu().numpy().astype(np.float64)
        act = rng.uniform(-1, 1, (E, 8)).astype(np.float32)
        obs, rew, done, info = env.step(torch.from_numpy(act))
        nxt, o2, a2, d2, r2, dn2, s2, sens = oracle_step_with_sens(ora, st, act.astype(np.float64), seed=k)
        dev = env.get_state().cpu().numpy()
        stats = parity.compare(dev[:, CONT], nxt[:, CONT], sens, frac_tight=0.7, max_exempt=0.15, what="StackTower state step %d" % k)
        parity.compare(obs["observation"].cpu().numpy(), o2, sens, frac_tight=0.7, max_exempt=0.15, what="StackTower obs step %d" % k)
        ok = sens < 1e-6
        assert np.array_equal(rew.cpu().numpy()[ok], r2[ok].astype(np.float32))
        assert np.array_equal(done.cpu().numpy(), dn2)
        worst = stats
    assert worst["frac_tight"] >= 0.7
    env.close()


@pytest.mark.gpu
def test_stack_gpu_scripted_pick_and_stack(oracle):
    """the behaviour the env exists for, on the HIP path: arm 1 picks cube 0 and puts it on cube 1"""
    import torch
    env = _make(1, seed=1, auto_reset=False)
    s = env.get_state()
    s[0, 54:63] = torch.tensor([-0.2, 0, 0.025, 0.0, 0.1, 0.025, 0.2, -0.1, 0.025], device=env.device)
    env.set_state(s)
    o = env.step(torch.zeros(1, 8))[0]["observation"].cpu().numpy()

    def servo(xy, z, g, n):
        nonlocal o
        for _ in range(n):
            hp = o[0, 39:42]
            a = np.zeros((1, 8), np.float32)
            a[0, 0:2] = np.clip((np.asarray(xy) - hp[:2]) / 0.0625, -1, 1)
            a[0, 2] = np.clip((z - hp[2]) / 0.0625, -1, 1)
            a[0, 3] = g
            o = env.step(torch.from_numpy(a))[0]["observation"].cpu().numpy()

    servo([-0.2, 0], 0.25, 1, 12); servo([-0.2, 0], 0.085, 1, 12); servo([-0.2, 0], 0.085, -1, 6); servo([-0.2, 0], 0.25, -1, 10)
    assert o[0, 2] > 0.15, "cube 0 was not lifted"
    servo([0.0, 0.1], 0.25, -1, 14); servo([0.0, 0.1], 0.139, -1, 10); servo([0.0, 0.1], 0.139, 1, 6); servo([0.0, 0.1], 0.3, 1, 8)
    c = o[0, 0:9].reshape(3, 3)
    assert abs(c[0, 2] - 0.075) < 1e-3 and abs(c[1, 2] - 0.025) < 1e-3 and np.linalg.norm(c[0, :2] - c[1, :2]) < 0.02, c
    env.close()


@pytest.mark.gpu
def test_stack_gpu_reward_kernel_matches_reference_numpy():
    import torch
    d = np.load(os.path.join(GOLDEN, "stack_reward_reference.npz"))
    ag, g = torch.tensor(d["achieved_goal"], dtype=torch.float32), torch.tensor(d["goal"], dtype=torch.float32)
    dist = np.linalg.norm(d["achieved_goal"] - d["goal"], axis=1)
    clear = np.abs(dist - 0.09) > 1e-5          # float32 vs float64 exactly on the threshold shell
    for rt in ("sparse", "dense"):
        env = _make(4, config={"reward_type": rt})
        out = env.compute_reward(ag, g).cpu().numpy()
        if rt == "sparse":
            assert np.array_equal(out[clear], d["reward_sparse"][clear])          # xarm_stack_tower.py:124-127
        else:
            assert np.allclose(out, d["reward_dense"], atol=1e-6)                 # :128-129
        env.close()


@pytest.mark.gpu
def test_stack_gpu_time_limit_autoreset_and_sharding():
    import torch
    E = 96
    env = _make(E, seed=21)
    obs = env.reset()
    g0 = obs["desired_goal"].clone()
    gen = torch.Generator(device=env.device)
    gen.manual_seed(0)
    for k in range(50):
        obs, rew, done, info = env.step(torch.rand(E, 8, device=env.device, generator=gen) * 2 - 1)
        assert bool((done != 0).all()) == (k == 49) and bool((done != 0).any()) == (k == 49)
        assert torch.isfinite(obs["observation"]).all()
    assert bool(info["TimeLimit.truncated"].all())
    assert int(env.episode_steps().max()) == 0                     # auto-reset happened inside the call
    assert not torch.equal(obs["desired_goal"], g0)                # a fresh tower goal per episode
    # freshly respawned cubes are upright unless two of them spawned overlapping (no rejection sampling, :207-209)
    upright = (obs["observation"][:, 9:21].reshape(E, 3, 4) - torch.tensor([0., 0., 0., 1.], device=env.device)).abs().amax(dim=(1, 2)) < 1e-3
    assert float(upright.float().mean()) > 0.8
    term = info["terminal_observation"]
    assert not torch.allclose(term[:, 0:9], obs["observation"][:, 0:9])
    # the same global env ids on a shard give the same episodes
    shard = _make(32, seed=21, env_id_offset=64)
    full = _make(E, seed=21)
    a, b = shard.reset(), full.reset()
    assert torch.equal(a["desired_goal"], b["desired_goal"][64:96]) and torch.allclose(a["observation"], b["observation"][64:96], atol=0)
    for e in (env, shard, full):
        e.close()


@pytest.mark.gpu
def test_stack_gpu_rollout_statistics_match_oracle(oracle):
    """long-horizon parity on distributions (trajectories diverge past contact onset): 1024 envs x 25 steps with the
    arms biased towards the table, HIP vs oracle from the same seeds"""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    E, T, W = 1024, 25, 16
    env = _make(E, seed=77, auto_reset=False)
    obs0 = env.reset()
    # cubes 0 and 1 are moved under the two hands, so that hands coming down meet them
    s0 = env.get_state()
    s0[:, 54:56] = obs0["observation"][:, 39:41]; s0[:, 57:59] = obs0["observation"][:, 47:49]
    env.set_state(s0)
    start = s0.cpu().numpy().astype(np.float64)
    acts = []
    for t in range(T):
        a = torch.rand(E, 8, generator=torch.Generator().manual_seed(500 + t)) * 2 - 1
        a[:, 0:2] *= 0.3; a[:, 4:6] *= 0.3
        a[:, 2] -= 0.7; a[:, 6] -= 0.7                 # push both hands down to cube height
        acts.append(a.clamp(-1, 1))
    for t in range(T):
        env.step(acts[t])
    dev = env.get_state().cpu().numpy().astype(np.float64)
    env.close()
    shards = [oracle.OracleStackTower(E // W, seed=77, env_id_offset=k * (E // W)) for k in range(W)]
    a_np = [a.numpy().astype(np.float64) for a in acts]

    def run(k):
        sh = shards[k]
        sh.reset()
        sh.set_state(start[k * (E // W):(k + 1) * (E // W)])      # identical start (the reset itself is checked elsewhere)
        for t in range(T):
            sh.step(a_np[t][k * (E // W):(k + 1) * (E // W)])
        return sh.state
    with ThreadPoolExecutor(W) as ex:
        ora = np.concatenate(list(ex.map(run, range(W))))

    def stats(s):
        cz = s[:, [56, 59, 62]]
        cxy = s[:, [54, 55, 57, 58, 60, 61]]
        moved = np.abs(s[:, 75:93]).max(axis=1) > 1e-2              # some cube is moving
        return np.array([(np.abs(cz - 0.025) < 2e-3).mean(), (cz > 0.03).mean(), moved.mean(), (s[:, 126:134] > 0).any(axis=1).mean(),
                         np.median(np.abs(cxy)), s[:, 7].mean(), s[:, 16].mean(), np.median(s[:, [0, 9]])])
    sd, so = stats(dev), stats(ora)
    tol = np.array([0.03, 0.03, 0.05, 0.05, 0.01, 0.002, 0.002, 0.02])
    assert (np.abs(sd - so) <= tol).all(), (sd, so)
    assert so[3] > 0.02, "the biased actions should bring pads into contact with cubes in a visible share of envs"
    # envs whose cubes were never touched agree closely in the arm joints
    calm = (np.abs(ora[:, 75:93]).max(axis=1) < 1e-6) & (np.abs(dev[:, 75:93]).max(axis=1) < 1e-4)
    assert calm.mean() > 0.2
    assert np.median(np.abs(dev[calm, 0:18] - ora[calm, 0:18]).max(axis=1)) < 2e-3


@pytest.mark.gpu
def test_stack_gpu_full_size_properties_8192():
    """BASELINE config 4's per-GPU shard (65 536 envs over 8 GPUs = 8 192: 256 wavefronts holding 151 KB of LDS each,
    one per CU) - size-independent properties: determinism, shard invariance, state invariants, auto-reset."""
    import torch
    E = 8192
    a = [torch.rand(E, 8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(k)) * 2 - 1 for k in range(4)]

    def run(n, off):
        env = _make(n, seed=9, env_id_offset=off)
        env.reset()
        s = env.get_state()
        s[: n // 4, 134] = 48                                   # a quarter of the shard hits the 50-step limit in call 2
        env.set_state(s)
        for k in range(4):
            obs, rew, done, info = env.step(a[k][off:off + n])
        out = env.get_state().clone(), obs["observation"].clone(), rew.clone(), done.clone(), env.episode_steps().clone()
        env.close()
        return out
    full, again = run(E, 0), run(E, 0)
    for x, y in zip(full, again):
        assert torch.equal(x, y)                                # deterministic, bitwise
    half = run(E // 2, E // 2)
    st, steps = full[0], full[4]
    quarter = E // 4
    # the second half of the batch carries no manipulated counters in either run: bitwise world-size invariant
    assert torch.equal(st[E // 2 + E // 8:], half[0][E // 8:]) and torch.equal(full[1][E // 2 + E // 8:], half[1][E // 8:])
    assert torch.isfinite(st).all()
    q = st[:, 63:75].reshape(E, 3, 4)
    assert float((q.norm(dim=2) - 1).abs().max()) < 1e-5                              # cube quaternions
    assert bool((st[:, 102:134] >= 0).all())                                          # normal impulses never pull
    assert bool((steps[:quarter] == 2).all()) and bool((steps[quarter:] == 4).all())  # reset inside call 2, then 2 more steps
    assert bool(((full[2] == 0) | (full[2] == -1)).all())                             # sparse reward (:124-127)


def test_class_order_is_a_permutation_with_one_class_per_wavefront(hostcore):
    """xs::class_layout / class_slot (the visiting order of k_st_step): every env appears once; while class 0 can fill
    the holes each wavefront-sized group holds at most ONE class other than 0; without enough class-0 envs the order
    falls back to plain contiguous classes - still a permutation"""
    rng = np.random.default_rng(0)
    for n, p0 in ((8192, 0.88), (1000, 0.9), (37, 0.5), (1, 1.0), (64, 0.0), (5000, 0.05), (4096, 1.0)):
        key = np.where(rng.random(n) < p0, 0, rng.choice([1, 2, 4, 9, 17, 3, 24, 31], n)).astype(np.uint8)
        order, aligned = hostcore.class_order(key, 32)
        assert np.array_equal(np.sort(order), np.arange(n))
        ko = key[order]
        holes = sum((-int((key == c).sum())) % 32 for c in range(1, 32) if (key == c).any())
        assert aligned == (holes <= int((key == 0).sum()))
        if aligned:
            for g in range(0, n, 32):
                assert len(set(ko[g:g + 32].tolist()) - {0}) <= 1, "group %d mixes classes" % (g // 32)
        else:
            nz = ko[ko != 0]
            assert np.all(np.diff(nz.astype(int)) >= 0) and np.all(ko[:nz.size] != 0)


@pytest.mark.gpu
def test_gpu_class_order_changes_nothing_but_the_time():
    """k_st_step visits the envs grouped by row-set class (xarm_stack_core.h class_layout) unless XARM_ST_CLASS_ORDER=0:
    states, observations, rewards and done flags are bitwise those of the plain order, through resets"""
    import torch
    import gym_xarm_amd
    E = 3000
    outs = []
    for flag in ("1", "0"):
        os.environ["XARM_ST_CLASS_ORDER"] = flag
        try:
            env = gym_xarm_amd.make("XarmPDStackTower-v0", num_envs=E, seed=4)
        finally:
            del os.environ["XARM_ST_CLASS_ORDER"]
        env.reset()
        s = env.get_state()
        # a third of the envs with cubes side by side / stacked, so that several classes exist from the first step on
        s[: E // 6, 54:63] = torch.tensor([-0.05, 0.0, 0.025, 0.0, 0.0, 0.025, 0.05, 0.0, 0.025], device=env.device)
        s[E // 6: E // 3, 54:63] = torch.tensor([0.0, 0.1, 0.025, 0.0, 0.1, 0.075, 0.2, 0.0, 0.025], device=env.device)
        s[: E // 3, 63:75] = torch.tensor([0.0, 0.0, 0.0, 1.0] * 3, device=env.device)
        s[: E // 3, 75:93] = 0
        env.set_state(s)
        env.set_episode_steps(torch.arange(E, device=env.device) % 100)     # time-limit resets inside the run
        gen = torch.Generator(device=env.device)
        gen.manual_seed(2)
        rec = []
        for _ in range(8):
            a = torch.rand(E, 8, device=env.device, generator=gen) * 2 - 1
            obs, rew, done, info = env.step(a)
            rec.append(torch.cat([obs["observation"], obs["achieved_goal"], obs["desired_goal"], rew[:, None], done[:, None].float(),
                                  env.get_state()], dim=1).clone())
        outs.append(torch.stack(rec))
        env.close()
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])
