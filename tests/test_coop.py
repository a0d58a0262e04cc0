"""The cooperative (16 lanes per environment, impulse-space) core of csrc/xarm_coop_core.h: the reset kernel that
takes the auto-reset off the step's critical path.  CPU part: the host instantiation (LV<T> = 16 explicit lane
values) against the oracle in float64 - the same Gauss-Seidel sweep in exact arithmetic - and in float32 within the
conditioned tolerance.  GPU part (-m gpu): k_reset_coop through the C ABI against the oracle's reset and against the
one-env-per-lane kernel."""
import numpy as np
import pytest
import torch


@pytest.mark.parametrize("key,seed", [("rand", 7), ("grasp", 1)])
def test_coop_step_f64_equals_oracle_on_golden_rollout(hostcore, golden_rollout, key, seed):
    g = golden_rollout
    S, A = g[key + "_states"], g[key + "_actions"]
    worst, n_ok = 0.0, 0
    for t in range(A.shape[0]):
        st, obs, ag, dg, rew, done, succ = hostcore.coop_step(S[t], A[t], f32=0, seed=seed)
        sens = g[key + "_sens"][t]
        err = np.abs(st - S[t + 1]).max(axis=1)
        ok = sens < 1e-2
        assert (err[ok] <= 1e-9 + 1e-4 * sens[ok]).all(), (t, err[ok].max())
        n_ok += ok.sum()
        np.testing.assert_allclose(obs[ok], g[key + "_obs"][t][ok], atol=1e-8)
        assert np.array_equal(rew[ok], g[key + "_rew"][t][ok])
        assert np.array_equal(done[ok], g[key + "_done"][t][ok])
        worst = max(worst, err[ok].max())
    assert worst < 1e-9
    assert n_ok >= 0.85 * A.shape[0] * A.shape[1]
    if key == "grasp":
        assert np.abs(S[:, :, 42:50]).max() > 0.1     # the pad rows (slot 2) really carried load


@pytest.mark.parametrize("key,seed", [("rand", 7), ("grasp", 1)])
def test_coop_step_f32_within_conditioned_tolerance(hostcore, golden_rollout, parity, key, seed):
    g = golden_rollout
    S, A = g[key + "_states"], g[key + "_actions"]
    for t in range(A.shape[0]):
        st, *_ = hostcore.coop_step(S[t], A[t], f32=1, seed=seed)
        parity.compare(st[:, parity.CONT], S[t + 1][:, parity.CONT], g[key + "_sens"][t], what="coop %s t=%d" % (key, t),
                       frac_tight=0.9 if key == "rand" else 0.75, max_exempt=0.15 if key == "rand" else 0.25)


def test_coop_reset_f64_equals_oracle(oracle, hostcore, golden_rollout):
    E = 16
    kw = dict(seed=21, ggr=0.5, igr=0.25)
    ora = oracle.OraclePnP(E, seed=21, goal_ground_rate=0.5, init_grasp_rate=0.25)
    st0 = hostcore.init(E, f32=0, **kw)
    o_obs, o_ag, o_dg = ora.reset()
    st, obs, ag, dg = hostcore.coop_reset(st0, f32=0, **kw)
    np.testing.assert_allclose(st, ora.state, atol=1e-10)
    np.testing.assert_allclose(obs, o_obs, atol=1e-10)
    # a reset out of a grasp (object between the pads while the arm is driven to the start pose); envs whose
    # reset is itself ill conditioned (the oracle's answer moves under a 1e-9 perturbation) are left out
    g = golden_rollout
    S = g["grasp_states"][-1]
    ora2 = oracle.OraclePnP(S.shape[0], seed=1)
    ora2.set_state(S)
    ora2.reset()
    ref = ora2.state.copy()
    from oracle import parity
    ora2.set_state(parity.perturb(S, np.random.default_rng(0), eps=1e-9))
    ora2.reset()
    calm = np.abs(ora2.state[:, :31] - ref[:, :31]).max(axis=1) < 1e-6
    assert calm.sum() >= 2 and (S[calm, 20] > 0.15).any()          # at least one really held object among them
    st2, *_ = hostcore.coop_reset(S, f32=0, seed=1)
    np.testing.assert_allclose(st2[calm], ref[calm], atol=1e-9)
    mask = np.zeros(E, np.uint8)
    mask[::3] = 1
    st3, *_ = hostcore.coop_reset(st, mask=mask, f32=0, **kw)
    assert np.array_equal(st3[mask == 0], st[mask == 0])
    assert (st3[mask == 1, 53] == 2).all() and (st3[mask == 1, 52] == 0).all()


def test_coop_row_sets_are_exact_subsets(hostcore, golden_rollout):
    """the host build picks the row set (pad rows / arm-limit rows present or not) per environment; an arm joint
    inside its limit window switches slot 3 on - the result must still match the one-env-per-lane core"""
    g = golden_rollout
    S = g["rand_states"][3].copy()
    S[:, 1] = 2.0          # joint 2 within 0.2 rad of its upper limit 2.0944
    S[:, 3] = -0.1         # joint 4 near its lower limit -0.19198
    A = g["rand_actions"][3]
    a, *_ = hostcore.coop_step(S, A, f32=0, seed=7)
    b, *_ = hostcore.step(S, A, f32=0, seed=7)
    np.testing.assert_allclose(a, b, atol=1e-9)


# ------------------------------------------------------------------------------------------- GPU
def _np(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def gx():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import gym_xarm_amd
    return gym_xarm_amd


@pytest.mark.gpu
def test_gpu_coop_reset_matches_oracle_and_lane_kernel(gx, oracle, parity):
    """masked resets out of a random rollout: k_reset_coop vs oracle.reset(mask) vs k_reset (same states)"""
    E = 512
    coop = gx.make("XarmPDPickAndPlace-v0", num_envs=E, seed=41, auto_reset=False)
    lane = gx.make("XarmPDPickAndPlace-v0", num_envs=E, seed=41, auto_reset=False, reset_coop_limit=-1)
    ora = oracle.OraclePnP(E, seed=41)
    coop.reset()
    gen = torch.Generator().manual_seed(5)
    for t in range(8):
        coop.step(torch.rand(E, 4, generator=gen) * 2 - 1)
    st0 = _np(coop.get_state()).astype(np.float64)
    lane.set_state(st0)
    ora.set_state(st0)
    for count in (1, 5, 64, 203):          # ragged counts: rows beyond the list shadow its last entry
        mask = np.zeros(E, np.uint8)
        mask[np.random.default_rng(count).choice(E, count, replace=False)] = 1
        before = _np(coop.get_state()).astype(np.float64)
        ora.set_state(before)
        lane.set_state(before)
        oc = coop.reset(mask=torch.tensor(mask))
        ol = lane.reset(mask=torch.tensor(mask))
        ora.reset(mask=mask)
        a, b = _np(coop.get_state()).astype(np.float64), _np(lane.get_state()).astype(np.float64)
        m = mask.astype(bool)
        assert np.array_equal(a[~m], before[~m].astype(np.float32).astype(np.float64))   # untouched rows
        assert (a[m, 52] == 0).all() and np.array_equal(a[m, 53], before[m, 53] + 1)
        # goals and spawn positions come from the counter RNG: exact
        np.testing.assert_allclose(a[m, 31:34], ora.state[m, 31:34], atol=1e-6)
        # the arm after six ticks; object away from the fingers = well conditioned
        calm = m & (np.abs(ora.state[:, 19]) > 0.09) & (np.abs(before[:, 19]) > 0.09)
        assert calm.sum() >= 0.25 * m.sum()
        np.testing.assert_allclose(a[calm, :31], ora.state[calm, :31], atol=2e-3)
        np.testing.assert_allclose(a[calm, :31], b[calm, :31], atol=2e-3)
        print("count %d: coop vs oracle %.2e, coop vs lane kernel %.2e" % (
            count, np.abs(a[calm, :31] - ora.state[calm, :31]).max(), np.abs(a[calm, :31] - b[calm, :31]).max()))
        np.testing.assert_allclose(_np(oc["observation"])[calm], _np(ol["observation"])[calm], atol=3e-3)
        np.testing.assert_allclose(_np(oc["desired_goal"])[m], ora.state[m, 31:34], atol=1e-6)
    coop.close()
    lane.close()


@pytest.mark.gpu
def test_gpu_reset_kernel_choice_by_count(gx):
    """counts up to reset_coop_limit run on the cooperative kernel, larger ones on the one-env-per-lane kernel;
    exactly one of the two does the work for any count (every selected env is reset exactly once)"""
    E = 256
    env = gx.make("XarmPDPickAndPlace-v0", num_envs=E, seed=3, auto_reset=False, reset_coop_limit=32)
    env.reset()                                   # 256 > 32: lane kernel
    s = _np(env.get_state())
    assert (s[:, 53] == 1).all()
    for count in (31, 32, 33, 256):
        m = torch.zeros(E, dtype=torch.uint8)
        m[:count] = 1
        before = _np(env.get_state())
        env.reset(mask=m)
        s = _np(env.get_state())
        assert np.array_equal(s[:count, 53], before[:count, 53] + 1) and np.array_equal(s[count:], before[count:])
        assert np.isfinite(s).all()
    env.close()


@pytest.mark.gpu
def test_gpu_coop_reset_is_deterministic_and_neighbour_independent(gx, golden_rollout):
    """an env's reset does not depend on which other envs share its wavefront: the row set (pad rows, arm-limit rows,
    which pads) is chosen per wavefront, absent rows carry exactly zero impulse and everything the row sets share
    rounds identically (the test that caught LLVM fusing a product into the lane reduction in one instantiation only).
    States: a random rollout, the grasp rollout (pads loaded) and arms pushed into their joint-limit windows."""
    E = 64
    env = gx.make("XarmPDPickAndPlace-v0", num_envs=E, seed=9, auto_reset=False)
    env.reset()
    gen = torch.Generator().manual_seed(1)
    for t in range(10):
        env.step(torch.rand(E, 4, generator=gen) * 2 - 1)
    rnd = _np(env.get_state())
    g = golden_rollout
    grasp = np.concatenate([g["grasp_states"][k] for k in range(6, 38, 2)])[:E].astype(np.float32)
    assert grasp.shape[0] == E
    mixed = rnd.copy()
    mixed[1::3] = grasp[1::3][:len(mixed[1::3])]
    mixed[2::7, 1] = 2.0            # joint 2 inside its upper limit window
    mixed[5::11, 3] = -0.1          # joint 4 inside its lower limit window
    for name, st0 in (("random", rnd), ("grasp", grasp), ("mixed", mixed)):
        st0 = torch.tensor(st0)
        env.set_state(st0)
        env.reset()                                   # all 64 through the cooperative kernel, 4 per wavefront
        full = env.get_state().clone()
        env.set_state(st0)
        env.reset()
        assert torch.equal(env.get_state(), full), name           # run-to-run
        for lo, n in ((0, 5), (3, 5), (17, 5), (20, 1), (30, 2), (41, 7)):
            env.set_state(st0)
            m = torch.zeros(E, dtype=torch.uint8)
            m[lo:lo + n] = 1
            env.reset(mask=m)
            part = env.get_state()
            assert torch.equal(part[lo:lo + n], full[lo:lo + n]), (name, lo, n)
    env.close()
