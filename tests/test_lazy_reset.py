"""Lazy auto-reset (opt-in, include/xarm_hip.h XARM_AUTO_RESET_LAZY): a finished env runs the reference's six reset
ticks (xarm_pick_and_place.py:250-266) one per step call.  The tick sequence is that of the strict reset, so the state
the new episode starts from must equal the strict reset's."""
import numpy as np
import pytest


def test_host_lazy_reset_reproduces_the_oracle_reset(oracle):
    from conftest import HostCore
    H = HostCore()
    E = 6
    ora = oracle.OraclePnP(E, seed=9)
    ora.reset()
    s = ora.get_state()
    s[:3, 52] = 48                                   # three envs reach the 50-step limit in two steps
    ora.set_state(s)
    st = s.copy()
    rng = np.random.default_rng(0)
    expect, checked = {}, 0
    finished = np.zeros(E, bool)
    for k in range(10):
        act = rng.uniform(-1, 1, (E, 4))
        done = ora.step(act)[4].astype(bool)
        finished |= done
        if done.any():
            ora.reset(mask=done.astype(np.uint8))
            for e in np.where(done)[0]:
                expect[e] = (k + 6, ora.get_state()[e].copy())
        st, obs, ag, dg, rew, phase, succ = H.step_lazy(st, act, f32=0, seed=9)
        assert np.array_equal(phase[:3], [0, 0, 0] if k < 1 else ([1] * 3 if k == 1 else ([2] * 3 if k <= 7 else [0] * 3)))
        finished |= phase != 0
        assert np.all(rew[phase == 2] == 0)
        for e, (kk, post) in list(expect.items()):
            if k == kk:
                assert np.abs(st[e] - post).max() < 1e-12, (e, np.abs(st[e] - post).max())
                checked += 1
                del expect[e]
    assert checked == 3
    # envs that never finished are untouched by the mode: identical to the strict path
    assert (~finished).any() and np.abs(st[~finished] - ora.get_state()[~finished]).max() < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("coop", [False, True])
def test_gpu_lazy_reset_equals_strict_reset_and_masks(coop):
    """coop=False: the strict reset on the one-env-per-lane kernel runs the lazy mode's instruction stream -> same
    bits.  coop=True (the default strict path, k_reset_coop): a different but algebraically equal sweep -> equal
    within the float32 tolerance on well-conditioned resets."""
    import torch
    import gym_xarm_amd
    E, K = 256, 14
    strict = gym_xarm_amd.make("XarmPDPickAndPlace-v0", num_envs=E, seed=4, auto_reset=True, reset_coop_limit=0 if coop else -1,
                                step_coop_limit=-1)       # the lazy mode steps one env per lane: same step kernel family
    lazy = gym_xarm_amd.make("XarmPDPickAndPlace-v0", num_envs=E, seed=4, auto_reset="lazy")
    strict.reset(); lazy.reset()
    s = strict.get_state(); s[:64, 52] = 47; strict.set_state(s); lazy.set_state(s)     # 64 envs hit the step limit at call 2
    gen = torch.Generator(device=strict.device); gen.manual_seed(0)
    first = torch.full((E,), -1, device=strict.device)                                   # call at which the env first finished
    post = torch.zeros(E, strict.state_dim, device=strict.device)
    checked = 0
    for k in range(K):
        a = torch.rand(E, 4, device=strict.device, generator=gen) * 2 - 1
        o1, r1, d1, i1 = strict.step(a)
        o2, r2, d2, i2 = lazy.step(a)
        never = first == -1
        fresh = (d1 != 0) & never
        # up to its first finish an env takes identical steps in both modes: same flag, same last observation
        assert torch.equal((d2 != 0)[never], (d1 != 0)[never])
        assert torch.equal(i2["terminal_observation"][fresh], i1["terminal_observation"][fresh])
        post[fresh] = strict.get_state()[fresh]                                           # strict: already reset here
        first[fresh] = k
        in_reset = (first >= 0) & (k > first) & (k <= first + 6)
        tracked = first != -2
        assert torch.equal(i2["resetting"][tracked], in_reset[tracked])                      # six masked calls after the finish
        assert bool((r2[in_reset] == 0).all()) and not bool(d2[in_reset].any())
        complete = (first >= 0) & (k == first + 6)
        if bool(complete.any()):
            if not coop:
                assert torch.equal(lazy.get_state()[complete], post[complete])            # same ticks, same bits
            else:
                a_, b_ = lazy.get_state()[complete], post[complete]
                assert torch.equal(a_[:, 31:34], b_[:, 31:34]) and torch.equal(a_[:, 52:], b_[:, 52:])   # goal, counters
                err = (a_[:, :31] - b_[:, :31]).abs().max(dim=1).values
                # the object spawns 15 mm inside the table and, for some envs, inside the closing fingers: those
                # resets are chaotic (oracle/parity.py); the rest agree to float32 rounding through six ticks
                assert float(err.median()) < 1e-4 and float((err < 2e-3).float().mean()) > 0.7, (err.median(), (err < 2e-3).float().mean())
            checked += int(complete.sum())
        first[complete] = -2                                                              # afterwards the two modes are out of phase
    assert checked >= 64
    quiet = first == -1
    assert int(quiet.sum()) > 100 and torch.equal(strict.get_state()[quiet], lazy.get_state()[quiet])
    strict.close(); lazy.close()


@pytest.mark.gpu
def test_gpu_lazy_mode_is_rejected_where_not_built():
    import gym_xarm_amd
    from gym_xarm_amd._native import XarmNativeError
    with pytest.raises(XarmNativeError):
        gym_xarm_amd.make("XarmReach-v0", num_envs=8, auto_reset="lazy")
