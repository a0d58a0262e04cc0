#!/usr/bin/env python3
"""Generate tests/golden/pnp_oracle_rollout.npz with the CPU oracle (float64): a short random
rollout of 32 envs plus a scripted reach-grasp-lift of 4 envs, with per-step sensitivity estimates
(oracle/parity.py).  GPU parity tests replay every recorded transition from its recorded state."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle import parity as PR  # noqa: E402


def scripted_actions(ora, phase_len=(8, 6, 8, 6, 8)):
    """closed-loop script per env: settle, move above the object, descend, close, lift"""
    E = ora.E
    acts = []
    tot = sum(phase_len)
    box0 = None
    for t in range(tot):
        st = ora.get_state()
        eef = np.stack([O.fk(st[e, :9])[0][7] for e in range(E)])
        if t == phase_len[0]:
            box0 = st[:, 18:21].copy()
        ph = np.searchsorted(np.cumsum(phase_len), t, side="right")
        if ph == 0:
            tgt, grip = eef, 1.0
        else:
            z = {1: 0.25, 2: 0.15, 3: 0.15, 4: 0.35}[ph]
            tgt = np.column_stack([box0[:, 0], box0[:, 1], np.full(E, z)])
            grip = 1.0 if ph < 3 else -1.0
        a = np.zeros((E, 4))
        a[:, :3] = np.clip((tgt - eef) / 0.0625, -1, 1)
        a[:, 3] = grip
        acts.append(a)
        ora.step(a)
    return np.stack(acts)


class JitteredGrasp:
    """The scripted reach-grasp-lift above as a closed-loop policy on STATES (oracle or device), with per-env jitter
    of the phase lengths, the approach offset and the descent height, so that a batch covers the contact onset, the
    closing fingers and the loaded lift at many different phases at once (tests/test_gpu_parity.py contact regime)."""

    def __init__(self, E, seed=0):
        rng = np.random.default_rng(seed)
        self.settle = rng.integers(5, 10, E)
        self.above = self.settle + rng.integers(5, 8, E)
        self.down = self.above + rng.integers(7, 10, E)
        self.close = self.down + rng.integers(5, 8, E)
        self.dxy = rng.uniform(-0.004, 0.004, (E, 2))
        self.zdown = rng.uniform(0.146, 0.154, E)
        self.box0 = None
        self.horizon = int(self.close.max()) + 9

    def __call__(self, st, t):
        E = st.shape[0]
        eef = np.stack([O.fk(st[e, :9])[0][7] for e in range(E)])
        if self.box0 is None:
            self.box0 = np.zeros((E, 3))
        latch = t == self.settle
        self.box0[latch] = st[latch, 18:21]
        ph = (t >= self.settle).astype(int) + (t >= self.above) + (t >= self.down) + (t >= self.close)
        z = np.choose(ph, [0.25 * np.ones(E), 0.25 * np.ones(E), self.zdown, self.zdown, 0.35 * np.ones(E)])
        tgt = np.column_stack([self.box0[:, 0] + self.dxy[:, 0], self.box0[:, 1] + self.dxy[:, 1], z])
        tgt[ph == 0] = eef[ph == 0]
        a = np.zeros((E, 4))
        a[:, :3] = np.clip((tgt - eef) / 0.0625, -1, 1)
        a[:, 3] = np.where(ph < 3, 1.0, -1.0)
        return a


def record(ora, actions):
    states, outs, sens = [ora.get_state()], [], []
    for t in range(actions.shape[0]):
        r = PR.oracle_step_with_sens(ora, states[-1], actions[t], seed=t)
        states.append(r[0])
        outs.append(r[1:7])
        sens.append(r[7])
    return (np.stack(states), np.stack([o[0] for o in outs]), np.stack([o[3] for o in outs]),
            np.stack([o[4] for o in outs]), np.stack([o[5] for o in outs]), np.stack(sens))


def main():
    out = {}
    # random rollout
    ora = O.OraclePnP(32, seed=7)
    s_init = ora.get_state()
    obs0, ag0, dg0 = ora.reset()
    out["rand_init_state"] = s_init
    out["rand_reset_obs"] = obs0
    rng = np.random.default_rng(11)
    acts = rng.uniform(-1, 1, size=(10, 32, 4))
    acts[3] *= 3.0  # exercises the action clip (:201)
    st, obs, rew, done, succ, sens = record(ora, acts)
    out.update(rand_actions=acts, rand_states=st, rand_obs=obs, rand_rew=rew, rand_done=done, rand_succ=succ, rand_sens=sens)
    # scripted grasp (4 envs, object away from the start pose so that the grasp is clean)
    ora = O.OraclePnP(4, seed=1)
    ora.reset()
    start = ora.get_state()
    acts = scripted_actions(ora)
    ora.set_state(start)
    st, obs, rew, done, succ, sens = record(ora, acts)
    out.update(grasp_actions=acts, grasp_states=st, grasp_obs=obs, grasp_rew=rew, grasp_done=done, grasp_succ=succ, grasp_sens=sens)
    path = os.path.join(ROOT, "tests", "golden", "pnp_oracle_rollout.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "lifted z:", st[-1][:, 20], "max sens rand %.2e grasp %.2e" % (out["rand_sens"].max(), sens.max()))


def reach():
    """XarmReach-v0: random rollout over a full episode (25 steps + the step after) of 32 envs, with the
    oracle's sensitivity of the OBSERVABLE quantities (obs 8 + arm joints) per transition."""
    E = 32
    ora = O.OracleReach(E, seed=3)
    init = ora.get_state()
    obs0 = ora.reset()[0]
    rng = np.random.default_rng(5)
    acts = rng.uniform(-1, 1, size=(26, E, 4))
    acts[2] *= 2.5
    states, obs_l, rew_l, done_l, succ_l, fut_l, sens_l = [ora.get_state()], [], [], [], [], [], []
    for t in range(acts.shape[0]):
        s0 = states[-1]
        ora.set_state(s0)
        o = ora.step(acts[t])
        nxt = ora.get_state()
        sens = np.zeros(E)
        for k in range(2):
            sp = s0.copy()
            sp[:, :26] += np.random.default_rng(100 * t + k).uniform(-1e-6, 1e-6, size=(E, 26))
            ora.set_state(sp)
            op = ora.step(acts[t])
            sens = np.maximum(sens, np.maximum(np.abs(op[0] - o[0]).max(1), np.abs(ora.get_state()[:, :7] - nxt[:, :7]).max(1)))
        ora.set_state(nxt)
        states.append(nxt); obs_l.append(o[0]); rew_l.append(o[3]); done_l.append(o[4]); succ_l.append(o[5]); fut_l.append(o[6]); sens_l.append(sens)
    path = os.path.join(ROOT, "tests", "golden", "reach_oracle_rollout.npz")
    np.savez_compressed(path, init_state=init, reset_obs=obs0, actions=acts, states=np.stack(states), obs=np.stack(obs_l),
                        rew=np.stack(rew_l), done=np.stack(done_l), succ=np.stack(succ_l), fut=np.stack(fut_l), sens=np.stack(sens_l))
    print("wrote", path, "max sens %.2e" % np.max(sens_l))


def _ezpolicy(o):
    """the control law of the reference's scripted policy (xarm_handover.py:404-446), restated for fixtures"""
    obj, g1, q1, g2, q2 = o[0:3], o[13:16], o[19], o[21:24], o[27]
    ig1 = q1 < 0.25 and np.linalg.norm(obj - g1) < 0.05
    ig2 = q2 < 0.25 and np.linalg.norm(obj - g2) < 0.05
    d1 = obj - g1 + [-0.07, 0, 0]
    d2 = obj - g2 + [0.07, 0, 0]
    a = [0.0] * 8
    a[3] = -0.5 if np.linalg.norm(obj - g1) < 0.1 else 0.5
    a[7] = -0.5 if np.linalg.norm(obj - g2) < 0.1 else 0.5
    if not ig1:
        a[0:3] = list(d1 / np.linalg.norm(d1))
    elif not ig2:
        a[0:3] = [0.5, 0, 0.5]
        a[4:7] = list(d2 / np.linalg.norm(d2))
    else:
        a[4] = -0.5
    return a


class JitteredHandover:
    """The reference's ezpolicy (xarm_handover.py:404-446, restated above) as a closed-loop policy on OBSERVATIONS with
    per-env jitter: a start delay, a constant steering bias and per-step action noise, so that a batch covers the
    reach, the first grasp, the lift, the second arm's approach and the two-arm phase at many different phases at once
    (tests/test_handover_coop.py: live-oracle step parity of both Handover kernel families)."""

    def __init__(self, E, seed=0, horizon=36):
        rng = np.random.default_rng(seed)
        self.delay = rng.integers(0, 6, E)
        self.bias = rng.uniform(-0.15, 0.15, (E, 8))
        self.bias[:, [3, 7]] = 0.0
        self.rng = rng
        self.horizon = horizon

    def __call__(self, obs, t):
        E = obs.shape[0]
        a = np.array([_ezpolicy(obs[e]) for e in range(E)], dtype=np.float64)
        a += self.bias + self.rng.uniform(-0.1, 0.1, (E, 8)) * np.array([1, 1, 1, 0, 1, 1, 1, 0])
        a[t < self.delay] = 0.0
        return np.clip(a, -1, 1)


def handover():
    """XarmHandover-v0: 24 envs under the reference's scripted handover policy for 30 steps + 8 random steps, with
    sensitivities of the continuous state (q, qd of both arms, object pose / velocity)."""
    E = 24
    ora = O.OracleHandover(E, seed=2)
    init = ora.get_state()
    obs = ora.reset()[0]
    reset_obs = obs.copy()
    rng = np.random.default_rng(8)
    states, acts, obs_l, rew_l, done_l, succ_l, sens_l = [ora.get_state()], [], [], [], [], [], []
    cont = np.r_[0:36, 38:51]
    for t in range(38):
        a = np.array([_ezpolicy(obs[e]) for e in range(E)]) if t < 30 else rng.uniform(-1, 1, (E, 8))
        s0 = states[-1]
        ora.set_state(s0)
        o = ora.step(a)
        nxt = ora.get_state()
        sens = np.zeros(E)
        for k in range(2):
            sp = s0.copy()
            sp[:, cont] += np.random.default_rng(1000 * t + k).uniform(-1e-6, 1e-6, size=(E, cont.size))
            qn = sp[:, 41:45]
            sp[:, 41:45] = qn / np.linalg.norm(qn, axis=1, keepdims=True)
            ora.set_state(sp)
            ora.step(a)
            sens = np.maximum(sens, np.abs(ora.get_state()[:, cont] - nxt[:, cont]).max(1))
        ora.set_state(nxt)
        obs = o[0]
        states.append(nxt); acts.append(a); obs_l.append(o[0]); rew_l.append(o[3]); done_l.append(o[4]); succ_l.append(o[5]); sens_l.append(sens)
    path = os.path.join(ROOT, "tests", "golden", "handover_oracle_rollout.npz")
    np.savez_compressed(path, init_state=init, reset_obs=reset_obs, actions=np.stack(acts), states=np.stack(states), obs=np.stack(obs_l),
                        rew=np.stack(rew_l), done=np.stack(done_l), succ=np.stack(succ_l), sens=np.stack(sens_l))
    st = states[30]
    print("wrote", path, "both arms touching at some step:", int((np.stack(states)[:, :, 70:72].sum(2) == 2).any(0).sum()),
          "handed over:", int(((st[:, 38] * states[0][:, 38] < 0) | (st[:, 40] > 0.08)).sum()), "of", E, "max sens %.2e" % np.max(sens_l))


def _ez_on(o, k):
    """_ezpolicy steering at stick k of a num_obj = 2 observation (42 wide: pos 6, quat 8, v 6, w 6, arms 16)"""
    o29 = np.concatenate([o[3 * k:3 * k + 3], o[6 + 4 * k:10 + 4 * k], o[14 + 3 * k:17 + 3 * k], o[20 + 3 * k:23 + 3 * k], o[26:42]])
    return _ezpolicy(o29)


HO2_CONT = np.r_[0:36, 38:64]   # q, qd of both arms, pose and velocity of both sticks


def ho2_sens(ora, s0, a, nxt, seed):
    E = s0.shape[0]
    sens = np.zeros(E)
    for k in range(2):
        sp = s0.copy()
        sp[:, HO2_CONT] += np.random.default_rng(1000 * seed + k).uniform(-1e-6, 1e-6, size=(E, HO2_CONT.size))
        for o in range(2):
            qn = sp[:, 44 + 4 * o:48 + 4 * o]
            sp[:, 44 + 4 * o:48 + 4 * o] = qn / np.linalg.norm(qn, axis=1, keepdims=True)
        ora.set_state(sp)
        ora.step(a)
        sens = np.maximum(sens, np.abs(ora.get_state()[:, HO2_CONT] - nxt[:, HO2_CONT]).max(1))
    return sens


def ho2_scene(st, rng):
    """the three scripted two-stick scenes of the rollout fixture, env e in group (e % 24) // 8, with per-env jitter: stick 1
    laid across stick 0; the sticks side by side and touching on arm 1's side; the sticks far apart (stick 1 on arm 1's side)"""
    for e in range(st.shape[0]):
        j = rng.uniform(-1, 1, 4)
        grp = (e % 24) // 8
        if grp == 0:
            st[e, 38:41] = [-0.19 + 0.03 * j[0], 0.05 * j[1], 0.025]
            st[e, 41:44] = [st[e, 38] + 0.03 * j[2], st[e, 39] + 0.012 * j[3], 0.0752]
        elif grp == 1:
            st[e, 38:41] = [-0.2 + 0.03 * j[0], 0.04 * j[1], 0.025]
            st[e, 41:44] = [st[e, 38] + 0.02 * j[2], st[e, 39] + (0.05 + 0.001 * j[3]) * (1 if e % 2 else -1), 0.025]
        else:
            st[e, 38:41] = [0.2 + 0.03 * j[0], 0.1 * j[1], 0.025]
            st[e, 41:44] = [-0.2 + 0.03 * j[2], 0.05 * j[3], 0.025]
        st[e, 44:52] = [0, 0, 0, 1, 0, 0, 0, 1]
        st[e, 52:64] = 0
        st[e, 70:94] = 0
    return st


class JitteredHandover2:
    """closed-loop policy of the two-stick live fixture (tests/test_handover2.py): group 0 small random arm motion over the
    crossed sticks, group 1 the reference's ezpolicy steering at stick 0 (dragging it against stick 1), group 2 ezpolicy
    steering at stick 1; per-env start delay, steering bias and action noise"""

    def __init__(self, E, seed=0, horizon=34):
        rng = np.random.default_rng(seed)
        self.delay = rng.integers(0, 5, E)
        self.bias = rng.uniform(-0.12, 0.12, (E, 8))
        self.bias[:, [3, 7]] = 0.0
        self.rng, self.horizon = rng, horizon

    def __call__(self, obs, t):
        E = obs.shape[0]
        a = np.zeros((E, 8))
        for e in range(E):
            grp = (e % 24) // 8
            a[e] = self.rng.uniform(-0.3, 0.3, 8) if grp == 0 else _ez_on(obs[e], 0 if grp == 1 else 1)
        a += self.bias + self.rng.uniform(-0.1, 0.1, (E, 8)) * np.array([1, 1, 1, 0, 1, 1, 1, 0])
        a[t < self.delay] = 0.0
        return np.clip(a, -1, 1)


def handover2():
    """XarmHandover-v0 with num_obj = 2 (the reference's test.py configuration): 24 envs, 30 scripted + 6 random steps.
    Envs 0-7: stick 1 laid across the top of stick 0 (stick/stick manifold under load) with small random arm motion;
    8-15: the two sticks side by side and touching on arm 1's side, the reference's ezpolicy steering at stick 0 (grasp
    flags set, stick 0 dragged against stick 1); 16-23: ezpolicy steering at stick 1 with stick 0 out of the way (pads hold
    stick 1: the grasp flags, which read contacts with stick 0 only, must stay clear)."""
    E = 24
    ora = O.OracleHandover(E, seed=2, num_obj=2, goal_shape="any")
    init = ora.get_state()
    obs = ora.reset()[0]
    reset_state, reset_obs = ora.get_state(), obs.copy()
    rng = np.random.default_rng(9)
    st = ora.get_state()
    for e in range(E):
        j = rng.uniform(-1, 1, 4)
        if e < 8:
            st[e, 38:41] = [-0.19 + 0.03 * j[0], 0.05 * j[1], 0.025]
            st[e, 41:44] = [st[e, 38] + 0.03 * j[2], st[e, 39] + 0.012 * j[3], 0.0752]
        elif e < 16:
            st[e, 38:41] = [-0.2 + 0.03 * j[0], 0.04 * j[1], 0.025]
            st[e, 41:44] = [st[e, 38] + 0.02 * j[2], st[e, 39] + (0.05 + 0.001 * j[3]) * (1 if e % 2 else -1), 0.025]
        else:
            st[e, 38:41] = [0.2 + 0.03 * j[0], 0.1 * j[1], 0.025]
            st[e, 41:44] = [-0.2 + 0.03 * j[2], 0.05 * j[3], 0.025]
        st[e, 44:52] = [0, 0, 0, 1, 0, 0, 0, 1]
        st[e, 52:64] = 0
        st[e, 70:94] = 0
    ora.set_state(st)
    obs = np.zeros((E, 42))
    z8 = np.zeros((E, 8))
    obs = ora.step(z8)[0]          # one quiet step so that the scripted scene has settled contact impulses
    states, acts, obs_l, rew_l, done_l, succ_l, sens_l = [ora.get_state()], [], [], [], [], [], []
    for t in range(36):
        if t < 30:
            a = np.zeros((E, 8))
            a[:8] = rng.uniform(-0.3, 0.3, (8, 8))
            a[8:16] = [_ez_on(obs[e], 0) for e in range(8, 16)]
            a[16:] = [_ez_on(obs[e], 1) for e in range(16, E)]
        else:
            a = rng.uniform(-1, 1, (E, 8))
        s0 = states[-1]
        ora.set_state(s0)
        o = ora.step(a)
        nxt = ora.get_state()
        sens = ho2_sens(ora, s0, a, nxt, t)
        ora.set_state(nxt)
        obs = o[0]
        states.append(nxt); acts.append(a); obs_l.append(o[0]); rew_l.append(o[3]); done_l.append(o[4]); succ_l.append(o[5]); sens_l.append(sens)
    path = os.path.join(ROOT, "tests", "golden", "handover2_oracle_rollout.npz")
    S = np.stack(states)
    np.savez_compressed(path, init_state=init, reset_state=reset_state, reset_obs=reset_obs, actions=np.stack(acts), states=S,
                        obs=np.stack(obs_l), rew=np.stack(rew_l), done=np.stack(done_l), succ=np.stack(succ_l), sens=np.stack(sens_l))
    print("wrote", path, "| stick 1 still on stick 0 at the end:", int((S[-7, :8, 43] > 0.06).sum()), "of 8",
          "| grasp flag arm 1 ever set (8-15):", int((S[:, 8:16, 94] > 0).any(0).sum()), "of 8",
          "| pad impulses on stick 1 (16-23):", int((S[:, 16:, 86:90] > 0).any(2).any(0).sum()), "of 8, grasp flags there:", int(S[:, 16:, 94:96].sum()),
          "| max sens %.2e, share < 1e-3: %.2f" % (np.max(sens_l), np.mean(np.stack(sens_l) < 1e-3)))


if __name__ == "__main__":
    if "handover2" in sys.argv:
        handover2()
        sys.exit(0)
    main()
    reach()
    handover()
    handover2()
