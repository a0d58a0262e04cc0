#!/usr/bin/env python3
"""Generate tests/golden/pnp_oracle_rollout.npz with the CPU oracle (float64): a short random
rollout of 32 envs plus a scripted reach-grasp-lift of 4 envs, with per-step sensitivity estimates
(oracle/parity.py).  GPU parity tests replay every recorded transition from its recorded state."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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


if __name__ == "__main__":
    main()
