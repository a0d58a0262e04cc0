"""Scripted pick with the CPU oracle (sanity of the physics spec; cf. the reference's
open-loop _run_demo, xarm_pick_and_place.py:310-349)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import oracle as O

def run(verbose=True):
    env = O.OraclePnP(1, seed=1)
    t0 = time.time()
    obs, ag, dg = env.reset()
    if verbose: print('reset %.2fs' % (time.time() - t0), 'hand', obs[0, :3], 'finger', obs[0, 6], 'box', ag[0], 'goal', dg[0])
    def eef():
        pos, _ = O.fk(env.state[0, :9]); return pos[7]
    def go(target, grip, n):
        for i in range(n):
            e = eef()
            a = np.zeros(4)
            a[:3] = np.clip((np.asarray(target) - e) / (0.25 * 0.25), -1, 1)
            a[3] = grip
            o, ag, dg, r, d, s = env.step(a[None])
            if verbose: print('eef', np.round(eef(), 3), 'fing', np.round(env.state[0, 7:9], 4), 'box', np.round(ag[0], 4), 'touch', env.state[0, 50], 'r', r[0], 'd', d[0])
        return ag
    # let the box settle
    go(eef(), 1.0, 25)
    box = env.state[0, 18:21].copy()
    go([box[0], box[1], 0.25], 1.0, 6)
    go([box[0], box[1], 0.15], 1.0, 8)
    go([box[0], box[1], 0.15], -1.0, 6)
    ag = go([box[0], box[1], 0.35], -1.0, 10)
    return ag[0, 2]

if __name__ == '__main__':
    z = run()
    print('final box z', z, 'LIFTED' if z > 0.1 else 'not lifted')
