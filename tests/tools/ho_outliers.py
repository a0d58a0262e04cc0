"""Development aid: two transitions of the live Handover fixture (tests/test_handover_coop.py) on which a float32 kernel differs from the
oracle by O(0.1) although two 1e-6 probes of the oracle saw a response of 3e-3: float64 cores agree to 1e-8, the oracle itself answers a 1e-7
perturbation with 0.2 - a stick held by sliding pads.  Why the fixture probes six draws at two amplitudes."""
import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, "."); sys.path.insert(0, "tests/tools")
from conftest import HostCore
from oracle import oracle as O
from gen_oracle_fixtures import JitteredHandover
hc = HostCore()
E = 256
ora = O.OracleHandover(E, seed=31)
obs = ora.reset()[0]
pol = JitteredHandover(E, seed=4)
CONT = np.r_[0:36, 38:51]
for t in range(20):
    st0 = ora.get_state()
    a = pol(obs, t)
    o = ora.step(a)
    nxt = ora.get_state()
    obs = o[0]
    if t in (13, 19):
        e = 160 if t == 13 else 178
        sl = slice(e, e + 1)
        for f32 in (0, 1):
            hs, *_ = hc.ho_step(st0[sl], a[sl], f32=f32, seed=31, off=e)
            hcs, *_ = hc.hoc_step(st0[sl], a[sl], f32=f32, seed=31, off=e)
            print(t, e, "f32" if f32 else "f64", "lane err", np.abs(hs - nxt[sl])[:, CONT].max(), "coop err", np.abs(hcs - nxt[sl])[:, CONT].max())
        # sensitivity to larger perturbations
        for eps in (1e-7, 1e-6, 1e-5, 1e-4):
            worst = 0
            for j in range(8):
                sp = st0[sl].copy()
                sp[:, CONT] += np.random.default_rng(j).uniform(-eps, eps, size=(1, CONT.size))
                sp[:, 41:45] /= np.linalg.norm(sp[:, 41:45], axis=1, keepdims=True)
                o2 = O.OracleHandover(1, seed=31, env_id_offset=e)
                o2.set_state(sp); o2.step(a[sl])
                worst = max(worst, np.abs(o2.get_state()[:, CONT] - nxt[sl][:, CONT]).max())
            print("   eps", eps, "response", worst)
        print("   state: obj", st0[e, 38:45], "touch", st0[e, 70:72], "->", nxt[e, 70:72], "lam_p", nxt[e, 62:70])
