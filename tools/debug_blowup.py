"""Find the first transition where a PickAndPlace env blows up on the GPU and replay it on the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gym_xarm_amd
from oracle import oracle as O
E = 16384
env = gym_xarm_amd.make("XarmPDPickAndPlace-v0", num_envs=E, seed=7)
env.reset()
g = torch.Generator(device=env.device); g.manual_seed(3)
found = 0
for k in range(300):
    a = torch.rand(E, 4, device=env.device, generator=g) * 2.4 - 1.2
    if k % 7 == 0: a = torch.sign(a)
    prev = env.get_state().clone()
    obs, rew, done, info = env.step(a)
    st = env.get_state()
    o = info["terminal_observation"].where((done != 0)[:, None], obs["observation"])
    big = (o.abs().amax(dim=1) > 3000) | (~torch.isfinite(o)).any(dim=1)
    if bool(big.any()):
        idx = big.nonzero().squeeze(1)[:3]
        for e in idx.tolist():
            p = prev[e].cpu().numpy().astype(np.float64)
            print("step", k, "env", e, "max|obs| %.3g" % float(o[e].abs().max()), "prev max|state| %.3g" % np.abs(p[:31]).max())
            np.set_printoptions(precision=4, suppress=True, linewidth=200)
            print(" prev q", p[0:9]); print(" prev qd", p[9:18]); print(" prev box p", p[18:21], "q", p[21:25], "v", p[25:28], "w", p[28:31], "touch/mug", p[50:52], "steps", p[52])
            print(" action", a[e].cpu().numpy())
            print(" gpu obs", o[e].cpu().numpy())
            orc = O.OraclePnP(1, seed=7, env_id_offset=e)
            orc.set_state(p[None])
            oo = orc.step(a[e].cpu().numpy().astype(np.float64)[None])[0]
            print(" oracle obs", oo[0])
        found += 1
        if found >= 3: break
