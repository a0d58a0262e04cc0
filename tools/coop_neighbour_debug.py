"""development aid: does an env's cooperative reset depend on its wavefront neighbours? prints the differing fields"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, gym_xarm_amd
E = 64
env = gym_xarm_amd.make("XarmPDPickAndPlace-v0", num_envs=E, seed=9, auto_reset=False)
env.reset()
gen = torch.Generator().manual_seed(1)
for t in range(10):
    env.step(torch.rand(E, 4, generator=gen) * 2 - 1)
st0 = env.get_state().clone()
env.reset()
full = env.get_state().clone()
env.set_state(st0); env.reset(); again = env.get_state()
print("full vs full again equal:", torch.equal(full, again))
for lo, n in ((0, 5), (3, 5), (17, 5), (0, 4), (4, 4), (0, 1), (20, 1), (0, 64)):
    env.set_state(st0)
    m = torch.zeros(E, dtype=torch.uint8); m[lo:lo + n] = 1
    env.reset(mask=m)
    part = env.get_state()
    d = (part[lo:lo + n] - full[lo:lo + n]).abs()
    bad = (d.max(dim=1).values > 0).nonzero().flatten().tolist()
    print("lo", lo, "n", n, "max diff", float(d.max()), "envs differing", [lo + b for b in bad][:10],
          "fields", sorted(set((d > 0).nonzero()[:, 1].tolist()))[:20])
