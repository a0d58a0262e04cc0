import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd as gx
E = 4096
res = {}
a = torch.rand(E, 4, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)) * 2 - 1
a[:, 2] = 1.0
base = None
for fam, kw in (("plain", dict(step_coop_limit=-1)), ("fast", dict(step_coop_limit=1))):
    env = gx.make("XarmPDPickAndPlace-v0", num_envs=E, seed=9, auto_reset=False, reset_coop_limit=-1, **kw)
    env.reset()
    if base is None:
        for _ in range(8):
            env.step(a)
        base = env.get_state().clone()
    env.set_state(base)
    env.step(a)
    res[fam] = env.get_state().clone()
    env.close()
d = (res["plain"] - res["fast"]).abs()
diff = d.max(dim=1).values > 0
print("contact at start: touch", int((base[:, 50] > 0).sum()), "lam_p", int((base[:, 42:46] != 0).any(dim=1).sum()))
print("different envs", int(diff.sum()), "of", E, "max", d.max().item())
idx = diff.nonzero()[:, 0][:8].tolist()
for i in idx:
    cols = (d[i] > 0).nonzero()[:, 0].tolist()
    print(" env", i, "cols", cols[:14], "maxdiff %.3g" % d[i].max().item(), "q", [round(x, 3) for x in base[i, :9].tolist()], "near limit:",
          [round(min(base[i, k].item() - lo, hi - base[i, k].item()), 3) for k, (lo, hi) in enumerate([(-6.283, 6.283), (-2.059, 2.0944), (-6.283, 6.283), (-0.19198, 3.927), (-6.283, 6.283), (-1.69297, 3.14159), (-6.283, 6.283)])])
