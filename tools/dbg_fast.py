import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd as gx
E = 8192
a = [torch.rand(E, 4, device="cuda", generator=torch.Generator(device="cuda").manual_seed(70 + j)) * 2 - 1 for j in range(4)]
def run(**kw):
    env = gx.make("XarmPDPickAndPlace-v0", num_envs=E, seed=31, auto_reset=False, reset_coop_limit=-1, **kw)
    o = env.reset()["observation"]
    sts, dist = [], [(o[:, 0:3] - o[:, 8:11]).norm(dim=1).clone()]
    for j in range(4):
        o = env.step(a[j])[0]["observation"]
        sts.append(env.get_state().clone()); dist.append((o[:, 0:3] - o[:, 8:11]).norm(dim=1).clone())
    env.close()
    return sts, dist
(fast, _), (plain, dist) = run(step_coop_limit=1), run(step_coop_limit=-1)
far = torch.ones(E, dtype=torch.bool, device="cuda")
for j in range(4):
    far &= (dist[j] > 0.22) & (dist[j + 1] > 0.22)
    eq = (fast[j] == plain[j]).all(dim=1)
    bad = far & ~eq
    print("step", j, "far", int(far.sum()), "far but different", int(bad.sum()), "all different", int((~eq).sum()))
    if bad.any():
        i = bad.nonzero()[0, 0].item()
        d = (fast[j][i] - plain[j][i]).abs()
        print("  env", i, "max diff", d.max().item(), "cols", (d > 0).nonzero()[:, 0].tolist()[:12], "dist", dist[j][i].item(), dist[j+1][i].item(), "obj v", plain[j][i][25:28].tolist())
