import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd as gx
E = 8192
g = torch.Generator(device="cuda").manual_seed(3)
a = [torch.rand(E, 4, device="cuda", generator=g) * 2 - 1 for j in range(12)]
for x in a:
    x[:, 2] = 1.0
def run(**kw):
    env = gx.make("XarmPDPickAndPlace-v0", num_envs=E, seed=31, auto_reset=False, reset_coop_limit=-1, **kw)
    env.reset()
    sts = []
    for j in range(12):
        env.step(a[j]); sts.append(env.get_state().clone())
    env.close()
    return sts
fast, plain = run(step_coop_limit=1), run(step_coop_limit=-1)
for j in range(12):
    eq = (fast[j] == plain[j]).all(dim=1)
    d = (fast[j] - plain[j]).abs()
    cols = (d > 0).any(dim=0).nonzero()[:, 0].tolist()
    print("step", j, "different", int((~eq).sum()), "max", d.max().item(), "cols", cols[:24], "touch", int((plain[j][:, 50] > 0).sum()), "lam_p", int((plain[j][:, 42:46] != 0).any(dim=1).sum()))
