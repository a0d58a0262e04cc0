import sys, os
sys.path.insert(0, '/root/repo')
import torch, gym_xarm_amd as gx
E = 65536
a = [torch.rand(E, 4, device="cuda", generator=torch.Generator(device="cuda").manual_seed(k)) * 2 - 1 for k in range(3)]
def run(n, off, steps=3):
    env = gx.make("XarmPDPickAndPlace-v0", num_envs=n, seed=17, env_id_offset=off)
    env.reset()
    outs = []
    for k in range(steps):
        obs, rew, done, info = env.step(a[k][off:off + n])
        outs.append(env.get_state().clone())
    env.close()
    return outs
full, half = run(E, 0), run(E // 2, E // 2)
for k in range(3):
    d = (full[k][E // 2:] - half[k]).abs()
    bad = (d.max(dim=1).values > 0).nonzero()[:, 0]
    print("step", k, "envs differing", bad.numel(), "max diff", d.max().item())
    if bad.numel():
        i = bad[0].item()
        cols = (d[i] > 0).nonzero()[:, 0].tolist()
        print("  env", i + E // 2, "cols", cols[:20], "vals", full[k][E // 2 + i][cols[:6]].tolist(), half[k][i][cols[:6]].tolist())
        print("  touch/mug/steps/episode", full[k][E//2+i][50:54].tolist(), half[k][i][50:54].tolist(), "lam_p", full[k][E//2+i][42:46].tolist())
