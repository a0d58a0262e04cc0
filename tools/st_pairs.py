import sys, os
sys.path.insert(0, os.getcwd())
import torch, gym_xarm_amd
E = 8192
env = gym_xarm_amd.make("XarmPDStackTower-v0", num_envs=E, seed=0)
env.reset()
g = torch.Generator(device=env.device); g.manual_seed(0)
def report(tag):
    s = env.get_state()
    p = s[:, 54:63].reshape(E, 3, 3)
    d01 = (p[:, 0] - p[:, 1]).norm(dim=1); d02 = (p[:, 0] - p[:, 2]).norm(dim=1); d12 = (p[:, 1] - p[:, 2]).norm(dim=1)
    for thr in (0.060, 0.075, 0.09):
        n = (d01 < thr).int() + (d02 < thr).int() + (d12 < thr).int()
        print(tag, "thr %.3f: envs with 0/1/2/3 near pairs:" % thr, [(n == k).sum().item() for k in range(4)], flush=True)
report("after reset")
for t in range(60):
    env.step(torch.rand(E, 8, device=env.device, generator=g) * 2 - 1)
    if t in (9, 29, 59): report("step %d" % (t + 1))
