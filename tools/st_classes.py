"""StackTower: population and cost of the row-set classes the step kernel groups the envs by (development aid).
For every class present after 40 random steps: how many envs it holds, and the step-kernel time of a batch in which EVERY
env is a copy of one env of that class (zero actions, auto-reset off) - the time a wavefront of that class takes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
E = 8192
env = gym_xarm_amd.make("XarmPDStackTower-v0", num_envs=E, seed=0)
env.reset()
g = torch.Generator(device=env.device); g.manual_seed(0)
for i in range(40):
    a = torch.rand(E, 8, device=env.device, generator=g) * 2 - 1
    env.step(a)
keys = env.class_keys().cpu()
state = env.get_state().clone()
env.close()
probe = gym_xarm_amd.make("XarmPDStackTower-v0", num_envs=E, seed=0, auto_reset=False)
probe.reset()
def name(k):
    pairs = [p for b, p in ((1, "01"), (2, "02"), (4, "12")) if k & b]
    return "pairs[%s] pads[%s%s]" % (",".join(pairs), "a" if k & 8 else "", "b" if k & 16 else "")
rows = []
for k in sorted(set(keys.tolist())):
    idx = (keys == k).nonzero()[:, 0]
    s = state[idx[0].item()].unsqueeze(0).repeat(E, 1)
    s[:, 134] = 0                                              # step counter: no time limit inside the probe
    probe.set_state(s)
    z = a[idx[0].item()].unsqueeze(0).repeat(E, 1)            # the action that env just took
    probe.step(z); probe.set_state(s)
    probe.timing_enable(True)
    for _ in range(3):
        probe.step(z); probe.set_state(s)
    torch.cuda.synchronize()
    ms, n = probe.timing_read()
    probe.timing_enable(False)
    rows.append((k, idx.numel(), ms / n))
    print("class %3d  %-40s envs %5d   homogeneous batch %.2f ms" % (k, name(k), idx.numel(), ms / n), flush=True)
