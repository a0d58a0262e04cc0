"""k_step_coop / k_reset_coop time at a small batch (development aid for tools/coop_split.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
E = 4096
env = gym_xarm_amd.make("XarmPDPickAndPlace-v0", num_envs=E, seed=0, auto_reset=False)
env.reset()
acts = [torch.rand(E, 4, device="cuda") * 2 - 1 for _ in range(8)]
for i in range(5):
    env.step(acts[i % 8])
torch.cuda.synchronize()
env.timing_enable(True)
for i in range(20):
    env.step(acts[i % 8])
torch.cuda.synchronize()
ms, n = env.timing_read()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for i in range(3):
    env.reset()
ev1.record()
torch.cuda.synchronize()
print("k_step_coop %.3f ms   full reset (6 ticks, coop) %.3f ms" % (ms / n, ev0.elapsed_time(ev1) / 3), flush=True)
