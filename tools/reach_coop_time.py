"""k_reach_step_coop time at 4096 envs (development aid, used with tools/coop_split.sh-style library variants)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
E = 4096
env = gym_xarm_amd.make("XarmReach-v0", num_envs=E, seed=0, auto_reset=False)
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
print("k_reach_step_coop %.3f ms" % (ms / n), flush=True)
