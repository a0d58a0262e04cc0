"""k_step_coop / k_reset_coop time at a small batch (development aid for tools/coop_split.sh and the setup probes):
random actions (some wavefront always carries finger-pad rows) and objects parked away from the gripper (none does)."""
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
env.timing_enable(False)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
for i in range(3):
    env.reset()
ev1.record()
torch.cuda.synchronize()
print("k_step_coop %.3f ms   full reset (6 ticks, coop) %.3f ms" % (ms / n, ev0.elapsed_time(ev1) / 3), flush=True)
far = env.get_state().clone()
far[:, 18] = 0.45; far[:, 19] = 0.28; far[:, 20] = 0.026; far[:, 21:24] = 0; far[:, 24] = 1; far[:, 25:31] = 0
zero = torch.zeros(E, 4, device="cuda")
env.set_state(far); env.step(zero); env.set_state(far)
env.timing_enable(True)
for i in range(10):
    env.step(zero)
torch.cuda.synchronize()
ms, n = env.timing_read()
print("k_step_coop, objects parked away from the gripper (no pad rows in any wavefront) %.3f ms" % (ms / n), flush=True)
