"""StackTower at its BASELINE per-GPU size (8192 envs), a few steps - target for rocprofv3 runs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
E = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
env = gym_xarm_amd.make("XarmPDStackTower-v0", num_envs=E, seed=0)
env.reset()
g = torch.Generator(device=env.device); g.manual_seed(0)
acts = [torch.rand(E, 8, device=env.device, generator=g) * 2 - 1 for _ in range(8)]
for i in range(3):
    env.step(acts[i])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    env.step(acts[i % 8])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("XarmPDStackTower-v0 E=%d: %.3e env steps/s (%.2f ms/step)" % (E, E * steps / dt, dt / steps * 1e3))
