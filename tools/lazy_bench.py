"""Throughput of the opt-in lazy auto-reset mode at the headline size (useful = non-reset env steps)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
E = 65536
for mode in (True, "lazy"):
    env = gym_xarm_amd.make("XarmPDPickAndPlace-v0", num_envs=E, seed=0, auto_reset=mode)
    env.reset()
    g = torch.Generator(device=env.device); g.manual_seed(1234)
    ring = [torch.rand(E, 4, device=env.device, generator=g) * 2 - 1 for _ in range(64)]
    for i in range(10): env.step(ring[i])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    useful = torch.zeros((), device=env.device)
    for i in range(200):
        obs, rew, done, info = env.step(ring[i % 64])
        useful += E if mode is True else (~info["resetting"]).sum()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("auto_reset=%-5s: %.3e useful env steps/s (%.2f ms per call, %.1f %% of lane-steps useful)" % (
        mode, float(useful) / dt, dt / 200 * 1e3, 100 * float(useful) / (E * 200)), flush=True)
    env.close()
