"""Throughput of the other BASELINE configs on one GPU (development aid; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
for env_id, E, A, steps in (("XarmReach-v0", 4096, 4, 100), ("XarmReach-v0", 65536, 4, 100), ("XarmPDHandover-v0", 16384, 8, 60), ("XarmPDHandover-v0", 32768, 8, 60),
                           ("XarmPDStackTower-v0", 8192, 8, 60), ("XarmPDStackTower-v0", 16384, 8, 60)):
    env = gym_xarm_amd.make(env_id, num_envs=E, seed=0)
    env.reset()
    acts = [torch.rand(E, A, device="cuda") * 2 - 1 for _ in range(8)]
    for i in range(5):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    env.timing_enable(True)
    t0 = time.perf_counter()
    for i in range(steps):
        env.step(acts[i % 8])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, n = env.timing_read()
    print("%-20s E=%6d  %.3e env steps/s  (%.2f ms/step, step kernel %.3f ms)" % (env_id, E, E * steps / dt, dt / steps * 1e3, ms / n), flush=True)
    env.close()
