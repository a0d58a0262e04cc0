"""Development aid: XarmPDHandover-v0 ms per step at small batch sizes - default (cooperative rows up to 2 048 envs), the fast pipeline pinned (step_coop_limit = 1), the lane-pair kernels pinned - DESIGN.md 10b."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, gym_xarm_amd as gx
for E in (64, 512, 2048, 4096):
    for tag, kw in (("default", {}), ("pipeline", dict(step_coop_limit=1)), ("lane", dict(step_coop_limit=-1, reset_coop_limit=-1))):
        env = gx.make("XarmPDHandover-v0", num_envs=E, seed=0, **kw)
        env.reset()
        a = [torch.rand(E, 8, device="cuda") * 2 - 1 for _ in range(8)]
        for k in range(30): env.step(a[k % 8])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(100): env.step(a[k % 8])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        print("E %5d %-9s %.3f ms/step  %.3g env steps/s  %s" % (E, tag, dt * 1e3, E / dt, env.pipeline_info()["fast_pipeline"]), flush=True)
        env.close()
