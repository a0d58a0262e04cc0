"""XarmReach-v0: one-env-per-lane kernels vs the cooperative (16 lanes per env) kernels over batch sizes, to place the
default cross-over (XARM_STEP_COOP_LIMIT_DEFAULT / XARM_RESET_COOP_LIMIT_DEFAULT).  Development aid; bench.py is the
contract.  Lockstep episodes (the reset launch falls on every 25th step), 100 steps per size."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
FAM = {"lane": dict(reset_coop_limit=-1, step_coop_limit=-1), "coop": dict(reset_coop_limit=1 << 30, step_coop_limit=1 << 30)}
for E in (256, 1024, 4096, 16384, 32768, 65536, 131072, 262144):
    row = []
    for fam in ("lane", "coop"):
        env = gym_xarm_amd.make("XarmReach-v0", num_envs=E, seed=0, **FAM[fam])
        env.reset()
        acts = [torch.rand(E, 4, device="cuda") * 2 - 1 for _ in range(8)]
        for i in range(5):
            env.step(acts[i % 8])
        torch.cuda.synchronize()
        env.timing_enable(True)
        t0 = time.perf_counter()
        for i in range(100):
            env.step(acts[i % 8])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ms, n = env.timing_read()
        row.append("%s %.3e steps/s (step kernel %.3f ms)" % (fam, E * 100 / dt, ms / n))
        env.close()
    print("E=%7d  %s" % (E, "   ".join(row)), flush=True)
