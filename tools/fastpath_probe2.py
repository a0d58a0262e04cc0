"""step-kernel time of Handover / StackTower batches with every arm driven up and away from the objects (no wavefront holds
a finger contact) against the random-action batch: the wave-level price of the few envs with pad rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
for env_id, E in (("XarmPDHandover-v0", 16384), ("XarmPDStackTower-v0", 8192)):
    for name in ("random", "arms_up"):
        env = gym_xarm_amd.make(env_id, num_envs=E, seed=0, auto_reset=False)
        env.reset()
        g = torch.Generator(device="cuda").manual_seed(1)
        acts = [torch.rand(E, 8, device="cuda", generator=g) * 2 - 1 for _ in range(16)]
        if name == "arms_up":
            for a in acts:
                a[:, 2] = 1.0; a[:, 6] = 1.0; a[:, 3] = 1.0; a[:, 7] = 1.0
        for i in range(30):
            env.step(acts[i % 16])
        torch.cuda.synchronize()
        env.timing_enable(True)
        for i in range(20):
            env.step(acts[i % 16])
        torch.cuda.synchronize()
        ms, n = env.timing_read()
        print(env_id, name, "step kernel %.3f ms" % (ms / n), flush=True)
        env.close()
