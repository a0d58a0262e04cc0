"""StackTower: step time with and without the class-homogeneous visiting order (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
E = 8192
for flag in ("0", "1"):
    os.environ["XARM_ST_CLASS_ORDER"] = flag
    env = gym_xarm_amd.make("XarmPDStackTower-v0", num_envs=E, seed=0)
    env.reset()
    g = torch.Generator(device=env.device); g.manual_seed(0)
    acts = [torch.rand(E, 8, device=env.device, generator=g) * 2 - 1 for _ in range(8)]
    for i in range(10): env.step(acts[i % 8])
    env.timing_enable(True)
    for i in range(100): env.step(acts[i % 8])
    torch.cuda.synchronize()
    ms, k = env.timing_read()
    print("XARM_ST_CLASS_ORDER=%s: step kernels %.3f ms per call (100 calls, random actions, auto-reset on)" % (flag, ms / k), flush=True)
    env.close()
