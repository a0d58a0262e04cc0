"""hand-off counts and ms per step of the staged Handover step (one MI355X): python tools/probes/ho_stage_counts.py [envs]
(XARM_HO_STAGES / XARM_HO_STAGE_TICKS / XARM_HO_SIDE_PRIO in the environment select the variant)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, gym_xarm_amd as gx
E = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
env = gx.make("XarmPDHandover-v0", num_envs=E, seed=0)
env.reset()
a = [torch.rand(E, 8, device="cuda") * 2 - 1 for _ in range(8)]
tot = []
for t in range(150):
    env.step(a[t % 8])
    if t >= 100:
        tot.append(env.debug_counts())
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(200): env.step(a[k % 8])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
print("stages %s ticks %s prio %s: finished %.1f handed off %.1f per call, %.3f ms/step %.3g env steps/s" % (
    os.environ.get("XARM_HO_STAGES", "default"), os.environ.get("XARM_HO_STAGE_TICKS", "-"), os.environ.get("XARM_HO_SIDE_PRIO", "0"),
    sum(x[0] for x in tot) / len(tot), sum(x[1] for x in tot) / len(tot), dt * 1e3, E / dt), flush=True)
