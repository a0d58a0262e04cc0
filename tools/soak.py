"""Soak: 1500 steps of the headline configuration (65 536 envs, auto-reset; `stack`: 3000 steps of 8 192 StackTower envs on
the class-ordered step kernel), finiteness and throughput drift.  usage: soak.py [pnp|lazy|stack] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
STACK = len(sys.argv) > 1 and sys.argv[1] == "stack"
E = 8192 if STACK else 65536
MODE = "lazy" if len(sys.argv) > 1 and sys.argv[1] == "lazy" else True
env = gym_xarm_amd.make("XarmPDStackTower-v0" if STACK else "XarmPDPickAndPlace-v0", num_envs=E, seed=11, auto_reset=MODE)
env.reset()
g = torch.Generator(device=env.device); g.manual_seed(5)
ring = [torch.rand(E, env.act_dim, device=env.device, generator=g) * 2 - 1 for _ in range(32)]
bad = 0
succ = 0
t0 = time.perf_counter()
N = int(sys.argv[2]) if len(sys.argv) > 2 else (3000 if STACK else 1500)
EVERY = max(100, N // 15)
for k in range(N):
    obs, rew, done, info = env.step(ring[k % 32])
    if k % EVERY == EVERY - 1:
        torch.cuda.synchronize()
        fin = bool(torch.isfinite(obs["observation"]).all())
        bad += 0 if fin else 1
        succ += int(info["is_success"].sum())
        dt = time.perf_counter() - t0
        print("steps %4d: %.3e env steps/s, finite %s, max|obs| %.1f, successes in this step %d" % (
            k + 1, E * EVERY / dt, fin, float(obs["observation"].abs().max()), int(info["is_success"].sum())), flush=True)
        t0 = time.perf_counter()
st = env.get_state()
print("final state finite:", bool(torch.isfinite(st).all()), " episodes per env: %.1f" % float(st[:, 135 if STACK else 53].mean()))
sys.exit(1 if bad or not bool(torch.isfinite(st).all()) else 0)
