"""Quick on-GPU sanity + timing (development aid; bench.py is the contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gym_xarm_amd
import __graft_entry__ as ge

E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if os.environ.get("SMOKE"): ge.smoke()
env = gym_xarm_amd.make("XarmPDPickAndPlace-v0", num_envs=E, seed=0)
t0 = time.time(); env.reset(); torch.cuda.synchronize(); print("reset all %.3fs" % (time.time() - t0), flush=True)
acts = [torch.rand(E, 4, device="cuda") * 2 - 1 for _ in range(8)]
for i in range(3):
    env.step(acts[i % 8])
torch.cuda.synchronize()
env.timing_enable(True)
t0 = time.time()
for i in range(steps):
    obs, rew, done, info = env.step(acts[i % 8])
torch.cuda.synchronize()
dt = time.time() - t0
ms, n = env.timing_read()
print("E=%d steps=%d wall %.3fs -> %.3e env steps/s ; k_step avg %.3f ms (%d launches)" % (E, steps, dt, E * steps / dt, ms / max(n, 1), n), flush=True)
print("done frac", done.float().mean().item(), "succ", info["is_success"].float().mean().item(), "box z mean", obs["achieved_goal"][:, 2].mean().item())
