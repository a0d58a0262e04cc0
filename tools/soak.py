"""Soak: 1500 steps of the headline configuration (65 536 envs, auto-reset; `stack`: 3000 steps of 8 192 StackTower envs on
the class-ordered step kernel; `handover` / `handover2`: the reference's own test length - test.py:19 runs
_max_episode_steps * 100 = 10 000 steps with a reset every 100 - on 4 096 envs of XarmHandover-v0 with one / two sticks),
finiteness and throughput drift.  usage: soak.py [pnp|lazy|stack|handover|handover2] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
WHAT = sys.argv[1] if len(sys.argv) > 1 else "pnp"
E = {"stack": 8192, "handover": 4096, "handover2": 4096}.get(WHAT, 65536)
MODE = "lazy" if WHAT == "lazy" else True
ENV_ID = {"stack": "XarmPDStackTower-v0", "handover": "XarmHandover-v0", "handover2": "XarmHandover-v0"}.get(WHAT, "XarmPDPickAndPlace-v0")
CFG = {"handover": dict(GUI=False, num_obj=1, same_side_rate=0.5, goal_shape="any", use_stand=False),
       "handover2": dict(GUI=False, num_obj=2, same_side_rate=0.5, goal_shape="any", use_stand=False)}.get(WHAT)   # test.py:9-15
env = gym_xarm_amd.make(ENV_ID, num_envs=E, seed=11, auto_reset=MODE, config=CFG)
env.reset()
g = torch.Generator(device=env.device); g.manual_seed(5)
ring = [torch.rand(E, env.act_dim, device=env.device, generator=g) * 2 - 1 for _ in range(32)]
bad = 0
succ = 0
t0 = time.perf_counter()
N = int(sys.argv[2]) if len(sys.argv) > 2 else {"stack": 3000, "handover": 10000, "handover2": 10000}.get(WHAT, 1500)
EVERY = max(100, N // 15)
EP = {"stack": 135, "handover": 75, "handover2": 99}.get(WHAT, 53)     # episode counter's column in the state row
for k in range(N):
    obs, rew, done, info = env.step(ring[k % 32])
    if k % EVERY == EVERY - 1:
        torch.cuda.synchronize()
        fin = bool(torch.isfinite(obs["observation"]).all())
        bad += 0 if fin else 1
        succ += int(info["is_success"].sum())
        dt = time.perf_counter() - t0
        print("steps %5d: %.3e env steps/s, finite %s, max|obs| %.1f, successes in this step %d" % (
            k + 1, E * EVERY / dt, fin, float(obs["observation"].abs().max()), int(info["is_success"].sum())), flush=True)
        t0 = time.perf_counter()
st = env.get_state()
print("final state finite:", bool(torch.isfinite(st).all()), " episodes per env: %.1f" % float(st[:, EP].mean()))
sys.exit(1 if bad or not bool(torch.isfinite(st).all()) else 0)
