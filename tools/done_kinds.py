"""PickAndPlace at the headline size: per step, how many episodes end by success (not predictable before the step) and how
many of those end with the hand at the object (= envs the fast step hands off; their reset is the call's critical path).
Development aid behind DESIGN.md 4b."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
E = 65536
env = gym_xarm_amd.make("XarmPDPickAndPlace-v0", num_envs=E, seed=0)
env.reset()
env.set_episode_steps(torch.arange(E, device=env.device) % 50)
g = torch.Generator(device=env.device); g.manual_seed(0)
ring = [torch.rand(E, 4, device=env.device, generator=g) * 2 - 1 for _ in range(16)]
hist = {}
near_hist = {}
late_hist = {}
prev_d = None
for t in range(400):
    o0 = env.get_state()
    prev_d = (o0[:, 18:21] - o0[:, 31:34]).norm(dim=1)      # object to goal before the step (state layout: S_BP = 18, S_GOAL = 31)
    obs, rew, done, info = env.step(ring[t % 16])
    if t < 100: continue
    succ = done.bool() & (info["is_success"] > 0.5) & ~info["TimeLimit.truncated"].bool()
    term = info["terminal_observation"]
    d = (term[:, 0:3] - term[:, 8:11]).norm(dim=1)
    n, m = int(succ.sum()), int((succ & (d < 0.07)).sum())
    for thr in (0.06, 0.08, 0.10):
        k = int((succ & (prev_d > thr)).sum())
        late_hist.setdefault(thr, {})
        late_hist[thr][k] = late_hist[thr].get(k, 0) + 1
    hist[n] = hist.get(n, 0) + 1
    near_hist[m] = near_hist.get(m, 0) + 1
print("success-ended episodes per step -> number of steps:", dict(sorted(hist.items())))
print("... of which with the hand within 7 cm of the object -> number of steps:", dict(sorted(near_hist.items())))
for thr, hh in late_hist.items():
    print("... of which the object was farther than %.2f from the goal before the step -> number of steps:" % thr, dict(sorted(hh.items())))
st = env.get_state()
d = (st[:, 18:21] - st[:, 31:34]).norm(dim=1)
print("envs with the object within 0.06 / 0.08 / 0.10 of the goal right now: %d / %d / %d of %d" % ((d < 0.06).sum(), (d < 0.08).sum(), (d < 0.10).sum(), E))
