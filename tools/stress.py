"""Long random-action rollouts of every env on the GPU: finiteness and physical sanity of the state (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
CASES = [("XarmPDPickAndPlace-v0", 16384, 4, 300, None), ("XarmPDPickAndPlace-v0", 16384, 4, 150, dict(init_grasp_rate=1.0, reward_type="dense", goal_shape="ground")),
         ("XarmReach-v0", 16384, 4, 200, None), ("XarmPDHandover-v0", 8192, 8, 250, None), ("XarmPDStackTower-v0", 8192, 8, 250, None)]
bad = 0
for env_id, E, A, steps, cfg in CASES:
    env = gym_xarm_amd.make(env_id, num_envs=E, seed=7, config=cfg)
    obs = env.reset()
    g = torch.Generator(device=env.device); g.manual_seed(3)
    t0 = time.time()
    worst_v = 0.0
    nonfinite = 0
    for k in range(steps):
        a = torch.rand(E, A, device=env.device, generator=g) * 2.4 - 1.2     # beyond the clip range on purpose
        if k % 7 == 0: a = torch.sign(a)                                    # saturated actions
        obs, rew, done, info = env.step(a)
        o = obs["observation"]
        nonfinite += int((~torch.isfinite(o)).any(dim=1).sum()) + int((~torch.isfinite(rew)).sum())
        worst_v = max(worst_v, float(o.abs().max()))
    st = env.get_state()
    nonfinite += int((~torch.isfinite(st)).any(dim=1).sum())
    ag = obs["achieved_goal"]
    print("%-24s cfg=%s E=%d steps=%d: non-finite rows %d, max |obs| %.2f, achieved_goal z range [%.3f, %.3f], %.1fs" % (
        env_id, cfg, E, steps, nonfinite, worst_v, float(ag.reshape(E, -1, 3)[..., 2].min()), float(ag.reshape(E, -1, 3)[..., 2].max()), time.time() - t0), flush=True)
    bad += nonfinite
    env.close()
print("TOTAL non-finite rows:", bad)
sys.exit(1 if bad else 0)
