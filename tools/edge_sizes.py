"""Odd env counts for every env (1, 33, 1000, 20000): reset + a few steps, finiteness, and shard consistency."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
ok = True
for env_id, A in (("XarmPDPickAndPlace-v0", 4), ("XarmReach-v0", 4), ("XarmPDHandover-v0", 8), ("XarmPDStackTower-v0", 8)):
    ref = None
    for E in (1, 33, 1000, 20000):
        env = gym_xarm_amd.make(env_id, num_envs=E, seed=5)
        obs = env.reset()
        g = torch.Generator(device=env.device); g.manual_seed(1)
        for k in range(3):
            a = torch.rand(20000, A, device=env.device, generator=g)[:E] * 2 - 1
            obs, rew, done, info = env.step(a)
        fin = bool(torch.isfinite(obs["observation"]).all())
        o0 = obs["observation"][0].clone()
        if ref is None: ref = o0
        same = bool(torch.equal(o0, ref))      # env 0 must not depend on the batch size
        print("%-22s E=%5d finite %s env0 identical across batch sizes %s" % (env_id, E, fin, same), flush=True)
        ok = ok and fin and same
        env.close()
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
