"""Soak of the cooperative kernels: 3000 steps at 4096 envs for PickAndPlace (k_step_coop / k_reset_coop) and Reach."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, gym_xarm_amd
for env_id, A, n in (("XarmPDPickAndPlace-v0", 4, 3000), ("XarmReach-v0", 4, 6000)):
    E = 4096
    env = gym_xarm_amd.make(env_id, num_envs=E, seed=13)
    env.reset()
    g = torch.Generator(device=env.device); g.manual_seed(7)
    ring = [torch.rand(E, A, device=env.device, generator=g) * 2 - 1 for _ in range(32)]
    bad = 0
    t0 = time.perf_counter()
    for k in range(n):
        obs, rew, done, info = env.step(ring[k % 32])
        if k % 500 == 499:
            torch.cuda.synchronize()
            fin = bool(torch.isfinite(obs["observation"]).all())
            bad += 0 if fin else 1
            print("%s steps %4d: %.3e env steps/s, finite %s, max|obs| %.1f" % (env_id, k + 1, E * 500 / (time.perf_counter() - t0), fin, float(obs["observation"].abs().max())), flush=True)
            t0 = time.perf_counter()
    st = env.get_state()
    assert torch.isfinite(st).all() and not bad
    env.close()
print("soak ok")
