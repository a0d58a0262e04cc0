import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, gym_xarm_amd
from gym_xarm_amd.sb3_adapter import make_vec_env
for E in (4096, 65536):
    venv = make_vec_env("XarmPDPickAndPlace-v0", n_envs=E, seed=0)
    venv.reset()
    a = np.random.default_rng(0).uniform(-1, 1, (E, 4)).astype(np.float32)
    for _ in range(3): venv.step(a)
    t0 = time.perf_counter()
    n = 20
    for _ in range(n): venv.step(a)
    dt = time.perf_counter() - t0
    print("SB3VecEnv (numpy in/out, list-of-dict infos), %d envs: %.1f ms per step = %.3g env steps/s" % (E, 1e3 * dt / n, E * n / dt), flush=True)
    venv.close()
