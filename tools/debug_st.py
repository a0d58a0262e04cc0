#!/usr/bin/env python3
"""GPU-box diagnostic: StackTower device path vs the CPU oracle (random actions, then a scripted pick-and-stack)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gym_xarm_amd
from oracle import oracle as O

E = 64
env = gym_xarm_amd.make("XarmPDStackTower-v0", num_envs=E, seed=4, auto_reset=False)
orc = O.OracleStackTower(E, seed=4)
print("init diff", np.abs(env.get_state().cpu().numpy() - orc.get_state()).max())
obs = env.reset()
o2, a2, d2 = orc.reset()
print("reset: state diff", np.abs(env.get_state().cpu().numpy() - orc.get_state()).max(), "obs", np.abs(obs["observation"].cpu().numpy() - o2).max())
rng = np.random.default_rng(1)
for k in range(8):
    act = rng.uniform(-1, 1, (E, 8))
    orc.set_state(env.get_state().cpu().numpy().astype(np.float64))
    obs, rew, done, info = env.step(torch.tensor(act, dtype=torch.float32, device=env.device))
    o2, a2, d2, r2, dn2, s2 = orc.step(act)
    sd = np.abs(env.get_state().cpu().numpy() - orc.get_state())
    print(k, "state diff max %.2e (field %d)" % (sd.max(), sd.max(0).argmax()), "obs %.2e" % np.abs(obs["observation"].cpu().numpy() - o2).max(), "rew", np.abs(rew.cpu().numpy() - r2).max())
# scripted pick-and-stack on the device, env 0
env1 = gym_xarm_amd.make("XarmPDStackTower-v0", num_envs=1, seed=1, auto_reset=False)
s = env1.get_state()
s[0, O.ST_BP:O.ST_BP + 9] = torch.tensor([-0.2, 0, 0.025, 0.0, 0.1, 0.025, 0.2, -0.1, 0.025], device=env1.device)
env1.set_state(s)
o = env1.step(torch.zeros(1, 8, device=env1.device))[0]["observation"].cpu().numpy()
def servo(xy, z, g, n):
    global o
    for _ in range(n):
        hp = o[0, 39:42]
        a = np.zeros((1, 8), np.float32)
        a[0, 0:2] = np.clip((np.asarray(xy) - hp[:2]) / 0.0625, -1, 1); a[0, 2] = np.clip((z - hp[2]) / 0.0625, -1, 1); a[0, 3] = g
        o = env1.step(torch.from_numpy(a).to(env1.device))[0]["observation"].cpu().numpy()
t = time.time()
servo([-0.2, 0], 0.25, 1, 12); servo([-0.2, 0], 0.085, 1, 12); servo([-0.2, 0], 0.085, -1, 6); servo([-0.2, 0], 0.25, -1, 10)
print("lifted cube0:", o[0, 0:3])
servo([0.0, 0.1], 0.25, -1, 14); servo([0.0, 0.1], 0.139, -1, 10); servo([0.0, 0.1], 0.139, 1, 6); servo([0.0, 0.1], 0.3, 1, 8)
print("stacked: cube0", o[0, 0:3], "cube1", o[0, 3:6], "%.1fs" % (time.time() - t))
# throughput
for En in (8192,):
    e2 = gym_xarm_amd.make("XarmPDStackTower-v0", num_envs=En, seed=0)
    e2.reset()
    g = torch.Generator(device=e2.device); g.manual_seed(0)
    acts = [torch.rand(En, 8, device=e2.device, generator=g) * 2 - 1 for _ in range(8)]
    for i in range(3): e2.step(acts[i])
    torch.cuda.synchronize(); t = time.time()
    for i in range(20): e2.step(acts[i % 8])
    torch.cuda.synchronize(); dt = time.time() - t
    print("E=%d: %.3e env steps/s (%.1f ms/step)" % (En, En * 20 / dt, dt / 20 * 1e3))
