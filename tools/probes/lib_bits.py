"""Development aid: do two builds of libxarm_hip.so compute the same bits?  Steps the cooperative kernels of PickAndPlace (4 096
envs) and Handover (2 048 envs) under random actions with the library named by XARM_HIP_LIB (or the product library) and writes
the states to <out>.npz; run it once per library and compare with `lib_bits.py --cmp a.npz b.npz`."""
import os, sys
import numpy as np
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        d = a[k] != b[k]
        print(k, "rows that differ: %d of %d, max |diff| %.3g" % (d.any(axis=-1).sum(), d.shape[0] * d.shape[1], np.abs(a[k] - b[k]).max()))
    sys.exit(0)
sys.path.insert(0, os.getcwd())
import torch, gym_xarm_amd as gx
out = {}
for name, env_id, E, A in (("pnp", "XarmPDPickAndPlace-v0", 4096, 4), ("handover", "XarmPDHandover-v0", 2048, 8)):
    env = gx.make(env_id, num_envs=E, seed=5)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(9)
    sts = []
    if name == "handover":      # contact states of the scripted hand-over (tests/golden), tiled: the pad rows and the coupled sweep are exercised from step 1
        gl = np.load(os.path.join(os.getcwd(), "tests", "golden", "handover_oracle_rollout.npz"))
        s0 = np.concatenate([gl["states"][t] for t in (14, 18, 22, 26, 30)])
        env.set_state(torch.tensor(np.tile(s0, (E // s0.shape[0] + 1, 1))[:E], dtype=torch.float32, device="cuda"))
    for t in range(60):
        env.step(torch.rand(E, A, device="cuda", generator=g) * 2 - 1)
        if t < 3 or t % 10 == 9:
            sts.append(env.get_state().cpu().numpy())
    out[name] = np.stack(sts)
    env.close()
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1])
