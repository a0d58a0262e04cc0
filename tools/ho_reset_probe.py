"""Development aid: the cooperative Handover reset forced through the coupled sweep (XARM_HO_FORCE_COUPLED=1) against the natural one, from identical states - the probe that found the v_permlane32_swap read-after-write hazard (xarm_k_handover_coop.hip SwapXchg)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, numpy as np, gym_xarm_amd as gx
E = 16384
acts = [torch.rand(E, 8, device="cuda", generator=torch.Generator(device="cuda").manual_seed(k)) * 2 - 1 for k in range(6)]
env = gx.make("XarmPDHandover-v0", num_envs=E, seed=6, auto_reset=False)
os.environ["XARM_HO_FORCE_COUPLED"] = "1"
forced = gx.make("XarmPDHandover-v0", num_envs=E, seed=6, auto_reset=False)
os.environ.pop("XARM_HO_FORCE_COUPLED")
env.reset()
for k in range(4): env.step(acts[k])
st0 = env.get_state().clone()
m = torch.zeros(E, dtype=torch.uint8, device="cuda"); m[::5] = 1     # 3277 envs -> cooperative reset
env.reset(mask=m); a = env.get_state().clone()
forced.set_state(st0); forced.reset(mask=m); b = forced.get_state().clone()
d = (a != b).any(dim=1)
print("forced reset differs in", int(d.sum()), "envs of", int(m.sum()))
bad = torch.nonzero(d).flatten()[:6].tolist()
np.save("gpurun_out/ho_reset_bad.npy", np.concatenate([st0[bad].cpu().numpy(), a[bad].cpu().numpy(), b[bad].cpu().numpy()]))
for e in bad:
    cols = torch.nonzero(a[e] != b[e]).flatten().tolist()
    print("env", e, "cols", cols[:14], "maxdiff %.3g" % float((a[e]-b[e]).abs().max()), "touch0", st0[e,70:72].tolist(), "lam_p0", [round(v,3) for v in st0[e,62:70].tolist()])
