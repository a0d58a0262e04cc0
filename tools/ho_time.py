"""Handover step-kernel time in controlled states (development aid; with the -DXK_SWEEP_ITERS=n variants under
gpurun_variants/ it splits k_ho_step / k_ho2_step into per-substep setup and sweeps): arms parked above the table,
random actions, arms driven down onto the stick."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
E = 16384
def run(name, env, act_fn, n=10):
    env.timing_enable(True)
    for i in range(n): env.step(act_fn(i))
    torch.cuda.synchronize()
    ms, k = env.timing_read()
    env.timing_enable(False)
    print("%-46s step kernel %.3f ms" % (name, ms / k), flush=True)
for tag, kw in (("num_obj 1", dict()), ("num_obj 2", dict(config=dict(GUI=False, num_obj=2, same_side_rate=0.5, goal_shape="any", use_stand=False)))):
    env = gym_xarm_amd.make("XarmPDHandover-v0" if not kw else "XarmHandover-v0", num_envs=E, seed=0, auto_reset=False, **kw)
    env.reset()
    z = torch.zeros(E, 8, device=env.device)
    up = z.clone(); up[:, 2] = 1; up[:, 6] = 1
    for _ in range(6): env.step(up)
    run(tag + ": arms up, stick resting", env, lambda i: z)
    if kw:   # two sticks: how much of that is the stick/stick manifold (state layout: xarm_handover2_core.h G_BP = 38 ...)
        s0 = env.get_state()
        s = s0.clone()
        s[:, 38:41] = torch.tensor([-0.3, -0.2, 0.02], device=env.device); s[:, 41:44] = torch.tensor([0.3, 0.2, 0.02], device=env.device)
        s[:, 44:52] = torch.tensor([0., 0, 0, 1, 0, 0, 0, 1], device=env.device); s[:, 52:64] = 0
        env.set_state(s)
        run(tag + ": arms up, sticks far apart", env, lambda i: z, n=5)
        s[:, 40] = 5.0; s[:, 43] = 6.0; env.set_state(s)
        run(tag + ": arms up, sticks in free fall", env, lambda i: z, n=3)
    g = torch.Generator(device=env.device); g.manual_seed(0)
    acts = [torch.rand(E, 8, device=env.device, generator=g) * 2 - 1 for _ in range(8)]
    env.reset()
    run(tag + ": random actions (first 10 steps)", env, lambda i: acts[i % 8])
    run(tag + ": random actions (steps 10-30)", env, lambda i: acts[i % 8], n=20)
    env.close()
