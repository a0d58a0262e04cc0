"""does an env's k_step result depend on whether a wave-mate holds a finger contact?  64 envs = one wavefront; env 5's object is
either where the reset left it (somewhere on the table) or teleported between the fingers; envs != 5 are compared."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd as gx
E = 64
for fam, kw in (("plain k_step", dict(step_coop_limit=-1)), ("k_step_fast", dict(step_coop_limit=1))):
    env = gx.make("XarmPDPickAndPlace-v0", num_envs=E, seed=9, auto_reset=False, reset_coop_limit=-1, **kw)
    env.reset()
    a = torch.rand(E, 4, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)) * 2 - 1
    a[:, 2] = 1.0
    for _ in range(6):
        env.step(a)                      # arms up, away from the objects
    base = env.get_state().clone()
    outs = []
    for grasp in (False, True):
        s = base.clone()
        if grasp:
            # put env 5's object between its fingers: at the hand position minus ~9 cm
            obs = env.step(torch.zeros(E, 4, device="cuda"))[0]["observation"]
            env.set_state(base)
            hand = obs[5, 0:3]
            s[5, 18:21] = hand + torch.tensor([0.0, 0.0, -0.075], device="cuda")
            s[5, 21:25] = torch.tensor([0, 0, 0, 1.0], device="cuda")
            s[5, 25:31] = 0
        env.set_state(s)
        env.step(a)
        outs.append(env.get_state().clone())
    d = (outs[0] - outs[1]).abs()
    others = [i for i in range(E) if i != 5]
    print(fam, "| env 5 touch", outs[1][5, 50].item(), "lam_p", outs[1][5, 42:46].tolist(), "| other envs changed:", int((d[others].max(dim=1).values > 0).sum()), "max", d[others].max().item(),
          "cols", (d[others] > 0).any(dim=0).nonzero()[:, 0].tolist()[:20])
    env.close()
