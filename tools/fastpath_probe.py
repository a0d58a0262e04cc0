"""k_step time of a batch in which no wavefront holds a finger contact (every arm driven up and kept there) against the
usual random-action batch: how much of k_step is the wave-level price of the ~2 % of envs with pad rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
E = 65536
for name in ("random", "arms_up"):
    env = gym_xarm_amd.make("XarmPDPickAndPlace-v0", num_envs=E, seed=0, auto_reset=False)
    env.reset()
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = [torch.rand(E, 4, device="cuda", generator=g) * 2 - 1 for _ in range(16)]
    if name == "arms_up":
        for a in acts:
            a[:, 2] = 1.0
            a[:, 3] = 1.0
    for i in range(30):
        env.step(acts[i % 16])
    torch.cuda.synchronize()
    env.timing_enable(True)
    for i in range(20):
        env.step(acts[i % 16])
    torch.cuda.synchronize()
    ms, n = env.timing_read()
    st = env.get_state()
    print(name, "k_step %.3f ms" % (ms / n), "envs with pad impulse %.4f" % (st[:, 42:46] > 0).any(dim=1).float().mean().item(),
          "touch %.4f" % (st[:, 50] > 0).float().mean().item(), flush=True)
    env.close()
