"""PickAndPlace at the headline size for N steps (desynchronised episodes, random actions): target for rocprofv3 runs that\ncompare per-kernel durations right after reset() (N = 150) with the aged state (N = 1500).  usage: aged_run.py N"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
E = 65536; N = int(sys.argv[1])
env = gym_xarm_amd.make("XarmPDPickAndPlace-v0", num_envs=E, seed=0)
env.reset()
env.set_episode_steps(torch.arange(E, device=env.device) % 50)
g = torch.Generator(device=env.device); g.manual_seed(0)
ring = [torch.rand(E, 4, device=env.device, generator=g) * 2 - 1 for _ in range(16)]
for t in range(N): env.step(ring[t % 16])
torch.cuda.synchronize()
