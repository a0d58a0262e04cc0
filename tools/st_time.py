"""StackTower step-kernel time in controlled states (development aid): parked arms / random actions / towers."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
E = 8192
def run(name, env, act_fn, n=10):
    env.timing_enable(True)
    for i in range(n): env.step(act_fn(i))
    torch.cuda.synchronize()
    ms, k = env.timing_read()
    print("%-34s step kernel %.3f ms" % (name, ms / k), flush=True)
env = gym_xarm_amd.make("XarmPDStackTower-v0", num_envs=E, seed=0, auto_reset=False)
env.reset()
z = torch.zeros(E, 8, device=env.device)
run("parked arms, cubes on table", env, lambda i: z)
s = env.get_state()
s[:, 54:63] = torch.tensor([-0.25, 0.0, 0.025, 0.0, 0.0, 0.025, 0.25, 0.0, 0.025], device=env.device); s[:, 63:75] = torch.tensor([0., 0, 0, 1] * 3, device=env.device); s[:, 75:93] = 0
env.set_state(s)
run("parked arms, cubes far apart on table", env, lambda i: z, n=5)
s[:, 54:63] = torch.tensor([-0.25, 0.0, 0.025, 0.0, 0.0, 0.025, 0.0, 0.0, 0.075], device=env.device); env.set_state(s)
run("parked arms, one cube on another", env, lambda i: z, n=5)
s[:, 56] = 0.5; s[:, 59] = 0.8; s[:, 62] = 1.1; env.set_state(s)
run("parked arms, cubes in free fall", env, lambda i: z, n=3)
env.reset()
g = torch.Generator(device=env.device); g.manual_seed(0)
acts = [torch.rand(E, 8, device=env.device, generator=g) * 2 - 1 for _ in range(8)]
run("random actions (first 10 steps)", env, lambda i: acts[i % 8])
run("random actions (steps 10-30)", env, lambda i: acts[i % 8], n=20)
st = env.get_state()
print("pad warm-start impulses active in %.1f %% of envs" % (100 * float((st[:, 126:134].abs().sum(1) > 0).float().mean())))
down = torch.zeros(E, 8, device=env.device); down[:, 2] = -1; down[:, 6] = -1
run("arms driven to the lowest height", env, lambda i: down, n=10)
st = env.get_state()
print("pad warm-start impulses active in %.1f %% of envs" % (100 * float((st[:, 126:134].abs().sum(1) > 0).float().mean())))
