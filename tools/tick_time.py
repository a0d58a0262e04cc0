"""k_step time with and without finger/object proximity (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd
E = 65536
env = gym_xarm_amd.make("XarmPDPickAndPlace-v0", num_envs=E, seed=0, auto_reset=False)
env.reset()
for _ in range(20):
    env.step(torch.zeros(E, 4, device="cuda"))
st = env.get_state()
def timeit(tag, state, acts, n=10):
    env.set_state(state)
    env.step(acts); torch.cuda.synchronize()
    env.set_state(state)
    env.timing_enable(True)
    for _ in range(n):
        env.step(acts)
    torch.cuda.synchronize()
    ms, k = env.timing_read()
    env.timing_enable(False)
    print("%-40s k_step %.3f ms" % (tag, ms / k), flush=True)
zero = torch.zeros(E, 4, device="cuda")
rnd = torch.rand(E, 4, device="cuda") * 2 - 1
timeit("settled, zero actions (as is)", st, zero)
far = st.clone(); far[:, 18] = 0.45; far[:, 19] = 0.28; far[:, 20] = 0.04; far[:, 21:24] = 0; far[:, 24] = 1; far[:, 25:31] = 0
timeit("objects parked away from the gripper, zero", far, zero)
timeit("objects parked away, random actions", far, rnd)
air = far.clone(); air[:, 20] = 5.0
timeit("objects in free fall (no table rows), zero", air, zero)
timeit("settled, random actions", st, rnd)
