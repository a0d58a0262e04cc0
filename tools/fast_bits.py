"""Development aid: which envs of a batch keep the bits of the plain step kernel under the fast pipeline (PickAndPlace, Handover), step by step, next to the hand-off count - all but the handed-off ones in a build with -ffp-contract=on, a minority in the default build (gym_xarm_amd/build.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, gym_xarm_amd as gx
for env_id, A, E in (("XarmPDPickAndPlace-v0", 4, 8192), ("XarmPDHandover-v0", 8, 8192)):
    a = [torch.rand(E, A, device="cuda", generator=torch.Generator(device="cuda").manual_seed(70 + j)) * 2 - 1 for j in range(6)]
    def run(**kw):
        env = gx.make(env_id, num_envs=E, seed=31, auto_reset=False, reset_coop_limit=-1, **kw)
        env.reset(); sts = []; ho = []
        for j in range(6):
            env.step(a[j]); sts.append(env.get_state().clone()); ho.append(env.debug_counts()[1])
        env.close(); return sts, ho
    (fast, ho), (plain, _) = run(step_coop_limit=1), run(step_coop_limit=-1)
    same = torch.ones(E, dtype=torch.bool, device="cuda")
    for j in range(6):
        same &= (fast[j] == plain[j]).all(dim=1)
        print(env_id, "step", j, "envs with identical bits so far", int(same.sum()), "of", E, "handed off this step", ho[j])
