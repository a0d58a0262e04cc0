"""Extended distribution parity, HIP (strict auto-reset) vs the CPU oracle with the same reset rule: 8192 envs x 60 steps
of XarmPDPickAndPlace-v0 from the same seeds and actions, through episode boundaries (development aid; the unit tests
hold the smaller versions)."""
import sys, os, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, gym_xarm_amd
from oracle import oracle as O
E, T, W = 8192, 60, 16
env = gym_xarm_amd.make("XarmPDPickAndPlace-v0", num_envs=E, seed=321)
env.reset()
acts = [torch.rand(E, 4, generator=torch.Generator().manual_seed(900 + t)) * 2 - 1 for t in range(T)]
dones_dev, succ_dev, rew_dev = [], [], []
for t in range(T):
    obs, rew, done, info = env.step(acts[t])
    dones_dev.append(int(done.sum())); succ_dev.append(int(info["is_success"].sum())); rew_dev.append(float(rew.sum()))
dev = env.get_state().cpu().numpy().astype(np.float64)
shards = [O.OraclePnP(E // W, seed=321, env_id_offset=k * (E // W)) for k in range(W)]
a_np = [a.numpy().astype(np.float64) for a in acts]
def run(k):
    s = shards[k]; s.reset(); dn, sc, rw = [], [], []
    for t in range(T):
        o = s.step(a_np[t][k * (E // W):(k + 1) * (E // W)])
        d = o[4].astype(np.uint8)
        dn.append(int(d.sum())); sc.append(int(o[5].sum())); rw.append(float(o[3].sum()))
        if d.any(): s.reset(mask=d)
    return s.state.copy(), dn, sc, rw
t0 = time.time()
with ThreadPoolExecutor(W) as ex: res = list(ex.map(run, range(W)))
ora = np.concatenate([r[0] for r in res])
dn = np.sum([r[1] for r in res], axis=0); sc = np.sum([r[2] for r in res], axis=0)
print("oracle time %.1fs" % (time.time() - t0))
print("episodes finished  HIP %d  oracle %d" % (sum(dones_dev), dn.sum()))
print("successes          HIP %d  oracle %d" % (sum(succ_dev), sc.sum()))
print("dones at the 50-step limit (step 50): HIP %d oracle %d" % (dones_dev[49], dn[49]))
def stats(s):
    return dict(on_table=float(((s[:, 20] > 0.02) & (s[:, 20] < 0.1)).mean()), fell=float((s[:, 20] < -0.05).mean()), lifted=float((s[:, 20] > 0.1).mean()),
                touch=float(s[:, 50].mean()), med_x=float(np.median(s[:, 18])), med_absy=float(np.median(np.abs(s[:, 19]))),
                finger=float(s[:, 7].mean()), mean_abs_qd=float(np.abs(s[:, 9:16]).mean()), episode=float(s[:, 53].mean()), steps=float(s[:, 52].mean()))
sd, so = stats(dev), stats(ora)
for k in sd: print("%-12s HIP %.4f  oracle %.4f" % (k, sd[k], so[k]))
