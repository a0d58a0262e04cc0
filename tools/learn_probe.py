import json, sys, time
sys.path.insert(0, '/root/repo')
from gym_xarm_amd.train import train
for kw in (dict(num_envs=2048, updates=3000, log_every=250), dict(num_envs=4096, updates=1500, log_every=150, n_steps=16), dict(num_envs=2048, updates=1500, log_every=150, lr=2e-3, gamma=0.95)):
    t = time.time()
    model, venv, hist = train("XarmPDHandoverNoGoal-v1", quiet=True, seed=0, **kw)
    print(kw, json.dumps([round(h["mean_raw_reward"], 4) for h in hist]), "succ", [round(h["success_rate"], 3) for h in hist][-3:], "%.0fs" % (time.time() - t), flush=True)
