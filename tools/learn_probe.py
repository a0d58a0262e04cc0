import json, sys
sys.path.insert(0, '/root/repo')
from gym_xarm_amd.train import train
model, venv, hist = train("XarmPDHandoverNoGoal-v1", num_envs=2048, updates=240, log_every=40, quiet=True, seed=0)
print(json.dumps([round(h["mean_raw_reward"], 5) for h in hist]), [h["env_steps_per_sec"] for h in hist][-1])
