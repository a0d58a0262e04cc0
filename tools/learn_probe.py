"""A2C learning probes on the on-device driver (gym_xarm_amd/train.py): mean raw reward per step over training.
usage: learn_probe.py handover|pnp"""
import json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_xarm_amd.train import train
which = sys.argv[1] if len(sys.argv) > 1 else "handover"
if which == "handover":
    runs = [("XarmPDHandoverNoGoal-v1", None, dict(num_envs=2048, updates=3000, log_every=250))]
else:
    cfg = {"GUI": False, "num_obj": 1, "reward_type": "dense", "init_grasp_rate": 0.0, "goal_ground_rate": 0.0, "goal_shape": "air"}
    runs = [("XarmPDPickAndPlace-v0", cfg, dict(num_envs=4096, updates=1500, log_every=150)),
            ("XarmPDPickAndPlace-v0", cfg, dict(num_envs=4096, updates=1500, log_every=150, lr=2e-3, gamma=0.95))]
for env_id, cfg, kw in runs:
    t = time.time()
    model, venv, hist = train(env_id, config=cfg, quiet=True, seed=0, **kw)
    print(env_id, kw, json.dumps([round(h["mean_raw_reward"], 4) for h in hist]), "succ", [round(h["success_rate"], 3) for h in hist][-3:],
          "%.2e steps/s" % hist[-1]["env_steps_per_sec"], "%.0fs" % (time.time() - t), flush=True)
