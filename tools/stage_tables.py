#!/usr/bin/env python3
"""Per-stage success fractions of the scripted controllers (gym_xarm_amd/policies.py) on the HIP envs - the table in
DESIGN.md 1.  One JSON line per controller.  (The oracle's figures come from tests/test_policies.py.)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_xarm_amd as gx  # noqa: E402
from gym_xarm_amd.policies import lift_stages, handover_stages, HandoverReleasePolicy  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = gx.make("XarmPDPickAndPlace-v0", num_envs=E, seed=5, auto_reset=False)
print(json.dumps({"controller": "PickAndLiftPolicy", "envs": E, **lift_stages(env)}), flush=True)
env.close()
env = gx.make("XarmPDHandover-v0", num_envs=E, seed=11, auto_reset=False)
print(json.dumps({"controller": "ezpolicy (xarm_handover.py:404-446), 40 steps", "envs": E, **handover_stages(env, 40)}), flush=True)
print(json.dumps({"controller": "ezpolicy, 100 steps", "envs": E, **handover_stages(env, 100)}), flush=True)
print(json.dumps({"controller": "ezpolicy + release step, 60 steps", "envs": E, **handover_stages(env, 60, HandoverReleasePolicy(env))}), flush=True)
env.close()
