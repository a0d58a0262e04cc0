#!/usr/bin/env python3
"""Counted floating-point operations per env step: builds tools/flopcount/count_flops.cpp twice -
(a) every predicated block executed, i.e. what a wavefront with at least one lane in finger contact runs for all its
lanes, (b) blocks skipped per environment, the algorithmic count - runs them on a 256-env x 20-step random rollout of
XarmPDPickAndPlace-v0 and merges the figures into profiles/flop_count.json and profiles/pmc_traffic.json (the file
bench.py reads for its `roofline.valu` entry)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "flopcount", "count_flops.cpp")


def run(extra):
    exe = "/tmp/xarm_count_flops"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wno-unknown-pragmas"] + extra + ["-o", exe, SRC])
    return json.loads(subprocess.check_output([exe, "256", "20"], text=True))


def main():
    full, algo = run([]), run(["-DXARM_HOST_ANY_PER_ENV"])
    out = {"workload": "XarmPDPickAndPlace-v0, 256 envs x 20 random steps after reset (tools/flopcount/count_flops.cpp)",
           "counting": "add, sub, mul, div, sqrt / sin / cos / atan2 / tanh = 1 each; a fused multiply-add = 2",
           "contact_path_per_env_step": full, "algorithmic_per_env_step": algo}
    path = os.path.join(ROOT, "profiles", "flop_count.json")
    json.dump(out, open(path, "w"), indent=1)
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    d = json.load(open(pmc)) if os.path.exists(pmc) else {}
    d["pnp_counted_flops_per_env_step"] = algo["flops_per_env_step"]
    d["pnp_counted_flops_per_env_step_contact_path"] = full["flops_per_env_step"]
    d["pnp_counted_flops_per_env_reset"] = algo["flops_per_env_reset"]
    json.dump(d, open(pmc, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    sys.exit(main())
