#!/usr/bin/env python3
"""Merge the step kernel's per-launch HBM traffic and VALU instruction count of one collect_profiles.sh run into a
pmc_traffic.json (the file bench.py reads for `roofline.traffic`):
    update_pmc_traffic.py <tag>_pmc_summary.json <workload> <stats.log with the bench JSON line> <out.json>
HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: rocprofv3 reports both in KiB and FETCH_SIZE counts the 128-byte
requests of wide coalesced reads as 64 bytes on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import json
import os
import sys

# candidates per role, the first one present in the summary with counters wins (the fast pipeline's first kernel, the plain
# kernel when the pipeline is off, the cooperative family at small batch sizes)
STEP_KERNEL = {"pnp": ["k_step_fast_stage", "k_step_fast", "k_step", "k_step_coop"], "reach": ["k_reach_step", "k_reach_step_coop"],
               "handover": ["k_ho_step_fast", "k_ho_step"], "stack": ["k_st_step"], "handover2": ["k_ho2_step"]}
RESET_KERNEL = {"pnp": ["k_reset_coop", "k_reset"], "reach": ["k_reach_reset", "k_reach_reset_coop"],
                "handover": ["k_ho_reset_coop", "k_ho_reset"], "stack": ["k_st_reset"], "handover2": ["k_ho2_reset"]}
HANDOFF_KERNEL = {"pnp": ["k_step_coop_list_stage", "k_step_coop_list"], "handover": ["k_ho_step_coop_list"]}


def main():
    summ, wl, log, out = sys.argv[1:5]
    d = json.load(open(summ))
    line = [l for l in open(log) if l.startswith("{")][-1]
    E = json.loads(line)["config"]["envs_per_gpu"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = os.path.join(root, "profiles", "pmc_traffic.json")
    t = json.load(open(out)) if os.path.exists(out) else (json.load(open(base)) if os.path.exists(base) else {})
    def usable(k):
        return k in d and "FETCH_SIZE" in d[k] and "WRITE_SIZE" in d[k] and d[k].get("SQ_WAVE_CYCLES", {"avg_per_launch": 1})["avg_per_launch"] > 0
    for role, names in (("step", STEP_KERNEL), ("reset", RESET_KERNEL), ("handoff", HANDOFF_KERNEL)):
        ks = [k for k in names.get(wl, []) if usable(k)]
        if not ks:
            continue
        # a kernel launched beside its twin and out of its count range does nothing: the first candidate that moved a real share
        nbytes = {k: 2 * d[k]["FETCH_SIZE"]["avg_per_launch"] + d[k]["WRITE_SIZE"]["avg_per_launch"] for k in ks}
        k = [k for k in ks if nbytes[k] > 0.01 * max(nbytes.values())][0]
        hbm = (2 * d[k]["FETCH_SIZE"]["avg_per_launch"] + d[k]["WRITE_SIZE"]["avg_per_launch"]) * 1024
        t["%s_hbm_bytes_per_launch_%d" % (k, E)] = hbm
        if "SQ_INSTS_VALU" in d[k]:
            t["%s_valu_wave_insts_per_launch_%d" % (k, E)] = d[k]["SQ_INSTS_VALU"]["avg_per_launch"]
    t["source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes (tools/collect_profiles.sh); HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB, gfx950 FETCH correction per MI355X_MICROARCH.md"
    json.dump(t, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in t.items() if k.endswith("_%d" % E)}, indent=1))


if __name__ == "__main__":
    main()
