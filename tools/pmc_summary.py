#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel average of each counter."""
import csv, glob, collections, sys, json
out = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            m = __import__("re").search(r"::(k_[a-z_0-9]+)[<(]", k)
            k = m.group(1) if m else None
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in agg:
            for c, v in agg[k].items():
                # the two reset kernels are launched side by side and the one out of its count range exits at once:
                # average over the launches that did work (counter above 1 % of the kernel's maximum)
                if ("_reset" in k or k in ("k_step", "k_step_coop_list", "k_step_from_stage", "k_step_coop_list_stage", "k_ho_step", "k_ho_step_coop_list")) and max(v) > 0:
                    v = [x for x in v if x > 0.01 * max(v)]
                out.setdefault(k, {})[c] = {"avg_per_launch": sum(v) / len(v), "launches": len(v)}
print(json.dumps(out, indent=1))
