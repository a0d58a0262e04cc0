#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per-kernel average of each counter."""
import csv, glob, collections, sys, json
out = {}
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = "k_step" if "k_step" in k else ("k_reset" if "k_reset" in k else None)
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in agg:
            for c, v in agg[k].items():
                out.setdefault(k, {})[c] = {"avg_per_launch": sum(v) / len(v), "launches": len(v)}
print(json.dumps(out, indent=1))
