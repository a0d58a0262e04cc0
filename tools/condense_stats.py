#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_stats.csv into a short table (our kernels by name, everything
else summed) for profiles/."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel,calls,total_ms,avg_us,min_us,max_us,percent")
other = [0, 0.0]
for r in rows:
    n = r["Name"]
    short = None
    m = __import__("re").search(r"::(k_[a-z_0-9]+)[<(]", n)      # template kernels: k_ho_step<xh::HandoverScene>(...)
    if m:
        short = m.group(1)
    if short is None:
        other[0] += int(r["Calls"]); other[1] += float(r["TotalDurationNs"]); continue
    print("%s,%s,%.3f,%.1f,%.1f,%.1f,%.2f" % (short, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                             float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("other(torch elementwise),%d,%.3f,,,,%.2f" % (other[0], other[1] / 1e6, 100 * other[1] / tot))
