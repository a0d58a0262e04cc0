#!/bin/bash
# rocprofv3 evidence for the StackTower kernels; at most 8 SQ counters per pass (a 9th aborts rocprofv3 with
# 'Request exceeds the capabilities of the hardware' and leaves it hanging), every pass under its own timeout (run through gpurun from the repo root): tools/collect_st_profiles.sh r01f
set -e
TAG=${1:-prof_st}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
B="$ROOT/tools/st_bench.py 8192 10"
timeout -k 5 200 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -- python3 $B > $OUT/stats.log 2>&1
timeout -k 5 200 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY -d $OUT/pmc_sq -- python3 $B > $OUT/pmc_sq.log 2>&1
timeout -k 5 200 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d $OUT/pmc_lds -- python3 $B > $OUT/pmc_lds.log 2>&1 || true
cd $ROOT
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = {}
for f in glob.glob(sys.argv[1] + "/pmc_*/*/*counter_collection.csv"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "k_st_step" if "k_st_step" in k else ("k_st_reset" if "k_st_reset" in k else None)
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in agg:
        for c, v in agg[k].items(): out.setdefault(k, {})[c] = {"avg_per_launch": sum(v) / len(v), "launches": len(v)}
json.dump(out, open(sys.argv[1] + "/pmc_summary.json", "w"), indent=1)
for k in out: print(k, {c: round(v["avg_per_launch"]) for c, v in out[k].items()})
PY
f=$(ls $OUT/stats/*/*kernel_stats.csv | head -1)
python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_st_" in r["Name"]: print(r["Name"].split("(")[0][-12:], r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3)
PY
find $OUT -name '*_kernel_trace.csv' -size +4M -delete
