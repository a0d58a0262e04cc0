#!/bin/bash
# Collect the rocprofv3 evidence set for profiles/ on a GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r02d [pnp|reach|handover|stack] [extra bench.py flags]
# One --kernel-trace --stats pass, then three separate --pmc passes (FETCH_SIZE; WRITE_SIZE; <= 8 SQ_* counters),
# never combined with the sys/hip/hsa trace domains; every pass under its own timeout.  Summaries land in
# gpurun_out/<tag>/ (<tag>_kernel_stats.csv, <tag>_pmc_summary.json) and the per-launch HBM traffic / VALU counts of
# the step kernel are merged into profiles/pmc_traffic.json's copy under gpurun_out/<tag>/pmc_traffic.json.
set -e
TAG=${1:-prof}
WL=${2:-pnp}
shift || true; shift || true
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="$ROOT/bench.py --workload $WL --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-extras $*"
timeout -k 5 300 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -- python3 $BENCH > $OUT/stats.log 2>&1
timeout -k 5 300 rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -- python3 $BENCH > $OUT/pmc_fetch.log 2>&1
timeout -k 5 300 rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -- python3 $BENCH > $OUT/pmc_write.log 2>&1
timeout -k 5 300 rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY \
    -d $OUT/pmc_sq -- python3 $BENCH > $OUT/pmc_sq.log 2>&1
cd $ROOT
python3 tools/condense_stats.py $(ls $OUT/stats/*/*kernel_stats.csv $OUT/stats/*kernel_stats.csv 2>/dev/null | head -1) > $OUT/${TAG}_kernel_stats.csv
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq > $OUT/${TAG}_pmc_summary.json
python3 tools/update_pmc_traffic.py $OUT/${TAG}_pmc_summary.json $WL $OUT/stats.log $OUT/pmc_traffic.json
# keep the merged-back scratch small
find $OUT -name '*_kernel_trace.csv' -size +4M -delete
find $OUT -name '*counter_collection.csv' -size +8M -delete
cat $OUT/${TAG}_kernel_stats.csv
