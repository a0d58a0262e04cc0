#!/bin/bash
# Collect the rocprofv3 evidence set for profiles/ on a GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh r01e
# One --kernel-trace --stats pass, then three separate --pmc passes (FETCH_SIZE; WRITE_SIZE; SQ_*),
# never combined with the sys/hip/hsa trace domains.  Summaries land in gpurun_out/<tag>/.
set -e
TAG=${1:-prof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="$ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-lazy"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/stats -- python3 $BENCH > $OUT/stats.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -- python3 $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -- python3 $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --output-format csv --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY \
    -d $OUT/pmc_sq -- python3 $BENCH > $OUT/pmc_sq.log 2>&1
cd $ROOT
python3 tools/condense_stats.py $(ls $OUT/stats/*/*kernel_stats.csv $OUT/stats/*kernel_stats.csv 2>/dev/null | head -1) > $OUT/${TAG}_kernel_stats.csv
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq > $OUT/${TAG}_pmc_summary.json
# keep the merged-back scratch small
find $OUT -name '*_kernel_trace.csv' -size +4M -delete
find $OUT -name '*counter_collection.csv' -size +8M -delete
cat $OUT/${TAG}_kernel_stats.csv
