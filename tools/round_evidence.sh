#!/bin/bash
# Round-end evidence on a GPU box (run through gpurun from the repo root, in two calls - a call is limited to 20 minutes):
#   tools/round_evidence.sh r04a suite     the whole -m gpu suite, smoke(), the bench lines quoted in DESIGN.md 6
#                                          (-> gpurun_out/<tag>_bench_lines.jsonl), the scripted-controller stage tables
#   tools/round_evidence.sh r04a profiles  one rocprofv3 set per workload (collect_profiles.sh)
set -e
TAG=${1:-r04a}
WHAT=${2:-suite}
if [ "$WHAT" = suite ]; then
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite_$TAG.log 2>&1 || { tail -30 gpurun_out/gpu_suite_$TAG.log; exit 1; }
tail -3 gpurun_out/gpu_suite_$TAG.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_$TAG.log 2>&1; tail -1 gpurun_out/smoke_$TAG.log
L=gpurun_out/${TAG}_bench_lines.jsonl
: > $L
python bench.py >> $L 2>> gpurun_out/${TAG}_bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline >> $L 2>> gpurun_out/${TAG}_bench.err
python bench.py --envs-per-gpu 16384 --no-cpu-baseline --no-lazy --no-lockstep >> $L 2>> gpurun_out/${TAG}_bench.err
python bench.py --envs-per-gpu 4096 --no-cpu-baseline --no-lazy --no-lockstep >> $L 2>> gpurun_out/${TAG}_bench.err
python bench.py --workload reach >> $L 2>> gpurun_out/${TAG}_bench.err
python bench.py --workload handover >> $L 2>> gpurun_out/${TAG}_bench.err
python bench.py --workload stack >> $L 2>> gpurun_out/${TAG}_bench.err
python bench.py --workload handover2 >> $L 2>> gpurun_out/${TAG}_bench.err
python - $L <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    a = d.get("aged_state", {})
    print(d['config']['workload'][:24], d['config']['envs_per_gpu'], d['steps'], '%.3e' % d['value'], '%.3f ms' % d['ms_per_step'],
          {k: round(v['avg_ms'], 3) for k, v in d['roofline']['kernels'].items()}, 'aged %.3e' % a.get('value', 0), d['config']['episode_phase'],
          'cpu %.3g' % d.get('cpu_baseline', {}).get('value', 0))
PY
python tools/stage_tables.py 4096 > gpurun_out/${TAG}_stages.jsonl 2>/dev/null; cat gpurun_out/${TAG}_stages.jsonl
else
for w in pnp handover stack reach; do tools/collect_profiles.sh ${TAG}_$w $w > gpurun_out/${TAG}_${w}_collect.log 2>&1; tail -9 gpurun_out/${TAG}_${w}_collect.log; done
tools/collect_profiles.sh ${TAG}_pnp4096 pnp --envs-per-gpu 4096 > gpurun_out/${TAG}_pnp4096_collect.log 2>&1; tail -7 gpurun_out/${TAG}_pnp4096_collect.log
fi
