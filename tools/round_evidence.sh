#!/bin/bash
# Round-end evidence on a GPU box (run through gpurun from the repo root): the whole -m gpu suite, smoke(), the bench
# lines quoted in DESIGN.md 6 (-> gpurun_out/r02g_bench_lines.jsonl) and the PMC set of the 4 096-env PickAndPlace case.
set -e
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite_r02g.log 2>&1 || { tail -30 gpurun_out/gpu_suite_r02g.log; exit 1; }
tail -3 gpurun_out/gpu_suite_r02g.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_r02g.log 2>&1; tail -1 gpurun_out/smoke_r02g.log
: > gpurun_out/r02g_bench_lines.jsonl
python bench.py >> gpurun_out/r02g_bench_lines.jsonl 2>> gpurun_out/r02g_bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline >> gpurun_out/r02g_bench_lines.jsonl 2>> gpurun_out/r02g_bench.err
python bench.py --envs-per-gpu 16384 --no-cpu-baseline --no-lazy >> gpurun_out/r02g_bench_lines.jsonl 2>> gpurun_out/r02g_bench.err
python bench.py --envs-per-gpu 4096 --no-cpu-baseline --no-lazy >> gpurun_out/r02g_bench_lines.jsonl 2>> gpurun_out/r02g_bench.err
python bench.py --workload reach >> gpurun_out/r02g_bench_lines.jsonl 2>> gpurun_out/r02g_bench.err
python bench.py --workload handover >> gpurun_out/r02g_bench_lines.jsonl 2>> gpurun_out/r02g_bench.err
python bench.py --workload stack >> gpurun_out/r02g_bench_lines.jsonl 2>> gpurun_out/r02g_bench.err
python - <<'PY'
import json
for l in open('gpurun_out/r02g_bench_lines.jsonl'):
    d=json.loads(l); print(d['config']['workload'][:24], d['config']['envs_per_gpu'], d['steps'], '%.3e'%d['value'], '%.3f ms'%d['ms_per_step'], d['roofline']['kernel'][:40], '%.3f'%d['roofline']['kernel_avg_ms'], d['config']['episode_phase'])
PY
tools/collect_profiles.sh r02g_pnp4096 pnp --envs-per-gpu 4096 > gpurun_out/r02g_pnp4096_collect.log 2>&1; tail -8 gpurun_out/r02g_pnp4096_collect.log
tools/collect_profiles.sh r02g_pnp pnp > gpurun_out/r02g_pnp_collect.log 2>&1; tail -8 gpurun_out/r02g_pnp_collect.log
tools/collect_profiles.sh r02g_reach reach > gpurun_out/r02g_reach_collect.log 2>&1; tail -6 gpurun_out/r02g_reach_collect.log
