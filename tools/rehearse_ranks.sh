#!/bin/bash
# Rehearsal of bench.py's multi-rank path on a ONE-GPU box (run through gpurun): 2 ranks weak and 4 ranks strong, every
# rank on device 0, gloo instead of RCCL (two ranks cannot share a device under RCCL).  Checks the launch line the
# driver uses, the shard arithmetic and the max-over-ranks timing on real kernels; it says nothing about xGMI scaling.
set -e
export XARM_BENCH_DEVICE=0 XARM_BENCH_BACKEND=gloo
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 10 --warmup 3 --envs-per-gpu 8192 --no-extras > gpurun_out/bench_2rank_gloo.json 2> gpurun_out/bench_2rank_gloo.err
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29512 \
    bench.py --gpus 4 --steps 10 --warmup 3 --scaling strong --no-extras > gpurun_out/bench_4rank_gloo.json 2> gpurun_out/bench_4rank_gloo.err
python - <<'PY'
import json
for f in ("gpurun_out/bench_2rank_gloo.json", "gpurun_out/bench_4rank_gloo.json"):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(d["n_gpus"], d["scaling"], "%.3e" % d["value"], d["config"]["total_envs"], d["config"]["envs_per_gpu"], d["config"]["parallelism"])
PY
