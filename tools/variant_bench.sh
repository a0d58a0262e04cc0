#!/bin/bash
# development aid: headline bench for selected library variants (restores the original)
cp gym_xarm_amd/csrc/libxarm_hip.so /tmp/lib_orig.so
for name in "$@"; do
  cp gpurun_variants/lib_$name.so gym_xarm_amd/csrc/libxarm_hip.so
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', 'value %.3e' % d['value'], 'ms/step %.2f' % d['ms_per_step'], 'k_step %.3f ms' % d['roofline']['kernel_avg_ms'])" || true
done
cp /tmp/lib_orig.so gym_xarm_amd/csrc/libxarm_hip.so
