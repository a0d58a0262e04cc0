#!/bin/bash
# development aid: time k_step for every prebuilt library variant under gpurun_variants/ (restores the original)
cp gym_xarm_amd/csrc/libxarm_hip.so /tmp/lib_orig.so
for f in gpurun_variants/lib_*.so; do
  cp $f gym_xarm_amd/csrc/libxarm_hip.so
  echo "== $f"
  timeout -k 10 120 python tools/tick_time.py 2>&1 | grep -v amdgpu | tail -4 || true
done
cp /tmp/lib_orig.so gym_xarm_amd/csrc/libxarm_hip.so
