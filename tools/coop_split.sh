#!/bin/bash
# development aid: split of the cooperative tick into per-substep setup and the 50 sweeps - times every prebuilt
# library variant under gpurun_variants/ (built here with -DXC_SWEEP_ITERS=n) and restores the original
cp gym_xarm_amd/csrc/libxarm_hip.so /tmp/lib_orig.so
for f in gpurun_variants/lib_*.so; do
  cp $f gym_xarm_amd/csrc/libxarm_hip.so
  echo "== $f"
  timeout -k 10 120 python tools/coop_time.py 2>&1 | grep -v amdgpu | tail -2 || true
done
cp /tmp/lib_orig.so gym_xarm_amd/csrc/libxarm_hip.so
