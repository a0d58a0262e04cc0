#!/bin/bash
# development aid: split of the cooperative tick into per-substep setup and the 50 sweeps - times every prebuilt
# library variant under gpurun_variants/ (built with -DXC_SWEEP_ITERS=n; xarm_version() then says "sweeps=n").
# A variant is loaded through XARM_HIP_LIB (gym_xarm_amd/_native.py); the product library is never touched.
set -e
for f in gpurun_variants/lib_*.so; do
  echo "== $f"
  XARM_HIP_LIB=$PWD/$f timeout -k 10 120 python tools/coop_time.py 2>&1 | grep -v amdgpu | tail -2 || true
done
