#!/bin/bash
# development aid: time k_step for every prebuilt library variant under gpurun_variants/.  A variant is loaded through
# XARM_HIP_LIB (gym_xarm_amd/_native.py); the product library in gym_xarm_amd/csrc/ is never touched.
set -e
for f in gpurun_variants/lib_*.so; do
  echo "== $f"
  XARM_HIP_LIB=$PWD/$f timeout -k 10 120 python tools/tick_time.py 2>&1 | grep -v amdgpu | tail -4 || true
done
