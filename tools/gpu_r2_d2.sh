#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r2d2_*.log
for lib in libkeraslm_hip_stamps.so libkeraslm_hip_stamps_t64.so libkeraslm_hip_stamps_t576.so; do
  echo "=== $lib" >> gpurun_out/r2d2_stamps.log
  KL_STAMPS_LIB=$lib KL_SCAN3=1 timeout -k 10 120 python tools/probe_scan2_stamps.py 3072 2>&1 | grep -v amdgpu.ids | head -14 >> gpurun_out/r2d2_stamps.log
done
cat gpurun_out/r2d2_stamps.log
