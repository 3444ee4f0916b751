#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r2e_*.log
for cfg in 1024 2048 2560 3072; do
  KL_PROBE_TRACE=1 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py $cfg 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2e_perf.log
done
for pfb in 0 1 2; do
  echo "--- 3072 PFB=$pfb" >> gpurun_out/r2e_perf.log
  KL_SCAN2_PFB=$pfb KL_PROBE_TRACE=1 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py 3072 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2e_perf.log
done
cat gpurun_out/r2e_perf.log
for cfg in 2048 3072; do
  timeout -k 10 120 python tools/probe_scan2_stamps.py $cfg 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2e_stamps.log
done
cat gpurun_out/r2e_stamps.log
