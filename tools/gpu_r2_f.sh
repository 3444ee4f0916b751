#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r2f_*.log
for cfg in 1024 1536 2048 3072; do
  for pfb in 0 1 2; do
    echo "--- B=$cfg PFB=$pfb" >> gpurun_out/r2f_perf.log
    KL_SCAN2_PFB=$pfb KL_PROBE_TRACE=1 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py $cfg 2>&1 | grep -v amdgpu.ids | grep -v fwd >> gpurun_out/r2f_perf.log
  done
  for pf in 0 1 2; do
    for rows in 16 32; do
      echo "--- B=$cfg PF=$pf rows=$rows" >> gpurun_out/r2f_perf.log
      KL_SCAN2_PFB=1 KL_SCAN2_PF=$pf KL_SCAN2_ROWS=$rows KL_PROBE_TRACE=1 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py $cfg 2>&1 | grep -v amdgpu.ids | grep fwd >> gpurun_out/r2f_perf.log
    done
  done
done
cat gpurun_out/r2f_perf.log
