#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r2d_*.log
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "scan3" > gpurun_out/r2d_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2d_tests.log
grep -v amdgpu.ids gpurun_out/r2d_tests.log | tail -12
for cfg in "1024" "2048" "3072"; do
  for s3 in 0 1; do
    echo "--- B=$cfg KL_SCAN3=$s3" >> gpurun_out/r2d_perf.log
    KL_SCAN3=$s3 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py $cfg 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2d_perf.log
  done
done
KL_SCAN3=1 KL_SCAN2_PF=1 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py 3072 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2d_perf.log
cat gpurun_out/r2d_perf.log
for cfg in 2048 3072; do
  KL_SCAN3=1 timeout -k 10 120 python tools/probe_scan2_stamps.py $cfg 2>&1 | grep -v amdgpu.ids | head -12 >> gpurun_out/r2d_stamps.log
done
cat gpurun_out/r2d_stamps.log
