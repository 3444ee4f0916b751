#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r2c_*.log
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "scan2 or gemm or env10 or env11 or env13" > gpurun_out/r2c_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2c_tests.log
grep -v amdgpu.ids gpurun_out/r2c_tests.log | tail -12
for cfg in "1024" "2048" "3072"; do
  KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py $cfg 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2c_perf.log
  KL_SCAN2_BF16=0 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py $cfg 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2c_perf.log
done
cat gpurun_out/r2c_perf.log
timeout -k 10 200 python tools/diag_scan2_err.py 2 512 64 2048 6 2>&1 | grep -v amdgpu.ids
