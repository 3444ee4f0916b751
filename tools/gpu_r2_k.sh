#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r2k_*.log
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "scan2 or env1" > gpurun_out/r2k_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2k_tests.log
grep -v amdgpu.ids gpurun_out/r2k_tests.log | tail -8
grep -q "rc=0" gpurun_out/r2k_tests.log || exit 1
for cfg in 1024 2048 3072; do
  for f in 0 1; do
    echo "--- B=$cfg KL_SCAN2_AHEAD=$f" >> gpurun_out/r2k_perf.log
    KL_SCAN2_AHEAD=$f KL_PROBE_TRACE=1 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py $cfg 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2k_perf.log
  done
done
cat gpurun_out/r2k_perf.log
timeout -k 10 120 python tools/probe_scan2_stamps.py 3072 2>&1 | grep -v amdgpu.ids | tail -13
