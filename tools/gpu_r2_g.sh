#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r2g_*.log
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "scan2 or gemm or train_window" > gpurun_out/r2g_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2g_tests.log
grep -v amdgpu.ids gpurun_out/r2g_tests.log | tail -12
for cfg in 1024 2048 3072; do
  for f in 0 1; do
    echo "--- B=$cfg KL_FUSE_WG=$f" >> gpurun_out/r2g_perf.log
    KL_FUSE_WG=$f KL_PROBE_TRACE=1 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py $cfg 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2g_perf.log
  done
done
cat gpurun_out/r2g_perf.log
