#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r2j_*.log
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or train_window or step" > gpurun_out/r2j_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2j_tests.log
grep -v amdgpu.ids gpurun_out/r2j_tests.log | tail -8
S="786432,2048,512,1 786432,256,512,0 786432,512,2048,1 786432,512,256,1"
KL_GEMM_PERS=0 timeout -k 10 200 python tools/probe_gemm.py $S 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2j.log
for cfg in 1024 3072; do
  KL_PROBE_TRACE=1 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py $cfg 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2j.log
done
cat gpurun_out/r2j.log
