#!/bin/bash
# round 2: parity of the second-generation wide scans + timing at a few stream counts
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/r2a_*.log
timeout -k 5 60 tools/micro/ldsdma_high > gpurun_out/r2a_micro.log 2>&1; echo "micro rc=$?" >> gpurun_out/r2a_micro.log
cat gpurun_out/r2a_micro.log
grep -q "micro rc=0" gpurun_out/r2a_micro.log || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "scan2 or env10 or env11 or env12 or env13" > gpurun_out/r2a_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r2a_tests.log
grep -v "^  File\|^Extension" gpurun_out/r2a_tests.log | tail -25
timeout -k 10 200 python tools/diag_scan2_err.py 2 256 40 2048 4 2>&1 | grep -v amdgpu.ids; timeout -k 10 200 python tools/diag_scan2_err.py 2 512 64 2048 4 2>&1 | grep -v amdgpu.ids
for cfg in "1024 0" "1024 1" "2048 1" "3072 1" "2048 0"; do
  set -- $cfg
  KL_SCAN2=$2 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py $1 >> gpurun_out/r2a_perf.log 2>&1 || { echo "probe $cfg failed" >> gpurun_out/r2a_perf.log; break; }
done
KL_SCAN2=1 KL_SCAN2_ROWS=32 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py 2048 >> gpurun_out/r2a_perf.log 2>&1
KL_SCAN2=1 KL_SCAN2_PF=1 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py 2048 >> gpurun_out/r2a_perf.log 2>&1
grep -v "amdgpu.ids" gpurun_out/r2a_perf.log
