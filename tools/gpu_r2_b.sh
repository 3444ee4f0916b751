#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
rm -f gpurun_out/r2b_*.log
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "scan2 and not 256" > gpurun_out/r2b_tests.log 2>&1
rc=$?
echo "tests rc=$rc" >> gpurun_out/r2b_tests.log
grep -v "^  File\|^Extension" gpurun_out/r2b_tests.log | tail -8
[ $rc -eq 0 ] || exit 1
for cfg in "2048 16 2" "2048 32 1" "2048 32 0"; do
  set -- $cfg
  echo "=== B=$1 rows=$2 pf=$3" >> gpurun_out/r2b_stamps.log
  KL_SCAN2_ROWS=$2 KL_SCAN2_PF=$3 timeout -k 10 150 python tools/probe_scan2_stamps.py $1 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2b_stamps.log || { echo "failed" >> gpurun_out/r2b_stamps.log; break; }
done
cat gpurun_out/r2b_stamps.log
for cfg in "2048 16 2" "2048 32 1" "3072 32 2" "1024 16 1"; do
  set -- $cfg
  echo "=== perf B=$1 rows=$2 pf=$3" >> gpurun_out/r2b_perf.log
  KL_SCAN2_ROWS=$2 KL_SCAN2_PF=$3 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py $1 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2b_perf.log
done
cat gpurun_out/r2b_perf.log
