#!/bin/bash
# A/B of one environment switch on one box: tests first, then the training step at three stream counts
# usage: bash tools/gpu_ab.sh <SWITCH> "<pytest -k expression>"
SW=$1
mkdir -p gpurun_out
rm -f gpurun_out/ab_$SW.log
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "$2" > gpurun_out/ab_${SW}_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/ab_${SW}_tests.log
grep -v amdgpu.ids gpurun_out/ab_${SW}_tests.log | tail -6
grep -q "tests rc=0" gpurun_out/ab_${SW}_tests.log || exit 1
for cfg in 1024 2048 3072; do
  for f in 0 1; do
    echo "--- B=$cfg $SW=$f" >> gpurun_out/ab_$SW.log
    env $SW=$f KL_PROBE_TRACE=1 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py $cfg 2>&1 | grep -v amdgpu.ids >> gpurun_out/ab_$SW.log
  done
done
cat gpurun_out/ab_$SW.log
