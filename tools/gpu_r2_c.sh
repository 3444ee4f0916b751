#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r2c_*.log
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r2c_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2c_tests.log
tail -8 gpurun_out/r2c_tests.log
for cfg in "1024" "2048" "3072"; do
  KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py $cfg 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2c_perf.log
done
cat gpurun_out/r2c_perf.log
