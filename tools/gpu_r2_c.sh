#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r2c_*.log
timeout -k 10 1100 python -m pytest tests/test_ddp_hip.py tests/test_cfg3_full.py -q -m gpu -s > gpurun_out/r2c_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2c_tests.log
grep -v "amdgpu.ids" gpurun_out/r2c_tests.log | tail -40
