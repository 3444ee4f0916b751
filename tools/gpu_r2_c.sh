#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_rescore_shard.py tests/test_rater_golden.py tests/test_rater_plumbing.py tests/test_cli.py tests/test_ddp_hip.py -q -m gpu > gpurun_out/r2c_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r2c_tests.log
grep -v amdgpu.ids gpurun_out/r2c_tests.log | tail -12
