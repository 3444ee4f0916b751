#!/bin/bash
# the round-end GPU tiers in one call: the whole -m gpu suite, then smoke()
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/full_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/full_tests.log
grep -v amdgpu.ids gpurun_out/full_tests.log | tail -8
grep -q "tests rc=0" gpurun_out/full_tests.log || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3
