#!/bin/bash
# round 3: width-1024 rating windows through layer-sequential split-precision scans -- parity, then the cfg5 window
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_full_size_training.py -q -m gpu -x -p no:cacheprovider -s -k "forward_window or cfg5_rating or stateless_window" > $OUT/r3p_tests.log 2>&1
rc=$?
grep -v amdgpu.ids $OUT/r3p_tests.log | grep -E "cfg5 rating|passed|failed|Error" | tail -8
if [ $rc -ne 0 ]; then echo "tests rc=$rc: stopping"; grep -v amdgpu.ids $OUT/r3p_tests.log | tail -30; exit $rc; fi
timeout -k 10 300 python tools/probe_rate_window.py cfg5 1 16 64
