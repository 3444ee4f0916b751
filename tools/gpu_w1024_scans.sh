#!/bin/bash
# round 3: the width-1024 scans (lstm_scan_w32.hip) -- parity tests, then cfg5 timing against the thin scans
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_full_size_training.py -q -m gpu -x -p no:cacheprovider -k "${KL_TESTS:-width_1024 or validation or cfg5}" > $OUT/r3k_tests.log 2>&1
rc=$?
grep -v amdgpu.ids $OUT/r3k_tests.log | tail -12
if [ $rc -ne 0 ]; then echo "tests rc=$rc: stopping"; exit $rc; fi
(KL_W32=0 timeout -k 10 300 python tools/probe_shapes.py cfg5 2>&1 | grep -E "^cfg" | sed "s/^/thin: /" || exit 1
KL_W32_LOCAL=1 timeout -k 10 300 python tools/probe_shapes.py cfg5 2>&1 | grep -E "^cfg" | sed "s/^/w32 local: /" || exit 1
timeout -k 10 300 python tools/probe_shapes.py cfg5 2>&1 | grep -E "^cfg" | sed 's/^/w32 write-through: /' || exit 1) | tee $OUT/r3k_shapes.log
