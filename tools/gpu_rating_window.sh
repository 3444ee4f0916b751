#!/bin/bash
# round 3: the rating window's split-precision scan with data-sentinel hand-offs -- parity tests, then timing both ways
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_cfg3_full.py tests/test_generate_equivalence.py tests/test_rater_plumbing.py -q -m gpu -x -p no:cacheprovider -k "forward_window or stateless or cfg3 or generate or rater or random_shapes" > $OUT/r3h_tests.log 2>&1
rc=$?
grep -v amdgpu.ids $OUT/r3h_tests.log | tail -8
if [ $rc -ne 0 ]; then echo "tests rc=$rc: stopping"; exit $rc; fi
for B in 1 16 64; do
  KL_SPLIT_SENTINEL=0 timeout -k 10 120 python tools/probe_rate_window.py $B 2>&1 | grep "rating window" | sed 's/^/counters: /' || exit 1
  KL_SPLIT8=0 timeout -k 10 120 python tools/probe_rate_window.py $B 2>&1 | grep "rating window" | sed 's/^/sentinels, 16 units per workgroup: /' || exit 1
  timeout -k 10 120 python tools/probe_rate_window.py $B 2>&1 | grep "rating window" | sed 's/^/sentinels, 8 units where they apply: /' || exit 1
done | tee $OUT/r3h_rate.log
