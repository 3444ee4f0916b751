#!/bin/bash
# round 3, first GPU call: the full-size parity tests, then the bench line with the new blocks
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_full_size_training.py -q -s -p no:cacheprovider > $OUT/r3a_full_tests.log 2>&1
rc=$?
tail -40 $OUT/r3a_full_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: stopping"; exit $rc; fi
timeout -k 10 700 python bench.py > $OUT/r3a_bench.json 2> $OUT/r3a_bench.err
rc2=$?
tail -c 3000 $OUT/r3a_bench.json
tail -5 $OUT/r3a_bench.err
exit $rc2
