#!/bin/bash
# round 3: backward scan with inputs a block ahead -- parity first, then timings (flags vs sentinels)
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -p no:cacheprovider -k "scan2 or handoff or flag or consecutive or trajectory" > $OUT/r3b_tests.log 2>&1
rc=$?
grep -v amdgpu.ids $OUT/r3b_tests.log | tail -15
if [ $rc -ne 0 ]; then echo "tests rc=$rc: stopping"; exit $rc; fi
timeout -k 10 400 python -m pytest tests/test_full_size_training.py -q -s -p no:cacheprovider -k "cfg2" > $OUT/r3b_full.log 2>&1
rc=$?
grep -v amdgpu.ids $OUT/r3b_full.log | tail -14
if [ $rc -ne 0 ]; then echo "full-size rc=$rc: stopping"; exit $rc; fi
rm -f $OUT/r3b_perf.log
for cfg in 3072 2048 1536 1024; do
  for f in 1 0; do
    echo "--- B=$cfg KL_SCAN2_FLAGS=$f" >> $OUT/r3b_perf.log
    KL_SCAN2_FLAGS=$f KL_PROBE_TRACE=1 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py $cfg 2>&1 | grep -v amdgpu.ids >> $OUT/r3b_perf.log || exit 1
  done
done
cat $OUT/r3b_perf.log
