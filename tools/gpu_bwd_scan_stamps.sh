#!/bin/bash
# round 3: where does the new backward scan spend its time?  parity, stamps, timing variants (KL_BWD_VAR builds)
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
rm -f $OUT/r3c.log
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -p no:cacheprovider -k "scan2 or flag" > $OUT/r3c_tests.log 2>&1
rc=$?
grep -v amdgpu.ids $OUT/r3c_tests.log | tail -5
if [ $rc -ne 0 ]; then echo "tests rc=$rc: stopping"; exit $rc; fi
echo "=== stamps B=3072" >> $OUT/r3c.log
KL_LIB=ocrd_keraslm_amd/libkeraslm_hip_stamps.so timeout -k 10 200 python tools/probe_scan2_stamps.py 3072 2>&1 | grep -v amdgpu.ids >> $OUT/r3c.log || exit 1
for e in "KL_RT_LOCAL=1" "KL_RT_LOCAL=0" "KL_REGTILE=0"; do
  echo "=== B=3072 $e" >> $OUT/r3c.log
  env $e KL_PROBE_TRACE=1 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py 3072 2>&1 | grep -v amdgpu.ids >> $OUT/r3c.log || exit 1
done
cat $OUT/r3c.log
