#!/bin/bash
# timing variants of the second-generation scans (wrong results by construction: timing only); KL_VARS = library suffixes
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
rm -f $OUT/r3d.log
for v in "" ${KL_VARS:-_var8 _var16 _var24 _var32 _var64 _var120}; do
  echo "=== lib libkeraslm_hip$v.so B=3072" >> $OUT/r3d.log
  KL_LIB=$PWD/ocrd_keraslm_amd/libkeraslm_hip$v.so KL_PROBE_TRACE=1 KL_PROBE_TRAIN_ONLY=1 timeout -k 10 120 python tools/probe_perf.py 3072 2>&1 | grep -v amdgpu.ids | grep -E "fwd|bwd|train" >> $OUT/r3d.log
done
cat $OUT/r3d.log
