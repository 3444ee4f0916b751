#!/bin/bash
# round 3: where the incremental step's kernels hand over to each other (rows), and the cfg5 topology
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
rm -f $OUT/r3s.log
for n in 16 32 64 80; do
  for m in 16 96; do
    echo "=== n=$n KL_INC_SMALL_MIN=$m" >> $OUT/r3s.log
    KL_INC_SMALL_MIN=$m KL_PROBE_PREC=3 timeout -k 10 120 python tools/probe_incremental.py $n 2>&1 | grep -v amdgpu.ids >> $OUT/r3s.log || exit 1
  done
done
for n in 1024 256; do
  for f in 1 0; do
    echo "=== n=$n KL_OUT_FUSED=$f" >> $OUT/r3s.log
    KL_OUT_FUSED=$f KL_PROBE_PREC=3 timeout -k 10 120 python tools/probe_incremental.py $n 2>&1 | grep -v amdgpu.ids >> $OUT/r3s.log || exit 1
  done
done
echo "=== n=192,224,255 small kernel; n=192.. through the tile kernel is not possible (KL_BIG_STEP_N)" >> $OUT/r3s.log
KL_PROBE_PREC=3 timeout -k 10 120 python tools/probe_incremental.py 192 255 2>&1 | grep -v amdgpu.ids >> $OUT/r3s.log || exit 1
echo "=== cfg5 topology (depth 4, width 1024, 2 contexts)" >> $OUT/r3s.log
for sw in "KL_INC_TILE=1 KL_INC_SMALL=1" "KL_INC_TILE=0 KL_INC_SMALL=0"; do
  echo "--- $sw" >> $OUT/r3s.log
  env $sw KL_PROBE_L=4 KL_PROBE_W=1024 KL_PROBE_C=2 KL_PROBE_STEPS=100 KL_PROBE_PREC=3 timeout -k 10 200 python tools/probe_incremental.py 128 1024 2>&1 | grep -v amdgpu.ids >> $OUT/r3s.log || exit 1
done
cat $OUT/r3s.log
