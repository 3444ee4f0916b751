#!/bin/bash
# round 3: incremental step at 256 hypotheses and more with step_tile.hip -- parity, then timings against the gather + GEMM
# path (KL_INC_TILE=0) and the timing variants of the kernel (KL_TILE_VAR: 1 = no epilogue, 2 = no main loop)
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_rater_golden.py -q -m gpu -x -p no:cacheprovider -k "step_batch or peaked or hip" > $OUT/r3q_tests.log 2>&1
rc=$?
grep -v amdgpu.ids $OUT/r3q_tests.log | tail -8
if [ $rc -ne 0 ]; then echo "tests rc=$rc: stopping"; exit $rc; fi
rm -f $OUT/r3q.log
for n in 1024 256 2048; do
  for sm in 1 0; do
    echo "=== n=$n KL_INC_TILE=$sm" >> $OUT/r3q.log
    KL_INC_TILE=$sm KL_PROBE_PREC=3 timeout -k 10 120 python tools/probe_incremental.py $n 2>&1 | grep -v amdgpu.ids >> $OUT/r3q.log || exit 1
  done
done
for n in 1024 128; do
  echo "=== n=$n KL_OUT_FUSED=0" >> $OUT/r3q.log
  KL_OUT_FUSED=0 KL_PROBE_PREC=3 timeout -k 10 120 python tools/probe_incremental.py $n 2>&1 | grep -v amdgpu.ids >> $OUT/r3q.log || exit 1
  echo "=== n=$n KL_OUT_FUSED=1" >> $OUT/r3q.log
  KL_PROBE_PREC=3 timeout -k 10 120 python tools/probe_incremental.py $n 2>&1 | grep -v amdgpu.ids >> $OUT/r3q.log || exit 1
done
for v in 1 2; do
  echo "=== n=1024 KL_TILE_VAR=$v" >> $OUT/r3q.log
  KL_TILE_VAR=$v KL_PROBE_PREC=3 timeout -k 10 120 python tools/probe_incremental.py 1024 2>&1 | grep -v amdgpu.ids >> $OUT/r3q.log || exit 1
done
KL_PROBE_PREC=3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r3q_inc1024_stats -- python3 tools/probe_incremental.py 1024 > $OUT/r3q_inc1024.log 2>&1
cp $OUT/r3q_inc1024_stats/*/*_kernel_stats.csv $OUT/r3q_incremental_n1024_kernel_stats.csv
head -8 $OUT/r3q_incremental_n1024_kernel_stats.csv | cut -c1-150
cat $OUT/r3q.log
