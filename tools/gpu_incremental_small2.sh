#!/bin/bash
# round 3: incremental step below 256 hypotheses with fragment-major weights -- parity, then timings
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_rater_golden.py tests/test_generate_equivalence.py -q -m gpu -x -p no:cacheprovider -k "step_batch or peaked or hip or generate" > $OUT/r3r_tests.log 2>&1
rc=$?
grep -v amdgpu.ids $OUT/r3r_tests.log | tail -8
if [ $rc -ne 0 ]; then echo "tests rc=$rc: stopping"; exit $rc; fi
rm -f $OUT/r3r.log
for n in 128 96 200 64 32; do
  for f in 1 0; do
    echo "=== n=$n KL_OUT_FUSED=$f" >> $OUT/r3r.log
    KL_OUT_FUSED=$f KL_PROBE_PREC=3 timeout -k 10 120 python tools/probe_incremental.py $n 2>&1 | grep -v amdgpu.ids >> $OUT/r3r.log || exit 1
  done
done
KL_OUT_FUSED=0 KL_PROBE_PREC=3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r3r_inc128_stats -- python3 tools/probe_incremental.py 128 > $OUT/r3r_inc128.log 2>&1
cp $OUT/r3r_inc128_stats/*/*_kernel_stats.csv $OUT/r3r_incremental_n128_kernel_stats.csv
head -6 $OUT/r3r_incremental_n128_kernel_stats.csv | cut -c1-150
cat $OUT/r3r.log
