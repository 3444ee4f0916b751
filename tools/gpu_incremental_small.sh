#!/bin/bash
# round 3: incremental step with step_small.hip -- parity, then timings (new kernels vs KL_INC_SMALL=0)
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_cfg3_full.py tests/test_rater_golden.py tests/test_generate_equivalence.py tests/test_wrapper_processor_gpu.py -q -m gpu -x -p no:cacheprovider -k "step_batch or peaked or state_dist or cfg3 or hip or generate or processor" > $OUT/r3e_tests.log 2>&1
rc=$?
grep -v amdgpu.ids $OUT/r3e_tests.log | tail -8
if [ $rc -ne 0 ]; then echo "tests rc=$rc: stopping"; exit $rc; fi
rm -f $OUT/r3e.log
for n in 128 64 32 256; do
  for sm in 1 0; do
    echo "=== n=$n KL_INC_SMALL=$sm" >> $OUT/r3e.log
    KL_INC_SMALL=$sm KL_PROBE_PREC=3 timeout -k 10 120 python tools/probe_incremental.py $n 2>&1 | grep -v amdgpu.ids >> $OUT/r3e.log || exit 1
  done
done
KL_PROBE_PREC=3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r3e_inc128_stats -- python3 tools/probe_incremental.py 128 > $OUT/r3e_inc128.log 2>&1
cp $OUT/r3e_inc128_stats/*/*_kernel_stats.csv $OUT/r3e_incremental_n128_kernel_stats.csv
head -8 $OUT/r3e_incremental_n128_kernel_stats.csv | cut -c1-150
cat $OUT/r3e.log
