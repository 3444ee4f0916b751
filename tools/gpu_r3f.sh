#!/bin/bash
# round 3: where does a cfg5 training step (depth 4, width 1024, length 512, 2 contexts, 512 streams) spend its time?
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r3f_cfg5_stats -- python3 tools/probe_cfg5.py 512 > $OUT/r3f_cfg5.log 2>&1 || { tail -5 $OUT/r3f_cfg5.log; exit 1; }
cp $OUT/r3f_cfg5_stats/*/*_kernel_stats.csv $OUT/r03_cfg5_B512_kernel_stats.csv
grep -v amdgpu.ids $OUT/r3f_cfg5.log | tail -3
head -25 $OUT/r03_cfg5_B512_kernel_stats.csv | cut -c1-200
