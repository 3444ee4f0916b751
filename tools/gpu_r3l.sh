#!/bin/bash
# round 3: per-launch times of the width-1024 scans in the cfg5 training step (rocprofv3 kernel stats); KL_W32_VAR is
# passed through to the handle for timing experiments
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
for v in ${KL_VARS:-0 2 4 8 12}; do
  KL_W32_VAR=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r3l_v$v -- python3 tools/probe_shapes.py cfg5 > $OUT/r3l_v$v.log 2>&1 || exit 1
  echo "== KL_W32_VAR=$v: $(grep '^cfg5' $OUT/r3l_v$v.log | head -1)"
  grep -h "w32_kernel" $OUT/r3l_v$v/*/*_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
