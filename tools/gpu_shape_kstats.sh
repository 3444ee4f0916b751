#!/bin/bash
# round 3: kernel times of the cfg5 training step (rocprofv3 kernel stats); env of the caller selects the variant
# usage: bash tools/gpu_shape_kstats.sh <tag> [small|cfg5]
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
TAG=${1:-x}
WHAT=${2:-cfg5}
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r3j_${WHAT}_$TAG -- python3 tools/probe_shapes.py $WHAT > $OUT/r3j_${WHAT}_$TAG.log 2>&1 || exit 1
cp $OUT/r3j_${WHAT}_$TAG/*/*_kernel_stats.csv $OUT/r3j_${WHAT}_${TAG}_kernel_stats.csv
grep "^cfg" $OUT/r3j_${WHAT}_$TAG.log
head -12 $OUT/r3j_${WHAT}_${TAG}_kernel_stats.csv | cut -c1-170
