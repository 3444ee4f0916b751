#!/bin/bash
# Reproduce the round's measurement artefacts on a GPU box (run from the repo root):
#   bench line, rocprofv3 kernel stats of the same command, and the two PMC passes for HBM traffic.
# Usage: bash tools/profile_round.sh <tag> [streams]      -> files under gpurun_out/<tag>_*
set -eo pipefail
TAG=${1:-r02}
B=${2:-3072}
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 700 python bench.py --streams $B > $OUT/${TAG}_bench_B${B}.json 2> $OUT/${TAG}_bench.err
tail -c 1500 $OUT/${TAG}_bench_B${B}.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 bench.py --streams $B --steps 10 --warmup 3 --no-cpu-baseline --no-incremental --no-end-to-end --no-extra-shapes > $OUT/${TAG}_bench_B${B}_under_rocprof.json 2> $OUT/${TAG}_rp.err
cp $OUT/${TAG}_stats/*/*_kernel_stats.csv $OUT/${TAG}_bench_B${B}_kernel_stats.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 bench.py --streams $B --steps 3 --warmup 1 --no-cpu-baseline --no-incremental --no-end-to-end --no-extra-shapes > $OUT/${TAG}_pmc_f.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -- python3 bench.py --streams $B --steps 3 --warmup 1 --no-cpu-baseline --no-incremental --no-end-to-end --no-extra-shapes > $OUT/${TAG}_pmc_w.log 2>&1
python tools/pmc_summary.py $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write $OUT/${TAG}_pmc_hbm_traffic_B${B}.json "cfg2 B=$B T=256"
head -8 $OUT/${TAG}_bench_B${B}_kernel_stats.csv | cut -c1-160
# the incremental step (cfg3 and the reference's 128-hypothesis cap): per-kernel times
KL_PROBE_PREC=3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_inc_stats -- python3 tools/probe_incremental.py 1024 > $OUT/${TAG}_inc1024.log 2>&1
cp $OUT/${TAG}_inc_stats/*/*_kernel_stats.csv $OUT/${TAG}_incremental_n1024_kernel_stats.csv
KL_PROBE_PREC=3 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_inc128_stats -- python3 tools/probe_incremental.py 128 > $OUT/${TAG}_inc128.log 2>&1
cp $OUT/${TAG}_inc128_stats/*/*_kernel_stats.csv $OUT/${TAG}_incremental_n128_kernel_stats.csv
# the rating window (Rater.rate): 1 stream x 256 chars, split precision
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_rate_stats -- python3 tools/probe_rate_window.py 1 > $OUT/${TAG}_rate1.log 2>&1
cp $OUT/${TAG}_rate_stats/*/*_kernel_stats.csv $OUT/${TAG}_rating_window_B1_kernel_stats.csv
grep "us/step" $OUT/${TAG}_inc1024.log $OUT/${TAG}_inc128.log
grep "rating window" $OUT/${TAG}_rate1.log
# cfg5 (depth 4, width 1024, length 512) at 512 streams and the reference's own model size (depth 2, width 128, length 256): per-kernel times
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_cfg5_stats -- python3 tools/probe_cfg5.py 512 > $OUT/${TAG}_cfg5.log 2>&1
cp $OUT/${TAG}_cfg5_stats/*/*_kernel_stats.csv $OUT/${TAG}_cfg5_B512_kernel_stats.csv
for WB in 1024 4096; do
  KL_SHAPE="2,128,256,$WB,20" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_w128_${WB}_stats -- python3 tools/probe_shapes.py shape > $OUT/${TAG}_w128_${WB}.log 2>&1
  cp $OUT/${TAG}_w128_${WB}_stats/*/*_kernel_stats.csv $OUT/${TAG}_w128_B${WB}_kernel_stats.csv
done
grep "cfg5 train" $OUT/${TAG}_cfg5.log
grep -h "^shape" $OUT/${TAG}_w128_1024.log $OUT/${TAG}_w128_4096.log
