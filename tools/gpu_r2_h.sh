#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/r2h_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r2h_stats -- python3 bench.py --streams 3072 --steps 10 --warmup 3 --no-cpu-baseline --no-incremental --no-end-to-end > gpurun_out/r2h_bench.json 2> gpurun_out/r2h_rp.err
cp gpurun_out/r2h_stats/*/*_kernel_stats.csv gpurun_out/r2h_kernel_stats.csv
cp gpurun_out/r2h_stats/*/*_kernel_trace.csv gpurun_out/r2h_kernel_trace.csv
rm -rf gpurun_out/r2h_stats
tail -c 600 gpurun_out/r2h_bench.json
