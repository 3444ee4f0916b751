#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python bench.py --steps 30 --warmup 5 > gpurun_out/r2c_bench.json 2> gpurun_out/r2c_bench.err
echo "rc=$?"; tail -c 3000 gpurun_out/r2c_bench.json; tail -5 gpurun_out/r2c_bench.err
