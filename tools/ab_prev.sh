#!/bin/bash
# working tree vs the library of the last commit (ocrd_keraslm_amd/libkeraslm_hip_prev.so) on one box
export KL_PROBE_TRAIN_ONLY=1 KL_PROBE_N=40 KL_PROBE_WARM=10
for rep in 1 2 3; do
for b in $1; do
  echo -n "B=$b new : "; timeout -k 10 120 python tools/probe_perf.py $b 2>&1 | grep train | cut -c1-60 || exit 1
  echo -n "B=$b prev: "; KL_LIB=$PWD/ocrd_keraslm_amd/libkeraslm_hip_prev.so timeout -k 10 120 python tools/probe_perf.py $b 2>&1 | grep train | cut -c1-60 || exit 1
done
done
