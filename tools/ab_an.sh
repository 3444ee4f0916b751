#!/bin/bash
# A/B of the K-major weight-gradient GEMMs on one box: tools/ab_an.sh "512 1024"
export KL_PROBE_TRAIN_ONLY=1 KL_PROBE_N=40 KL_PROBE_WARM=10
for rep in 1 2; do
for b in $1; do
  echo "== B=$b rep=$rep"
  echo -n "default (dz K-major): "; timeout -k 10 120 python tools/probe_perf.py $b 2>&1 | grep train || exit 1
  echo -n "KL_GEMM_AN=0        : "; KL_GEMM_AN=0 timeout -k 10 120 python tools/probe_perf.py $b 2>&1 | grep train || exit 1
done
done
