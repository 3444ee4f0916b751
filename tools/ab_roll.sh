#!/bin/bash
export KL_PROBE_TRAIN_ONLY=1 KL_PROBE_N=40 KL_PROBE_WARM=10
run() { echo -n "$1 : "; shift; env "$@" timeout -k 10 120 python tools/probe_perf.py $b 2>&1 | grep train | cut -c1-60 || exit 1; }
for rep in 1 2; do
for b in $1; do
  echo "== B=$b rep=$rep"
  run "default (rolling sentinels)" KL_NOP=1
  run "KL_SENTINEL_ROLL=0         " KL_SENTINEL_ROLL=0
done
done
