#!/bin/bash
# A/B of the scan hand-off switches on one box: bash tools/ab_flags.sh "512 1024" > gpurun_out/ab_flags.txt
export KL_PROBE_TRAIN_ONLY=1 KL_PROBE_N=40 KL_PROBE_WARM=10
run() { echo -n "$1 : "; shift; env "$@" timeout -k 10 120 python tools/probe_perf.py $b 2>&1 | grep train | cut -c1-60 || exit 1; }
for rep in 1 2; do
for b in $1; do
  echo "== B=$b rep=$rep"
  run "default                 " KL_NOP=1
  run "SENT_BWD=2 (all shapes) " KL_SENTINEL_BWD=2
  run "SENT_BWD=0 (counters)   " KL_SENTINEL_BWD=0
  run "XCD_LOCAL=1             " KL_XCD_LOCAL=1
  run "XCD_LOCAL=1 LOCAL_BWD=1 SENT_BWD=2" KL_XCD_LOCAL=1 KL_XCD_LOCAL_BWD=1 KL_SENTINEL_BWD=2
  run "SENTINEL=0 (all counters)" KL_SENTINEL=0
done
done
