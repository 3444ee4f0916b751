#!/bin/bash
# A/B of the scan hand-off switches on one box: tools/ab_flags.sh "512 1024" > gpurun_out/ab_flags.txt
export KL_PROBE_TRAIN_ONLY=1 KL_PROBE_N=40 KL_PROBE_WARM=10
for rep in 1 2 3; do
for b in $1; do
  echo "== B=$b rep=$rep"
  if [ -f ocrd_keraslm_amd/libkeraslm_hip_prev.so ]; then
    echo -n "prev lib            : "; KL_LIB=$PWD/ocrd_keraslm_amd/libkeraslm_hip_prev.so timeout -k 10 120 python tools/probe_perf.py $b 2>&1 | grep train || exit 1
  fi
  echo -n "all on              : "; timeout -k 10 120 python tools/probe_perf.py $b 2>&1 | grep train || exit 1
  echo -n "LOCAL_BWD=1         : "; KL_XCD_LOCAL_BWD=1 timeout -k 10 120 python tools/probe_perf.py $b 2>&1 | grep train || exit 1
  echo -n "SENT_BWD=0          : "; KL_SENTINEL_BWD=0 timeout -k 10 120 python tools/probe_perf.py $b 2>&1 | grep train || exit 1
  echo -n "XCD_LOCAL=0         : "; KL_XCD_LOCAL=0 timeout -k 10 120 python tools/probe_perf.py $b 2>&1 | grep train || exit 1
  echo -n "SENT_BWD=0 LOCAL=0  : "; KL_SENTINEL_BWD=0 KL_XCD_LOCAL=0 timeout -k 10 120 python tools/probe_perf.py $b 2>&1 | grep train || exit 1
done
done
