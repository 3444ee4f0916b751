#!/bin/bash
# per-kernel time of the training step (rocprofv3 kernel trace of tools/probe_perf.py): bash tools/gpu_kstats.sh <streams>
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/ks_stats
KL_PROBE_TRAIN_ONLY=1 KL_PROBE_N=6 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_stats -- python3 tools/probe_perf.py ${1:-3072} > gpurun_out/ks.log 2>&1
cp gpurun_out/ks_stats/*/*_kernel_stats.csv gpurun_out/ks_kernel_stats.csv
rm -rf gpurun_out/ks_stats
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/ks_kernel_stats.csv')))
for r in rows[:14]:
    print(r['Name'][:90].ljust(90), r['Calls'], int(float(r['AverageNs'])), r['Percentage'])
PY
