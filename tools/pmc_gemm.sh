#!/bin/bash
# PMC counters of the ring GEMMs (separate passes, --kernel-trace only): bash tools/pmc_gemm.sh
export TMPDIR=/tmp
OUT=gpurun_out/pmc_gemm
mkdir -p $OUT
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/an_$tag -- python3 tools/probe_gemm_an.py 1024 > $OUT/an_$tag.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/tn_$tag -- python3 tools/probe_gemm.py 1024 "f32 out" > $OUT/tn_$tag.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/pmc_gemm/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'gemm' not in n:
            continue
        key = (n.split('(')[0][-60:], r.get('Grid_Size', ''))
        agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, cs in sorted(agg.items()):
    print(key)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
