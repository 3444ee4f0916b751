#!/bin/bash
# the whole GPU tier as the driver runs it: all -m gpu tests, smoke, then the bench line
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 1100 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/full_tests.log 2>&1
rc=$?
grep -v amdgpu.ids $OUT/full_tests.log | tail -12
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests timed out: stopping"; exit $rc; fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
timeout -k 10 700 python bench.py > $OUT/full_bench.json 2> $OUT/full_bench.err
rc2=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/full_bench.json').read().strip().splitlines()[-1])
print("value %.3fM ms/step %.2f"%(d['value']/1e6,d['ms_per_step']))
print("roofline",{k:d['roofline'][k] for k in ('kernel','achieved','frac','launch_us','whole_step_frac')}, d['roofline']['other_kernel'])
print("inc", d['incremental']['gpu_us_per_step'], d['incremental']['n128']['gpu_us_per_step'])
print("cfg5", d['cfg5'].get('training'), d['cfg5'].get('error'))
print("small", {k:v.get('value') if isinstance(v,dict) else v for k,v in d['small_batch'].items()})
print("e2e", d['end_to_end'])
PY
exit $(( rc != 0 ? rc : rc2 ))
