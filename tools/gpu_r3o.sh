#!/bin/bash
# round 3: new width-1024 test + the bench's new blocks
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -p no:cacheprovider -k "width_1024" > $OUT/r3o_tests.log 2>&1
rc=$?
grep -v amdgpu.ids $OUT/r3o_tests.log | tail -6
if [ $rc -ne 0 ]; then echo "tests rc=$rc: stopping"; exit $rc; fi
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > $OUT/r3o_bench.json 2> $OUT/r3o_bench.err || { tail -5 $OUT/r3o_bench.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3o_bench.json').read().strip().splitlines()[-1])
print("value %.3fM"%(d['value']/1e6))
print("cfg5 rating", d['cfg5'].get('rating_window'), d['cfg5'].get('error'))
print("ref models", json.dumps(d['reference_models']))
PY
