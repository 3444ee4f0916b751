#!/bin/bash
# round 3: Rater.train with the batched streams -- GPU tests that train, then the end-to-end leg
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_rater_plumbing.py tests/test_cli.py tests/test_ddp_hip.py tests/test_gpu_kernels.py tests/test_wrapper_processor_gpu.py -q -m gpu -x -p no:cacheprovider -k "plumbing or cli or ddp or step_batch or processor or rater" > $OUT/r3g_tests.log 2>&1
rc=$?
grep -v amdgpu.ids $OUT/r3g_tests.log | tail -8
if [ $rc -ne 0 ]; then echo "tests rc=$rc: stopping"; exit $rc; fi
timeout -k 10 600 python -c "
import bench, json, time
for B in (3072, 1024):
    print(json.dumps(bench.end_to_end_leg(B)))
" 2>&1 | grep -v amdgpu.ids | tee $OUT/r3g_e2e.log
