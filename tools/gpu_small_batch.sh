#!/bin/bash
# round 3: thin backward scan with two partial accumulators -- gradient tests, the batched-streams training test, small shapes
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_rater_plumbing.py -q -m gpu -x -p no:cacheprovider -k "train_window_gradients or trajectory or consecutive or batched_streams or launch_per_step" > $OUT/r3m_tests.log 2>&1
rc=$?
grep -v amdgpu.ids $OUT/r3m_tests.log | tail -8
if [ $rc -ne 0 ]; then echo "tests rc=$rc: stopping"; exit $rc; fi
timeout -k 10 300 python tools/probe_shapes.py small 2>&1 | grep -E "^cfg" | tee $OUT/r3m_shapes.log
