#!/bin/bash
# timing experiments on the second-generation scans: variant builds of lstm_scan2.o (KL_BWD_VAR, or KL_FWD_VAR with
# VARMACRO=KL_FWD_VAR) linked with the shipped objects
# usage (on the build host): [VARMACRO=KL_FWD_VAR] bash tools/gpu_variants.sh build "1 2 4"   -> ocrd_keraslm_amd/libkeraslm_hip_var<N>.so
cd "$(dirname "$0")/../ocrd_keraslm_amd/csrc"
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../../include -I. -Wall -Wno-unused-function"
for v in $2; do
  bash ../../tools/build_agpr_tu.sh lstm_scan2.hip /tmp/scan2_var$v.o /tmp/scan2_var$v.s -- $F -D${VARMACRO:-KL_BWD_VAR}=$v || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libkeraslm_hip_var$v.so api.o gemm.o lstm_step.o lstm_scan.o /tmp/scan2_var$v.o lstm_scan_w32.o step_big.o step_small.o elementwise.o tables.o || exit 1
done
