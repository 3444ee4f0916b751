#!/bin/bash
mkdir -p gpurun_out
rm -f gpurun_out/r2i_*.log
S="786432,2048,512,1"
timeout -k 10 200 python tools/probe_gemm.py $S 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2i_2.log
KL_LIB=ocrd_keraslm_amd/libkeraslm_hip_nostore.so timeout -k 10 200 python tools/probe_gemm.py $S 2>&1 | grep -v amdgpu.ids >> gpurun_out/r2i_2.log
cat gpurun_out/r2i_2.log
