#!/bin/bash
# Builds lstm_scan2.o (and its ISA listing for the audit).
#
# The backward scan keeps six asynchronous loads in flight around its loop in ACCUMULATOR registers (lstm_scan2.hip,
# KL_BWD_INPUTS_*).  That needs two things HIP source cannot say:
#   * the function attribute "amdgpu-agpr-alloc"="8,8" on those kernels -- without it the compiler halves the 128-register
#     budget of a 1024-thread workgroup between VGPRs and AGPRs as soon as an AGPR is named (64 + 64: the weights alone take 64);
#   * -amdgpu-mfma-vgpr-form, so that the MFMA accumulator stays in VGPRs and the compiler has no reason to touch an AGPR.
# So the translation unit is built in the steps hipcc runs internally, with the attribute set on the device IR in between:
#   device IR (optimised, -emit-llvm)  ->  attribute on the lstm_scan_bwd_wide2 / wide3 kernel definitions  ->  code object
#   (-disable-llvm-optzns: the IR is already optimised; the forward kernels come out instruction for instruction as from
#   plain hipcc)  ->  offload bundle  ->  host object with the bundle embedded (-fcuda-include-gpubinary).
# Usage: build_agpr_tu.sh <src.hip> <out.o> <out.s> -- <hipcc flags...>
set -eo pipefail
SRC=$1; OUT=$2; ASM=$3; shift 4
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
LLVM=${LLVM_BIN:-/opt/rocm/lib/llvm/bin}
ARCH=${ARCH:-gfx950}
TMP=$(mktemp -d /tmp/kl_agpr_XXXXXX)
trap 'rm -rf "$TMP"' EXIT
"$HIPCC" "$@" --cuda-device-only -emit-llvm -S "$SRC" -o "$TMP/dev.ll" 2> "$TMP/err" || { cat "$TMP/err" >&2; exit 1; }
grep -v "argument unused during compilation" "$TMP/err" >&2 || true
N=$(grep -cE '^define .*lstm_scan_bwd_(wide[23]|regtile)_kernel' "$TMP/dev.ll" || true)
# (8 accumulator registers: the epilogue inputs; 40: + the register-landing tile of lstm_scan_bwd_regtile_kernel)
sed -E -i '/^define .*lstm_scan_bwd_wide[23]_kernel/ s/\) local_unnamed_addr (#[0-9]+)/) local_unnamed_addr \1 "amdgpu-agpr-alloc"="8,8"/' "$TMP/dev.ll"
sed -E -i '/^define .*lstm_scan_bwd_regtile_kernel/ s/\) local_unnamed_addr (#[0-9]+)/) local_unnamed_addr \1 "amdgpu-agpr-alloc"="40,40"/' "$TMP/dev.ll"
M=$(grep -cE '"amdgpu-agpr-alloc"="(8,8|40,40)"' "$TMP/dev.ll" || true)
if [ "$N" -lt 1 ] || [ "$N" != "$M" ]; then echo "build_agpr_tu: attribute set on $M of $N backward-scan kernels" >&2; exit 1; fi
CG="-target amdgcn-amd-amdhsa -mcpu=$ARCH -O3 -Xclang -disable-llvm-optzns -mllvm -amdgpu-mfma-vgpr-form=1"
"$LLVM/clang" $CG -S "$TMP/dev.ll" -o "$ASM"
"$LLVM/clang" $CG "$TMP/dev.ll" -o "$TMP/dev.co"
"$LLVM/clang-offload-bundler" -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--$ARCH \
    -input=/dev/null -input="$TMP/dev.co" -output="$TMP/dev.hipfb"
"$HIPCC" "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$TMP/dev.hipfb" -c "$SRC" -o "$OUT" 2> "$TMP/err2" || { cat "$TMP/err2" >&2; exit 1; }
grep -v "argument unused during compilation" "$TMP/err2" >&2 || true
