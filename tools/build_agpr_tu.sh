#!/bin/bash
# Builds lstm_scan2.o (and its ISA listing for the audit).
#
# The 8-wave backward scan keeps its asynchronous loads in flight around its loop in ACCUMULATOR registers (lstm_scan2.hip,
# lstm_scan_bwd_regtile_kernel).  That needs two things HIP source cannot say:
#   * the function attribute "amdgpu-agpr-alloc"="40,40" on those kernels -- without it the compiler halves the register
#     budget between VGPRs and AGPRs as soon as an AGPR is named (128 + 128: the weights alone take 128 VGPRs);
#   * -amdgpu-mfma-vgpr-form, so that the MFMA accumulator stays in VGPRs and the compiler has no reason to touch an AGPR.
# So the translation unit is built in the steps hipcc runs internally, with the attribute set on the device IR in between:
#   device IR (optimised, -emit-llvm)  ->  attribute on the lstm_scan_bwd_regtile_kernel definitions  ->  code object
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
N=$(grep -cE '^define .*lstm_scan_bwd_regtile_kernel' "$TMP/dev.ll" || true)
# (40 accumulator registers: six epilogue inputs + the 32 of the register-landing tile)
sed -E -i '/^define .*lstm_scan_bwd_regtile_kernel/ s/\) local_unnamed_addr (#[0-9]+)/) local_unnamed_addr \1 "amdgpu-agpr-alloc"="40,40"/' "$TMP/dev.ll"
M=$(grep -cE '"amdgpu-agpr-alloc"="40,40"' "$TMP/dev.ll" || true)
if [ "$N" -lt 1 ] || [ "$N" != "$M" ] || [ -n "$KL_NO_REGTILE" ]; then
  # Another compiler version prints the definition differently (or KL_NO_REGTILE=1 asks for it): build the translation unit with
  # plain hipcc WITHOUT the register-tile kernel (-DKL_NO_REGTILE: kl_scan_bwd_regtile_min_np() then keeps every shape on the
  # 16-wave backward scan), and say so in the listing so that `make audit` knows what it is looking at.
  echo "build_agpr_tu: attribute set on $M of $N register-tile kernels -- building $SRC without them (-DKL_NO_REGTILE)" >&2
  "$HIPCC" "$@" -DKL_NO_REGTILE -c "$SRC" -o "$OUT"
  "$HIPCC" "$@" -DKL_NO_REGTILE --cuda-device-only -S "$SRC" -o "$TMP/plain.s" 2> /dev/null
  { echo "; KL_NO_REGTILE"; cat "$TMP/plain.s"; } > "$ASM"
  exit 0
fi
CG="-target amdgcn-amd-amdhsa -mcpu=$ARCH -O3 -Xclang -disable-llvm-optzns -mllvm -amdgpu-mfma-vgpr-form=1"
"$LLVM/clang" $CG -S "$TMP/dev.ll" -o "$ASM"
"$LLVM/clang" $CG "$TMP/dev.ll" -o "$TMP/dev.co"
"$LLVM/clang-offload-bundler" -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--$ARCH \
    -input=/dev/null -input="$TMP/dev.co" -output="$TMP/dev.hipfb"
"$HIPCC" "$@" --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang "$TMP/dev.hipfb" -c "$SRC" -o "$OUT" 2> "$TMP/err2" || { cat "$TMP/err2" >&2; exit 1; }
grep -v "argument unused during compilation" "$TMP/err2" >&2 || true
