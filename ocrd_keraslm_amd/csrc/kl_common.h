// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels of the Rater hot path.
// wave = 64 lanes; MFMA = v_mfma_f32_16x16x32_bf16 (f32 accumulate).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned short bf16_t;   // storage type of bf16 tensors in HBM

#define KL_WAVE 64

__device__ __forceinline__ bf16_t f2bf(float x) {
  __bf16 h = (__bf16)x;   // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float bf2f(bf16_t x) {
  return __builtin_bit_cast(float, ((uint32_t)x) << 16);
}

// 16 bytes = 8 bf16 as one vector register group
union frag16 {
  uint4 u;
  bf16x8 v;
  bf16_t s[8];
};

__device__ __forceinline__ f32x4 mfma16(const bf16x8 a, const bf16x8 b, f32x4 c) {
  // D[16x16] += A[16x32] . B[32x16];  lane l holds A[row l&15][k 8(l>>4)..+7],
  // B[k 8(l>>4)..+7][col l&15]; D: col = l&15, row = 4(l>>4)+reg.
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// full-precision ocml exp/tanh: the gate math is never the bottleneck and the
// inference path is checked against the oracle to 1e-5 in split-bf16 mode
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }

// split an f32 value into bf16 hi + bf16 lo (hi+lo carries ~16 mantissa bits)
__device__ __forceinline__ void split_bf16(float x, bf16_t& hi, bf16_t& lo) {
  hi = f2bf(x);
  lo = f2bf(x - bf2f(hi));
}
