// Incremental step for FEW hypotheses (n < 256: the regime the reference's callers use -- rate_best feeds at most
// 128 rows per call, generate at most 256; rating.py:49, 704, 809) and the fused output layer of every incremental step.
//
// What the launch-per-layer thin kernels of lstm_step.hip cost at n = 128 (round 2: 2 x 17.5 us + 5.9 + 4.6 us): 512
// four-unit workgroups, each gathering the f32 state rows of 32 hypotheses 16 bytes at a time (fragment-shaped loads: 64
// different 64-byte segments per wave instruction) -- 192 KB per workgroup for 48 MFMAs per wave -- and a separate thin GEMM
// and softmax for the tied output projection.  Here:
//  * inc_cell_kernel: one workgroup = 16 hidden units x 4 gates (four MFMA column tiles) x 16 or 32 hypotheses, all of K.
//    The state rows come in COALESCED (whole 2 KiB rows, 1 KiB per wave instruction), are split into bf16 hi + lo once and
//    laid into LDS (16-byte chunks XOR-swizzled by row, so that the 16 rows of an MFMA fragment fall on different banks);
//    every wave then contracts a quarter of K against weight fragments it loads straight into registers (each weight
//    element is used by exactly one wave: no staging), the four partial tiles meet in LDS and the cell update runs on the
//    reduced tile.  A row's bytes are fetched by W/16 workgroups instead of W/4.
//  * logits_softmax_kernel: logits = h . E^T (split precision), softmax and the store of the probabilities in one launch:
//    a workgroup owns 16 hypotheses and ALL characters, so the row maximum and sum never leave the CU.
// Restates rating.py:578-639 (Rater.predict: one LSTM step per layer with explicit states, softmax over the tied
// embedding) for the arithmetic; rows = hypotheses, state rows addressed through pool slots.
#include <string.h>

#include "kl_common.h"
#include "kl_kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// f32 x 4 -> bf16 hi x 4 (8 bytes) and, LO, the bf16 of the residuals
template <bool LO>
__device__ __forceinline__ void split4(const float4 v, uint2& hi, uint2& lo) {
  const bf16_t h0 = f2bf(v.x), h1 = f2bf(v.y), h2 = f2bf(v.z), h3 = f2bf(v.w);
  hi.x = (unsigned)h0 | ((unsigned)h1 << 16);
  hi.y = (unsigned)h2 | ((unsigned)h3 << 16);
  if (LO) {
    lo.x = (unsigned)f2bf(v.x - bf2f(h0)) | ((unsigned)f2bf(v.y - bf2f(h1)) << 16);
    lo.y = (unsigned)f2bf(v.z - bf2f(h2)) | ((unsigned)f2bf(v.w - bf2f(h3)) << 16);
  }
}

// LDS image of the activation rows: plane p (0 = hi, 1 = lo), row r, K elements; 16-byte chunk c of row r sits at chunk
// position c ^ (r & 15) of its 128-byte-aligned group of 16 chunks... (K is a multiple of 128 elements = 16 chunks)
__device__ __forceinline__ unsigned a_off(int row, int k, int K) {      // byte offset inside a plane of element k of row `row` (k % 4 == 0)
  const int chunk = k >> 3;
  const int pos = (chunk & ~15) | ((chunk ^ row) & 15);
  return (unsigned)((row * K + pos * 8 + (k & 7)) * 2);
}

struct IncCell {
  int n, W;
  float* pool; long slot_ld;
  const int* slot_in; const int* slot_out;
  int h_off, c_off, x_off;          // float offsets inside a slot: this layer's h and c, the layer below's h (-1: layer 0)
  const bf16_t* UT_hi; const bf16_t* UT_lo; const bf16_t* KT_hi; const bf16_t* KT_lo;      // [4W][W]
  const float* T1; const int* i1; const float* T2; const int* i2; const float* bias;       // z init (tables [.][4W], bias [4W])
};

// grid (W / 16, ceil(n / (16 NMT))), 256 threads = 4 waves (K split)
template <int NMT, bool LO>
__global__ __launch_bounds__(256) void inc_cell_kernel(const IncCell a) {
  constexpr int ROWS = 16 * NMT;
  const int W = a.W;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u0 = blockIdx.x * 16, r0 = blockIdx.y * ROWS;
  const int K = a.x_off >= 0 ? 2 * W : W;          // [x | h] or h alone
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const plane_hi = smem;
  unsigned char* const plane_lo = smem + (size_t)ROWS * K * 2;
  __shared__ int s_in[ROWS], s_out[ROWS];
  if (tid < ROWS) {
    const int r = min(r0 + tid, a.n - 1);
    s_in[tid] = a.slot_in[r];
    s_out[tid] = a.slot_out[r];
  }
  __syncthreads();

  // ---- weight fragments of this wave's first k-steps go out first (they do not depend on anything)
  const int nks = K >> 5, nks_x = a.x_off >= 0 ? (W >> 5) : 0;      // k-steps of 32; the first nks_x belong to x . K
  const int col = lane & 15, kg = lane >> 4;
  auto wfrag = [&](int ks, int g, bool lo) -> u32x4_t {
    const bool is_x = ks < nks_x;
    const bf16_t* base = is_x ? (lo ? a.KT_lo : a.KT_hi) : (lo ? a.UT_lo : a.UT_hi);
    const int kk = (is_x ? ks : ks - nks_x) * 32 + kg * 8;
    return *reinterpret_cast<const u32x4_t*>(base + (long)(g * W + u0 + col) * W + kk);
  };
  // One wave per SIMD: nothing hides a load but the loads issued beside it.  So ALL weight fragments of up to GS k-steps of this
  // wave go out at once (GS x 8 fragments = 256 registers in split precision -- a 256-thread workgroup may take 512), in front of
  // the activation staging; width 512 needs one such group per layer, width 1024 two for the layers above the first.
  constexpr int GS = 8;
  u32x4_t bh[GS][4], bl[GS][4];
  const int my_steps = (nks - wave + 3) >> 2;      // k-steps wave, wave + 4, ...
  auto load_group = [&](int s0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < GS; ++j) {
      if (s0 + j < my_steps) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bh[j][g] = wfrag(wave + 4 * (s0 + j), g, false);
          if (LO) bl[j][g] = wfrag(wave + 4 * (s0 + j), g, true);
        }
      }
    }
  };
  load_group(0);

  // ---- activation rows: coalesced, split, into LDS
  {
    const int per_row = K >> 2;                    // float4 pieces per row
    for (int e = tid; e < ROWS * per_row; e += 256) {
      const int r = e / per_row, q = e - r * per_row;
      const int k = q * 4;
      const float* src = (a.x_off >= 0 && k < W) ? a.pool + (long)s_out[r] * a.slot_ld + a.x_off + k
                                                 : a.pool + (long)s_in[r] * a.slot_ld + a.h_off + (k - (a.x_off >= 0 ? W : 0));
      const float4 v = *reinterpret_cast<const float4*>(src);
      uint2 hi, lo;
      split4<LO>(v, hi, lo);
      const unsigned off = a_off(r, k, K);
      *reinterpret_cast<uint2*>(plane_hi + off) = hi;
      if (LO) *reinterpret_cast<uint2*>(plane_lo + off) = lo;
    }
  }
  __syncthreads();

  // ---- contraction: wave w takes k-steps w, w + 4, ...; 3 MFMAs per (row tile, gate) and k-step in split precision
  f32x4 acc[NMT][4];
#pragma unroll
  for (int m = 0; m < NMT; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[m][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < my_steps; s0 += GS) {
    if (s0 > 0) load_group(s0);
#pragma unroll
    for (int j = 0; j < GS; ++j) {
      if (s0 + j < my_steps) {
        const int ks = wave + 4 * (s0 + j);
#pragma unroll
        for (int m = 0; m < NMT; ++m) {
          const unsigned off = a_off(m * 16 + col, ks * 32 + kg * 8, K);
          const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_t*>(plane_hi + off));
          bf16x8 al = ah;
          if (LO) al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_t*>(plane_lo + off));
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const bf16x8 wh = __builtin_bit_cast(bf16x8, bh[j][g]);
            acc[m][g] = mfma16(ah, wh, acc[m][g]);
            if (LO) {
              acc[m][g] = mfma16(al, wh, acc[m][g]);
              acc[m][g] = mfma16(ah, __builtin_bit_cast(bf16x8, bl[j][g]), acc[m][g]);
            }
          }
        }
      }
    }
  }
  __syncthreads();      // (every wave has read its last fragments: the planes make room for the partial tiles)
  float* const part = reinterpret_cast<float*>(smem);      // [4 waves][NMT][4 gates][16 rows][17]
#pragma unroll
  for (int m = 0; m < NMT; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(((wave * NMT + m) * 4 + g) * 16 + kg * 4 + r) * 17 + col] = acc[m][g][r];
  __syncthreads();

  // ---- cell update on the reduced tile: thread = (row, unit), NMT cells each
#pragma unroll
  for (int m = 0; m < NMT; ++m) {
    const int lr = m * 16 + (tid >> 4), ej = tid & 15;
    const int row = r0 + lr;
    if (row >= a.n) continue;
    const int u = u0 + ej;
    float z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v = a.bias ? a.bias[g * W + u] : 0.f;
      if (a.T1) v += a.T1[(long)(a.i1 ? a.i1[row] : row) * 4 * W + g * W + u];
      if (a.T2) v += a.T2[(long)(a.i2 ? a.i2[row] : row) * 4 * W + g * W + u];
#pragma unroll
      for (int w = 0; w < 4; ++w) v += part[(((w * NMT + m) * 4 + g) * 16 + (tid >> 4)) * 17 + ej];
      z[g] = v;
    }
    const float cp = a.pool[(long)s_in[lr] * a.slot_ld + a.c_off + u];
    const float gi = sigmoidf_(z[0]), gf = sigmoidf_(z[1]), gg = tanhf_(z[2]), go = sigmoidf_(z[3]);
    const float c = gf * cp + gi * gg;
    const float h = go * tanhf_(c);
    float* out = a.pool + (long)s_out[lr] * a.slot_ld;
    out[a.c_off + u] = c;
    out[a.h_off + u] = h;
  }
}

struct LogitsSoftmax {
  int n, W, V;
  const float* pool; long slot_ld; const int* slot_out; int h_off;
  const bf16_t* E_hi; const bf16_t* E_lo;      // [Vp][W], rows >= V zero
  float* probs;                                 // [n][V]
};

// grid ceil(n / 16), 512 threads = 8 waves; a wave owns the character tiles w, w + 8, ... (16 characters each)
template <bool LO>
__global__ __launch_bounds__(512) void logits_softmax_kernel(const LogitsSoftmax a) {
  const int W = a.W, V = a.V;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = blockIdx.x * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const plane_hi = smem;
  unsigned char* const plane_lo = smem + (size_t)16 * W * 2;
  const int Vt = (V + 15) & ~15;
  float* const lg = reinterpret_cast<float*>(smem + (size_t)16 * W * 4);      // [16][Vt + 1]
  __shared__ int s_out[16];
  if (tid < 16) s_out[tid] = a.slot_out[min(r0 + tid, a.n - 1)];
  __syncthreads();
  {
    const int per_row = W >> 2;
    for (int e = tid; e < 16 * per_row; e += 512) {
      const int r = e / per_row, k = (e - r * per_row) * 4;
      const float4 v = *reinterpret_cast<const float4*>(a.pool + (long)s_out[r] * a.slot_ld + a.h_off + k);
      uint2 hi, lo;
      split4<LO>(v, hi, lo);
      const unsigned off = a_off(r, k, W);
      *reinterpret_cast<uint2*>(plane_hi + off) = hi;
      if (LO) *reinterpret_cast<uint2*>(plane_lo + off) = lo;
    }
  }
  __syncthreads();
  const int col = lane & 15, kg = lane >> 4, nks = W >> 5;
  for (int tile = wave; tile * 16 < V; tile += 8) {
    const long erow = (long)(tile * 16 + col) * W + kg * 8;      // (E is padded to a multiple of 32 rows: no clamp)
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    // (all fragments of up to 16 k-steps at once: 128 registers in split precision, of the 256 a 512-thread workgroup may take)
    constexpr int GS = 16;
    u32x4_t eh[GS], el[GS];
    for (int k0 = 0; k0 < nks; k0 += GS) {
#pragma unroll
      for (int j = 0; j < GS; ++j) {
        if (k0 + j < nks) {
          eh[j] = *reinterpret_cast<const u32x4_t*>(a.E_hi + erow + (k0 + j) * 32);
          if (LO) el[j] = *reinterpret_cast<const u32x4_t*>(a.E_lo + erow + (k0 + j) * 32);
        }
      }
#pragma unroll
      for (int j = 0; j < GS; ++j) {
        if (k0 + j < nks) {
          const unsigned off = a_off(col, (k0 + j) * 32 + kg * 8, W);
          const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_t*>(plane_hi + off));
          const bf16x8 wh = __builtin_bit_cast(bf16x8, eh[j]);
          acc = mfma16(ah, wh, acc);
          if (LO) {
            const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4_t*>(plane_lo + off));
            acc = mfma16(al, wh, acc);
            acc = mfma16(ah, __builtin_bit_cast(bf16x8, el[j]), acc);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) lg[(kg * 4 + r) * (Vt + 1) + tile * 16 + col] = acc[r];
  }
  __syncthreads();
  // softmax: wave w takes rows w and w + 8
  for (int lr = wave; lr < 16; lr += 8) {
    const int row = r0 + lr;
    if (row >= a.n) continue;
    const float* x = lg + lr * (Vt + 1);
    float mx = -INFINITY;
    for (int v = lane; v < V; v += 64) mx = fmaxf(mx, x[v]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.f;
    for (int v = lane; v < V; v += 64) sum += expf(x[v] - mx);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float inv = 1.f / sum;
    float* out = a.probs + (long)row * V;
    for (int v = lane; v < V; v += 64) out[v] = expf(x[v] - mx) * inv;
  }
}

constexpr size_t LDS_LIMIT = 150 * 1024;

}  // namespace

// one LSTM cell step of layer `l` for n hypotheses with pool slots; KL_ERR_SHAPE = not applicable (the caller takes the
// launch-per-layer kernels of lstm_step.hip)
int kl_launch_inc_cell(const KlIncCellArgs& p, hipStream_t stream) {
  const int W = p.W;
  if (p.n < 1 || (W & 127) || !p.pool || !p.slot_in || !p.slot_out || !p.UT_hi) return KL_ERR_SHAPE;
  const bool lo = p.split == 3;
  if (lo && (!p.UT_lo || (p.x_off >= 0 && !p.KT_lo))) return KL_ERR_ARG;
  if (p.x_off >= 0 && !p.KT_hi) return KL_ERR_ARG;
  const int K = p.x_off >= 0 ? 2 * W : W;
  int nmt = p.n > 128 ? 2 : 1;
  if ((size_t)16 * nmt * K * 4 > LDS_LIMIT) nmt = 1;
  size_t lds = (size_t)16 * nmt * K * 4;
  if (lds < (size_t)4 * nmt * 4 * 16 * 17 * 4) lds = (size_t)4 * nmt * 4 * 16 * 17 * 4;      // (the partial tiles re-use the planes' room)
  if (lds > LDS_LIMIT) return KL_ERR_SHAPE;
  IncCell a;
  a.n = p.n; a.W = W; a.pool = p.pool; a.slot_ld = p.slot_ld; a.slot_in = p.slot_in; a.slot_out = p.slot_out;
  a.h_off = p.h_off; a.c_off = p.c_off; a.x_off = p.x_off;
  a.UT_hi = p.UT_hi; a.UT_lo = p.UT_lo; a.KT_hi = p.KT_hi; a.KT_lo = p.KT_lo;
  a.T1 = p.T1; a.i1 = p.i1; a.T2 = p.T2; a.i2 = p.i2; a.bias = p.bias;
  dim3 grid(W / 16, (p.n + 16 * nmt - 1) / (16 * nmt));
#define KL_IC_CASE(NMT_, LO_)                                                                                               \
  do {                                                                                                                      \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&inc_cell_kernel<NMT_, LO_>),                                    \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return KL_ERR_LAUNCH;     \
    hipLaunchKernelGGL((inc_cell_kernel<NMT_, LO_>), grid, dim3(256), lds, stream, a);                                      \
  } while (0)
  if (nmt == 2) { if (lo) KL_IC_CASE(2, true); else KL_IC_CASE(2, false); }
  else { if (lo) KL_IC_CASE(1, true); else KL_IC_CASE(1, false); }
#undef KL_IC_CASE
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

// probs[n][V] = softmax(h . E^T) for the top layer's new states; KL_ERR_SHAPE = not applicable (thin GEMM + softmax kernel)
int kl_launch_logits_softmax(const float* pool, long slot_ld, const int* slot_out, int h_off, const bf16_t* E_hi, const bf16_t* E_lo,
                             int n, int W, int V, int split, float* probs, hipStream_t stream) {
  if (n < 1 || V < 1 || (W & 127) || !pool || !slot_out || !E_hi || !probs) return KL_ERR_SHAPE;
  const bool lo = split == 3;
  if (lo && !E_lo) return KL_ERR_ARG;
  const int Vt = (V + 15) & ~15;
  const size_t lds = (size_t)16 * W * 4 + (size_t)16 * (Vt + 1) * 4;
  if (lds > LDS_LIMIT) return KL_ERR_SHAPE;
  LogitsSoftmax a;
  a.n = n; a.W = W; a.V = V; a.pool = pool; a.slot_ld = slot_ld; a.slot_out = slot_out; a.h_off = h_off;
  a.E_hi = E_hi; a.E_lo = E_lo; a.probs = probs;
  if (lo) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&logits_softmax_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return KL_ERR_LAUNCH;
    hipLaunchKernelGGL((logits_softmax_kernel<true>), dim3((n + 15) / 16), dim3(512), lds, stream, a);
  } else {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&logits_softmax_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return KL_ERR_LAUNCH;
    hipLaunchKernelGGL((logits_softmax_kernel<false>), dim3((n + 15) / 16), dim3(512), lds, stream, a);
  }
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}
