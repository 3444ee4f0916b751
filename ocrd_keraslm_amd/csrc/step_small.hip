// Incremental step for 96..255 hypotheses (the reference's callers: rate_best feeds at most 128 rows per call, generate
// at most 256; rating.py:49, 704, 809).
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
// Measured at width 512, depth 2 (round 3): 128 hypotheses 36.2 us per step against 44.3 (10.4 + 17.6 us for the two layers
// against 2 x 17.5); at 64 and 32 hypotheses the four-unit workgroups stay faster (28.8 / 27.7 us against 35.0 / 34.7: with
// few rows the weights dominate a workgroup's bytes and the finer column split spreads them over more CUs), so the launcher
// is only asked from 96 rows on.  Tried and dropped: the output layer with its softmax in one launch (a workgroup = 16
// hypotheses x all characters, so that the row maximum and sum never leave the CU): 22 us at 128 hypotheses -- eight
// workgroups each pulling all of E -- against 5.9 + 4.6 us for the thin GEMM and the softmax kernel.
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
  // ---- everything the epilogue will need is requested FIRST: this thread's cell = (row lr of each row tile, unit u0 + ej); the
  // table rows, the bias and c_prev depend only on the index arrays, and a 256-thread workgroup alone on its CU has nothing to
  // hide a late load behind
  const int ej = tid & 15;
  int e_in[NMT], e_out[NMT];
  float ez[NMT][4], ecp[NMT];
#pragma unroll
  for (int m = 0; m < NMT; ++m) {
    const int row = min(r0 + m * 16 + (tid >> 4), a.n - 1);
    e_in[m] = a.slot_in[row];
    e_out[m] = a.slot_out[row];
    const long t1 = a.T1 ? (long)(a.i1 ? a.i1[row] : row) * 4 * W : 0, t2 = a.T2 ? (long)(a.i2 ? a.i2[row] : row) * 4 * W : 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v = a.bias ? a.bias[g * W + u0 + ej] : 0.f;
      if (a.T1) v += a.T1[t1 + g * W + u0 + ej];
      if (a.T2) v += a.T2[t2 + g * W + u0 + ej];
      ez[m][g] = v;
    }
    ecp[m] = a.pool[(long)e_in[m] * a.slot_ld + a.c_off + u0 + ej];
  }

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

  // ---- activation rows: coalesced, split, into LDS.  Eight pieces per thread are requested before the first one is used (a
  // loop that loads, converts and stores piece by piece is a chain of sixteen or more memory latencies); the rows' slots sit
  // in lanes 0 .. ROWS - 1 of every wave and are broadcast from there (a piece's row is the same for a whole wave)
  {
    const int my_row = min(r0 + (lane < ROWS ? lane : 0), a.n - 1);
    const int sl_in = a.slot_in[my_row], sl_out = a.slot_out[my_row];
    const int per_row = K >> 2;                    // float4 pieces per row (a multiple of 32: a wave's 64 pieces lie in one row or two)
    const int total = ROWS * per_row;
    for (int e0 = 0; e0 < total; e0 += 8 * 256) {
      float4 v[8];
      int rr[8], kk[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int e = e0 + i * 256 + tid;
        const int ec = e < total ? e : total - 1;
        rr[i] = ec / per_row;
        kk[i] = (ec - rr[i] * per_row) * 4;
        // (per_row >= 64: the 64 pieces of a wave instruction lie in one row -- its slot comes from one lane)
        const int ru = __builtin_amdgcn_readfirstlane(rr[i]);
        const int so = __builtin_amdgcn_readlane(sl_out, ru), si = __builtin_amdgcn_readlane(sl_in, ru);
        const float* src = (a.x_off >= 0 && kk[i] < W) ? a.pool + (long)so * a.slot_ld + a.x_off + kk[i]
                                                       : a.pool + (long)si * a.slot_ld + a.h_off + (kk[i] - (a.x_off >= 0 ? W : 0));
        v[i] = *reinterpret_cast<const float4*>(src);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (e0 + i * 256 + tid < total) {
          uint2 hi, lo;
          split4<LO>(v[i], hi, lo);
          const unsigned off = a_off(rr[i], kk[i], K);
          *reinterpret_cast<uint2*>(plane_hi + off) = hi;
          if (LO) *reinterpret_cast<uint2*>(plane_lo + off) = lo;
        }
      }
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
    const int row = r0 + m * 16 + (tid >> 4);
    if (row >= a.n) continue;
    const int u = u0 + ej;
    float z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v = ez[m][g];
#pragma unroll
      for (int w = 0; w < 4; ++w) v += part[(((w * NMT + m) * 4 + g) * 16 + (tid >> 4)) * 17 + ej];
      z[g] = v;
    }
    const float gi = sigmoidf_(z[0]), gf = sigmoidf_(z[1]), gg = tanhf_(z[2]), go = sigmoidf_(z[3]);
    const float c = gf * ecp[m] + gi * gg;
    const float h = go * tanhf_(c);
    float* out = a.pool + (long)e_out[m] * a.slot_ld;
    out[a.c_off + u] = c;
    out[a.h_off + u] = h;
  }
}

constexpr size_t LDS_LIMIT = 150 * 1024;

}  // namespace

// one LSTM cell step of layer `l` for n hypotheses with pool slots; KL_ERR_SHAPE = not applicable (the caller takes the
// launch-per-layer kernels of lstm_step.hip)
int kl_launch_inc_cell(const KlIncCellArgs& p, hipStream_t stream) {
  const int W = p.W;
  if (p.n < 1 || (W & 255) || !p.pool || !p.slot_in || !p.slot_out || !p.UT_hi) return KL_ERR_SHAPE;
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
