// One-layer LSTM scans for width 128 -- the reference's own model sizes (the published model is depth 2, width 128, length
// 256: README.md:252-254, ocrd-tool.json:52-55; scripts/run.py:34 defaults to width 128) -- with NO hand-off between
// workgroups at all (round 4).
//
// At width 128 a layer's recurrent weights are 128 KiB in bf16: ONE workgroup's registers hold them (64 per lane at 512
// threads).  So a workgroup owns a 16-row block of streams with ALL hidden units of the layer and runs the whole window on
// its own: the state tile goes from one step to the next through its own LDS, one workgroup barrier per step, nothing to
// publish, nothing to poll, no co-residency assumption (any number of workgroups; a batch of 4096 streams fills the 256 CUs).
// The thin fused scans this replaces (lstm_scan.hip: 16-unit workgroups, W / 16 x layers of them per row block exchanging state
// tiles through L2 every step) spent the step on that exchange -- 5.4 us per step at 1024 streams (profiles/
// r04_w128_B1024_kernel_stats.csv: 1.38 ms forward + 2 x 0.88 ms backward per window, + 0.78 ms of bias-gradient column sums
// that the backward scan here accumulates on the way).
// Layer-sequential as the wide scans: the input side of a layer comes from one product over all steps (P = X . K^T + b by
// the ring GEMM, layer 0: the table gather), the gradient from above likewise (dX = dZ_{l+1} . K_{l+1}^T).
// Everything a step reads from or writes to memory is staged through LDS as whole rows (16-byte pieces, consecutive lanes on
// consecutive addresses): the gate inputs a step ahead through registers, the outputs behind the step's barrier.
//
// Forward: wave w owns units 16 w .. 16 w + 15 as four MFMA column tiles (tile c: column 4 k + g = gate g of unit 16 w + 4 k + c),
// so that after the 4 x 4 lane-quad transpose a lane holds one row's gates of FOUR CONSECUTIVE units: 16-byte reads of the
// gate inputs, 8-byte writes of h / the gates.  Backward: wave w owns dh of units 16 w .. 16 w + 15 (one column tile, K = 4W =
// 16 k-steps against the dZ tile of the step after), a lane = 4 rows x 1 unit, the cell state carried in registers.
// Restates the Keras LSTM cell (rating.py:130-145: gates i, f, c, o; recurrent activation sigmoid) and its gradient as
// oracle/lstm_oracle.py: forward_window / backward_window do.
#include <stdlib.h>
#include <string.h>

#include "kl_common.h"
#include "kl_kernels.h"

namespace {

#include "kl_scan_common.h"

#include "kl_scan2_helpers.h"

constexpr int W8 = 128, NT8 = 512;
// forward LDS (bytes): h tiles [2][16][272] | gate inputs [2][16][2112] | gates out [2][16][1056] | cell states out [2][16][576] |
// masked h out [2][16][272]   (row strides padded so that the 16 lanes the LDS serves together fall on different banks)
constexpr int F_H_LD = 272, F_P_LD = 2112, F_G_LD = 1056, F_C_LD = 576;
constexpr int F_HT = 0, F_PB = F_HT + 2 * 16 * F_H_LD, F_GS = F_PB + 2 * 16 * F_P_LD, F_CS = F_GS + 2 * 16 * F_G_LD,
              F_HS = F_CS + 2 * 16 * F_C_LD, F_LDS = F_HS + 2 * 16 * F_H_LD;
// backward LDS: dZ tiles [2][16][1040] | gates [2][16][1024] | c_{t-1} [2][16][512] | dH [2][16][512]
constexpr int B_Z_LD = 1040;
constexpr int B_ZT = 0, B_GB = B_ZT + 2 * 16 * B_Z_LD, B_CB = B_GB + 2 * 16 * 1024, B_DH = B_CB + 2 * 16 * 512, B_LDS = B_DH + 2 * 16 * 512,
              B_LDS_XIN = B_DH + 2 * 16 * B_Z_LD;      // (XIN: the layer above's dZ tiles instead of the dH rows)

__device__ __forceinline__ unsigned pack2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }

// KIN: the layer's input contraction happens HERE as well -- z = bias + X[t] . K^T + h[t-1] . U^T with K resident beside U (2 x 64
// registers) and the 16 x 128 input rows X[t] (the layer below's -- masked -- outputs) staged like the state tile -- instead of
// f32 gate-input rows P from a product over all steps: a step reads 4 KiB instead of 32, the P rows (2.1 GB written and read
// per window at 4096 streams) and their GEMM disappear.
// POLL / PUB (the layers of a window in ONE launch, lstm_scan_fwd_w128_multi_kernel): PUB -- the rows the layer above reads (h,
// or the masked h where it has a mask) leave as write-through stores, into an array the caller pre-filled with 0xFFFF halfwords;
// POLL -- the input rows X[t] are taken with coherent loads a step ahead and asked for again until none of their halfwords is
// that sentinel: a layer follows the one below it a step or more behind, on other CUs, nothing waits for a launch boundary.
// NC >= 0 (layer 0, a.P == null): the gate inputs come straight from the look-up tables -- EK[idx[b, t]] + sum over NC context
// variables of CtxK_n[ctx[b, t, n]] (f32 rows of 4W, L2-resident), the bias from the accumulators' start -- instead of f32 rows P
// that a gather kernel wrote for every position first (537 MB written and read back per window at 1024 streams): a thread
// serves ONE stream row (4 pieces of 16 bytes, 512 bytes apart), so that it holds one set of indices, fetched two steps ahead.
template <bool KIN, bool POLL, bool PUB, int NC = -1>
__device__ __forceinline__ void fwd_w128_body(const KlScanFwdWide& a, const int rb) {
  constexpr bool GIN = NC >= 0;
  constexpr int NCX = NC > 0 ? NC : 1;
  constexpr int W = W8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int B = a.B, T = a.T;
  const int row0 = rb * 16;
  const int nrow = min(16, B - row0);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int jr = lane & 3, a4 = (lane >> 2) & 3, q4 = lane >> 4;
  const int crow = 4 * q4 + jr, cu = 16 * wave + 4 * a4;      // this lane's row and the first of its four units

  // resident weights: B fragments of four column tiles, all of K = 128
  u32x4 bu[4][4];
  {
    const int col = lane & 15;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const long wrow = ((long)(col & 3) * W + 16 * wave + 4 * (col >> 2) + c) * W + (lane >> 4) * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) bu[c][j] = *reinterpret_cast<const u32x4*>(a.UT + wrow + j * 32);
    }
  }
  u32x4 bk[KIN ? 4 : 1][4];
  float bias_c[4] = {0.f, 0.f, 0.f, 0.f};      // (KIN / GIN: the bias of this lane's column in each tile -- the accumulators start from it)
  if (KIN || GIN) {
    const int col = lane & 15;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const long wrow = ((long)(col & 3) * W + 16 * wave + 4 * (col >> 2) + c) * W + (lane >> 4) * 8;
      if (KIN) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bk[c][j] = *reinterpret_cast<const u32x4*>(a.KT + wrow + j * 32);
      }
      bias_c[c] = a.bias[(col & 3) * W + 16 * wave + 4 * (col >> 2) + c];
    }
  }
  // cell states and keep-masks of this lane's four cells
  float cst[4], mk[4];
  {
    const long at = (long)(row0 + min(crow, nrow - 1)) * W + cu;
    const float4 c0 = *reinterpret_cast<const float4*>(a.C + at);
    cst[0] = c0.x; cst[1] = c0.y; cst[2] = c0.z; cst[3] = c0.w;
    if (a.mask) {
      const float4 m = *reinterpret_cast<const float4*>(a.mask + at);
      mk[0] = m.x; mk[1] = m.y; mk[2] = m.z; mk[3] = m.w;
    } else {
      mk[0] = mk[1] = mk[2] = mk[3] = 1.f;
    }
  }
  // the rows a thread moves between memory and LDS: gate inputs (4 pieces of 16 bytes per step), gates out (2), cell states (1),
  // h (threads 0 .. 255) / masked h (256 .. 511)
  const int h_i = tid & 255, h_row = h_i >> 4, h_seg = h_i & 15;
  const int c_row = tid >> 5, c_seg = tid & 31;
  auto p_load = [&](int t, int k) __attribute__((always_inline)) {
    const int i = tid + NT8 * k, row = i >> 7, seg = i & 127;
    float4 v = float4{0.f, 0.f, 0.f, 0.f};
    if (row < nrow) v = *reinterpret_cast<const float4*>(a.P + ((long)t * B + row0 + row) * 4 * W + seg * 4);
    return v;
  };
  auto p_put = [&](int buf, int k, float4 v) __attribute__((always_inline)) {
    const int i = tid + NT8 * k, row = GIN ? (tid >> 5) : (i >> 7), seg = GIN ? (tid & 31) + 32 * k : (i & 127);
    *reinterpret_cast<float4*>(smem + F_PB + (buf * 16 + row) * F_P_LD + seg * 16) = v;
  };
  // GIN: this thread's stream row and its indices (rows beyond the batch: the last one's)
  struct Ids { int id; int cx[NCX]; };
  const long g_src = (long)(row0 + min(tid >> 5, nrow - 1)) * T;
  auto ids_load = [&](int t) __attribute__((always_inline)) {
    Ids v;
    v.id = a.idx[g_src + t];
#pragma unroll
    for (int n = 0; n < NCX; ++n) v.cx[n] = NC > 0 ? a.ctx[(g_src + t) * NC + n] : 0;
    return v;
  };
  // (the character row's and the context rows' pieces stay apart until they are laid into LDS: summed where they are loaded,
  //  the sum is a wait for the loads right behind them -- the whole round trip to L2 at the top of every step)
  auto g_load = [&](const Ids& ids, int k, float4& ek, float4 (&ck)[NCX]) __attribute__((always_inline)) {
    const int seg = (tid & 31) + 32 * k;
    ek = *reinterpret_cast<const float4*>(a.EK + (long)ids.id * 4 * W + seg * 4);
#pragma unroll
    for (int n = 0; n < NCX; ++n)
      ck[n] = n < NC ? *reinterpret_cast<const float4*>(a.CtxK[n] + (long)ids.cx[n] * 4 * W + seg * 4) : float4{0.f, 0.f, 0.f, 0.f};
  };
  auto g_sum = [&](float4 v, const float4 (&ck)[NCX]) __attribute__((always_inline)) {
#pragma unroll
    for (int n = 0; n < NCX; ++n) {
      if (n < NC) { v.x += ck[n].x; v.y += ck[n].y; v.z += ck[n].z; v.w += ck[n].w; }
    }
    return v;
  };
  // (KIN: the input rows X[t] -- 16 x 256 bytes, threads 0 .. 255 -- take the place of the gate-input rows; tile [2][16][272] at F_PB)
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(KIN ? a.X : nullptr, KIN ? (long)T * B * W * 2 : 0);
  auto x_load = [&](int t) __attribute__((always_inline)) {
    uint4 v = uint4{0u, 0u, 0u, 0u};
    if (tid < 256 && h_row < nrow) {
      if (POLL) v = load16_sc1(rs_x, (unsigned)((((long)t * B + row0 + h_row) * W + h_seg * 8) * 2));
      else v = *reinterpret_cast<const uint4*>(a.X + ((long)t * B + row0 + h_row) * W + h_seg * 8);
    }
    return v;
  };
  // POLL: the rows of step t as they are now, asked for again until the layer below has published all of them (bounded)
  auto x_await = [&](int t, uint4 v) __attribute__((always_inline)) {
    if (!POLL || tid >= 256) return v;
    for (unsigned spin = 0; spin < SPIN_LIMIT; ++spin) {
      const bool ok = h_row >= nrow || sentinel_free(sentinel_bits(v));
      if (__all(ok)) return v;
      if ((spin & 63) == 63 && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
      __builtin_amdgcn_s_sleep(2);
      v = x_load(t);
    }
    __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
  };
  auto x_put = [&](int buf, uint4 v) __attribute__((always_inline)) {
    if (tid < 256) *reinterpret_cast<uint4*>(smem + F_PB + (buf * 16 + h_row) * F_H_LD + h_seg * 16) = v;
  };
  // ---- prologue: the carried-in h rows, the gate inputs of step 0
  if (tid < 256) {
    uint4 v = uint4{0u, 0u, 0u, 0u};
    if (h_row < nrow) v = *reinterpret_cast<const uint4*>(a.H + ((long)row0 + h_row) * W + h_seg * 8);
    *reinterpret_cast<uint4*>(smem + F_HT + h_row * F_H_LD + h_seg * 16) = v;
  }
  Ids ids_nx;      // GIN: the indices of the step after the one whose rows are on their way
  if (KIN) {
    x_put(0, x_await(0, x_load(0)));
  } else if (GIN) {
    const Ids i0 = ids_load(0);
    ids_nx = ids_load(min(1, T - 1));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 ek, ck[NCX];
      g_load(i0, k, ek, ck);
      p_put(0, k, g_sum(ek, ck));
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) p_put(0, k, p_load(0, k));
  }
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    const int p = t & 1;
    // the gate inputs of the next step are on their way while this one computes (two steps ahead measured the same: the step
    // is bound by its ~350 vector instructions per wave -- 4 cells per lane, 10 transcendentals each --, not by the loads)
    float4 pn[KIN ? 1 : 4];
    float4 pc[GIN ? 4 : 1][NCX];      // (table mode: the context rows' pieces of the next step)
    uint4 xn = uint4{0u, 0u, 0u, 0u};
    if (t + 1 < T) {
      if (KIN) {
        xn = x_load(t + 1);
      } else if (GIN) {
#pragma unroll
        for (int k = 0; k < 4; ++k) g_load(ids_nx, k, pn[k], pc[k]);
        ids_nx = ids_load(min(t + 2, T - 1));
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) pn[k] = p_load(t + 1, k);
      }
    }
    // ---- h[t-1] . U^T (+ X[t] . K^T): 16 rows x this wave's 64 columns
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{bias_c[c], bias_c[c], bias_c[c], bias_c[c]};
    {
      const unsigned char* tb = smem + F_HT + (p * 16 + (lane & 15)) * F_H_LD + (lane >> 4) * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bf16x8 fa = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(tb + j * 64));
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = mfma16(fa, __builtin_bit_cast(bf16x8, bu[c][j]), acc[c]);
      }
      if (KIN) {
        const unsigned char* xb = smem + F_PB + (p * 16 + (lane & 15)) * F_H_LD + (lane >> 4) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bf16x8 fx = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(xb + j * 64));
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c] = mfma16(fx, __builtin_bit_cast(bf16x8, bk[c][j]), acc[c]);
        }
      }
    }
    // ---- the cell: after the quad transpose lane = (row, unit), registers = gates
    f32x4 pin[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (KIN) pin[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      else pin[g] = *reinterpret_cast<const f32x4*>(smem + F_PB + (p * 16 + crow) * F_P_LD + (g * W + cu) * 4);
    }
    float hv[4], gi[4], gf[4], gg[4], go[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      quad_transpose(acc[c], jr);
      gi[c] = fast_sigmoid(acc[c][0] + pin[0][c]);
      gf[c] = fast_sigmoid(acc[c][1] + pin[1][c]);
      gg[c] = fast_tanh(acc[c][2] + pin[2][c]);
      go[c] = fast_sigmoid(acc[c][3] + pin[3][c]);
      cst[c] = gf[c] * cst[c] + gi[c] * gg[c];
      hv[c] = go[c] * fast_tanh(cst[c]);
    }
    *reinterpret_cast<uint2*>(smem + F_HT + ((p ^ 1) * 16 + crow) * F_H_LD + cu * 2) = uint2{pack2(hv[0], hv[1]), pack2(hv[2], hv[3])};
    if (a.Hd)
      *reinterpret_cast<uint2*>(smem + F_HS + (p * 16 + crow) * F_H_LD + cu * 2) =
          uint2{pack2(hv[0] * mk[0], hv[1] * mk[1]), pack2(hv[2] * mk[2], hv[3] * mk[3])};
    *reinterpret_cast<float4*>(smem + F_CS + (p * 16 + crow) * F_C_LD + cu * 4) = float4{cst[0], cst[1], cst[2], cst[3]};
    {
      unsigned char* gs = smem + F_GS + (p * 16 + crow) * F_G_LD + cu * 2;
      *reinterpret_cast<uint2*>(gs) = uint2{pack2(gi[0], gi[1]), pack2(gi[2], gi[3])};
      *reinterpret_cast<uint2*>(gs + 256) = uint2{pack2(gf[0], gf[1]), pack2(gf[2], gf[3])};
      *reinterpret_cast<uint2*>(gs + 512) = uint2{pack2(gg[0], gg[1]), pack2(gg[2], gg[3])};
      *reinterpret_cast<uint2*>(gs + 768) = uint2{pack2(go[0], go[1]), pack2(go[2], go[3])};
    }
    if (t + 1 < T) {
      if (KIN) {
        x_put(p ^ 1, x_await(t + 1, xn));
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) p_put(p ^ 1, k, GIN ? g_sum(pn[k], pc[GIN ? k : 0]) : pn[k]);
      }
    }
    __syncthreads();
    // ---- this step's rows to memory, whole rows
    const long trow = (long)t * B + row0;
    if (a.G) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = tid + NT8 * k, row = i >> 6, seg = i & 63;
        if (row < nrow)
          *reinterpret_cast<uint4*>(a.G + (trow + row) * 4 * W + seg * 8) = *reinterpret_cast<const uint4*>(smem + F_GS + (p * 16 + row) * F_G_LD + seg * 16);
      }
    }
    if (c_row < nrow)
      *reinterpret_cast<float4*>(a.C + (trow + B + c_row) * W + c_seg * 4) = *reinterpret_cast<const float4*>(smem + F_CS + (p * 16 + c_row) * F_C_LD + c_seg * 16);
    if (h_row < nrow) {
      // (PUB, a.sentinel == 2 -- rolling sentinels: the rows of step t + 2 become sentinels again while those of step t are
      //  published, by the thread that is going to publish them; only the first two steps start out armed)
      const bool arm = PUB && a.sentinel == 2 && t + 2 < T;
      const uint4 ones = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
      if (tid < 256) {
        const uint4 v = *reinterpret_cast<const uint4*>(smem + F_HT + ((p ^ 1) * 16 + h_row) * F_H_LD + h_seg * 16);
        if (PUB) {
          const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.H, (long)(T + 1) * B * W * 2);
          if (arm && !a.Hd) store16_sc1(rs, (unsigned)(((trow + 3L * B + h_row) * W + h_seg * 8) * 2), ones);
          store16_sc1(rs, (unsigned)(((trow + B + h_row) * W + h_seg * 8) * 2), v);
        } else {
          *reinterpret_cast<uint4*>(a.H + (trow + B + h_row) * W + h_seg * 8) = v;
        }
      } else if (a.Hd) {
        const uint4 v = *reinterpret_cast<const uint4*>(smem + F_HS + (p * 16 + h_row) * F_H_LD + h_seg * 16);
        if (PUB) {
          const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.Hd, (long)T * B * W * 2);
          if (arm) store16_sc1(rs, (unsigned)(((trow + 2L * B + h_row) * W + h_seg * 8) * 2), ones);
          store16_sc1(rs, (unsigned)(((trow + h_row) * W + h_seg * 8) * 2), v);
        } else {
          *reinterpret_cast<uint4*>(a.Hd + (trow + h_row) * W + h_seg * 8) = v;
        }
      }
    }
  }
}

// MODE 0: gate inputs from f32 rows P; 1: KIN; 2 + n: from the look-up tables with n context variables
template <int MODE>
__global__ __launch_bounds__(NT8, 1) void lstm_scan_fwd_w128_kernel(const KlScanFwdWide a) {
  fwd_w128_body<MODE == 1, false, false, MODE >= 2 ? MODE - 2 : -1>(a, blockIdx.x);
}

// all layers of a window in one launch: workgroup = (layer = blockIdx / n_rb, row block = blockIdx % n_rb) -- the lowest layer's
// workgroups first, every workgroup resident at once (the launcher checks layers x row blocks <= CUs)
__global__ __launch_bounds__(NT8, 1) void lstm_scan_fwd_w128_multi_kernel(const KlScanFwdWide a0, const KlScanFwdWide a1, const KlScanFwdWide a2,
                                                                          const KlScanFwdWide a3, const int L, const int n_rb) {
  const int layer = blockIdx.x / n_rb, rb = blockIdx.x - layer * n_rb;
  if (layer == 0) {      // (L >= 2: always publishing)
    if (a0.P) fwd_w128_body<false, false, true>(a0, rb);
    else if (a0.n_ctx == 0) fwd_w128_body<false, false, true, 0>(a0, rb);
    else if (a0.n_ctx == 1) fwd_w128_body<false, false, true, 1>(a0, rb);
    else fwd_w128_body<false, false, true, 2>(a0, rb);
  } else if (layer == 1) {
    if (L > 2) fwd_w128_body<true, true, true>(a1, rb);
    else fwd_w128_body<true, true, false>(a1, rb);
  } else if (layer == 2) {
    if (L > 3) fwd_w128_body<true, true, true>(a2, rb);
    else fwd_w128_body<true, true, false>(a2, rb);
  } else {
    fwd_w128_body<true, true, false>(a3, rb);
  }
}

// dh[t] = dH[t] (from above, all steps at once: the softmax side or the layer above's dX product) * mask + dZ[t+1] . U^T;
// dZ[t] from the gate derivatives; db summed on the way
// XIN: the gradient from above is contracted HERE -- dh[t] = mask * (dZ_above[t] . K_above^T) + dZ[t+1] . U^T with K_above
// resident beside U and the 16 x 512 rows of the layer above's dZ[t] staged like this layer's own tile -- instead of f32 dH rows
// from a product over all steps (a.Kn[1] / a.dZ[1] = the layer above's input kernel [W][4W] and dZ [T*B][4W]).
// LY: the layer's slot in the argument arrays (the one-layer launches: 0; a.Kn / a.dZ [LY + 1] = the layer above).
// POLL / PUB (all layers of a window in ONE launch, lstm_scan_bwd_w128_multi_kernel): PUB -- the dZ rows leave as write-through
// stores into an array the caller pre-filled with 0xFFFF halfwords; POLL -- the layer above's dZ rows are taken with coherent
// loads a step ahead and asked for again until none of their halfwords is that sentinel.
template <bool XIN, bool POLL, bool PUB, int LY>
__device__ __forceinline__ void bwd_w128_body(const KlScanBwd& a, const int rb, float* const db) {
  constexpr int W = W8;
  constexpr int LA = LY + 1 < KL_SCAN_MAXL ? LY + 1 : LY;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int B = a.B, T = a.T;
  const int row0 = rb * 16;
  const int nrow = min(16, B - row0);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int col = lane & 15, q4 = lane >> 4;
  const int u = 16 * wave + col;                      // this lane's unit; its rows: 4 q4 + r

  // resident weights: B fragments of this wave's 16 output units, K = 4W = 512 (U natural layout [W][4W]: row = unit)
  u32x4 bu[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) bu[j] = *reinterpret_cast<const u32x4*>(a.Un[LY] + (long)u * 4 * W + j * 32 + q4 * 8);
  u32x4 bk[XIN ? 16 : 1];
  if (XIN) {
#pragma unroll
    for (int j = 0; j < 16; ++j) bk[j] = *reinterpret_cast<const u32x4*>(a.Kn[LA] + (long)u * 4 * W + j * 32 + q4 * 8);
  }
  float ccur[4], dc[4], mk[4], dbacc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = min(4 * q4 + r, nrow - 1);
    ccur[r] = a.C[LY][((long)T * B + row0 + row) * W + u];       // c_{T-1} = block T
    mk[r] = a.mask[LY] ? a.mask[LY][((long)row0 + row) * W + u] : 1.f;
    dc[r] = 0.f;
    dbacc[r] = 0.f;
  }
  // rows between memory and LDS: gates (2 pieces of 16 bytes per thread and step), c_{t-1} (1), dH (1), dZ out (2)
  const int c_row = tid >> 5, c_seg = tid & 31;
  struct In { uint4 g[2]; float4 c, d; uint4 za[2]; };
  const __amdgpu_buffer_rsrc_t rs_za = make_rsrc(XIN ? a.dZ[LA] : nullptr, XIN ? (long)T * B * 4 * W * 2 : 0);
  auto za_load = [&](int t, int k) __attribute__((always_inline)) {
    const int i = tid + NT8 * k, row = i >> 6, seg = i & 63;
    uint4 v = uint4{0u, 0u, 0u, 0u};
    if (row < nrow) {
      if (POLL) v = load16_sc1(rs_za, (unsigned)((((long)t * B + row0 + row) * 4 * W + seg * 8) * 2));
      else v = *reinterpret_cast<const uint4*>(a.dZ[LA] + ((long)t * B + row0 + row) * 4 * W + seg * 8);
    }
    return v;
  };
  // POLL: the layer above's rows of step t as they are now, asked for again until all of them have been published (bounded)
  auto za_await = [&](int t, In& v) __attribute__((always_inline)) {
    if (!POLL) return;
    for (unsigned spin = 0; spin < SPIN_LIMIT; ++spin) {
      const bool ok = sentinel_free(sentinel_bits(v.za[0]) | sentinel_bits(v.za[1]));      // (rows beyond the batch: zeros)
      if (__all(ok)) return;
      if ((spin & 63) == 63 && __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
      __builtin_amdgcn_s_sleep(2);
      v.za[0] = za_load(t, 0);
      v.za[1] = za_load(t, 1);
    }
    __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto in_load = [&](int t) __attribute__((always_inline)) {
    In v;
    const long trow = (long)t * B + row0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + NT8 * k, row = i >> 6, seg = i & 63;
      v.g[k] = uint4{0u, 0u, 0u, 0u};
      v.za[k] = XIN ? za_load(t, k) : uint4{0u, 0u, 0u, 0u};
      if (row < nrow) {
        v.g[k] = *reinterpret_cast<const uint4*>(a.G[LY] + (trow + row) * 4 * W + seg * 8);
      }
    }
    v.c = float4{0.f, 0.f, 0.f, 0.f};
    v.d = float4{0.f, 0.f, 0.f, 0.f};
    if (c_row < nrow) {
      v.c = *reinterpret_cast<const float4*>(a.C[LY] + (trow + c_row) * W + c_seg * 4);      // block t = c_{t-1}
      if (!XIN) v.d = *reinterpret_cast<const float4*>(a.dH + (trow + c_row) * W + c_seg * 4);
    }
    return v;
  };
  // (XIN: the rows of the layer above's dZ -- tile [2][16][1040] -- take the place of the dH rows, from B_DH on)
  auto in_put = [&](int buf, const In& v) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + NT8 * k, row = i >> 6, seg = i & 63;
      *reinterpret_cast<uint4*>(smem + B_GB + (buf * 16 + row) * 1024 + seg * 16) = v.g[k];
      if (XIN) *reinterpret_cast<uint4*>(smem + B_DH + (buf * 16 + row) * B_Z_LD + seg * 16) = v.za[k];
    }
    *reinterpret_cast<float4*>(smem + B_CB + (buf * 16 + c_row) * 512 + c_seg * 16) = v.c;
    if (!XIN) *reinterpret_cast<float4*>(smem + B_DH + (buf * 16 + c_row) * 512 + c_seg * 16) = v.d;
  };
  {
    In first = in_load(T - 1);
    za_await(T - 1, first);
    in_put((T - 1) & 1, first);
  }
  __syncthreads();

  for (int t = T - 1; t >= 0; --t) {
    const int p = t & 1;
    In nxt;
    if (t > 0) nxt = in_load(t - 1);
    // ---- dZ[t+1] . U^T (nothing comes back from beyond the window)
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t + 1 < T) {
      const unsigned char* tb = smem + B_ZT + ((p ^ 1) * 16 + (lane & 15)) * B_Z_LD + q4 * 16;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        acc = mfma16(__builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(tb + j * 64)), __builtin_bit_cast(bf16x8, bu[j]), acc);
    }
    f32x4 above = f32x4{0.f, 0.f, 0.f, 0.f};
    if (XIN) {
      const unsigned char* tb = smem + B_DH + (p * 16 + (lane & 15)) * B_Z_LD + q4 * 16;
#pragma unroll
      for (int j = 0; j < 16; ++j)
        above = mfma16(__builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(tb + j * 64)), __builtin_bit_cast(bf16x8, bk[j]), above);
    }
    // ---- gate derivatives of this lane's four cells (rows 4 q4 + r, unit u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * q4 + r;
      const unsigned char* gb = smem + B_GB + (p * 16 + row) * 1024 + u * 2;
      const float gi = bf2f(*reinterpret_cast<const bf16_t*>(gb)), gf = bf2f(*reinterpret_cast<const bf16_t*>(gb + 256));
      const float gg = bf2f(*reinterpret_cast<const bf16_t*>(gb + 512)), go = bf2f(*reinterpret_cast<const bf16_t*>(gb + 768));
      const float cprev = *reinterpret_cast<const float*>(smem + B_CB + (p * 16 + row) * 512 + u * 4);
      const float dh = acc[r] + mk[r] * (XIN ? above[r] : *reinterpret_cast<const float*>(smem + B_DH + (p * 16 + row) * 512 + u * 4));
      const float tc = fast_tanh(ccur[r]);
      const float d_o = dh * tc;
      dc[r] += dh * go * (1.f - tc * tc);
      const float dzi = dc[r] * gg * gi * (1.f - gi), dzf = dc[r] * cprev * gf * (1.f - gf);
      const float dzg = dc[r] * gi * (1.f - gg * gg), dzo = d_o * go * (1.f - go);
      dc[r] *= gf;
      ccur[r] = cprev;
      unsigned char* zt = smem + B_ZT + (p * 16 + row) * B_Z_LD + u * 2;
      const bf16_t bi = f2bf(dzi), bff = f2bf(dzf), bg = f2bf(dzg), bo = f2bf(dzo);
      *reinterpret_cast<bf16_t*>(zt) = bi;
      *reinterpret_cast<bf16_t*>(zt + 256) = bff;
      *reinterpret_cast<bf16_t*>(zt + 512) = bg;
      *reinterpret_cast<bf16_t*>(zt + 768) = bo;
      dbacc[0] += bf2f(bi); dbacc[1] += bf2f(bff); dbacc[2] += bf2f(bg); dbacc[3] += bf2f(bo);      // (what the weight gradients see: the rounded values)
    }
    if (t > 0) {
      za_await(t - 1, nxt);
      in_put(p ^ 1, nxt);
    }
    __syncthreads();
    // ---- dZ[t] to memory, whole rows
    const long trow = (long)t * B + row0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + NT8 * k, row = i >> 6, seg = i & 63;
      if (row < nrow) {
        const uint4 v = *reinterpret_cast<const uint4*>(smem + B_ZT + (p * 16 + row) * B_Z_LD + seg * 16);
        if (PUB) {
          // (a.sentinel == 2 -- rolling sentinels: step t - 2 becomes sentinels again while step t is published; the first two start out armed)
          const __amdgpu_buffer_rsrc_t rs = make_rsrc(a.dZ[LY], (long)T * B * 4 * W * 2);
          if (a.sentinel == 2 && t >= 2)
            store16_sc1(rs, (unsigned)(((trow - 2L * B + row) * 4 * W + seg * 8) * 2), uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu});
          store16_sc1(rs, (unsigned)(((trow + row) * 4 * W + seg * 8) * 2), v);
        } else {
          *reinterpret_cast<uint4*>(a.dZ[LY] + (trow + row) * 4 * W + seg * 8) = v;
        }
      }
    }
  }
  // bias gradient: column sums of what was written (the four row groups of a wave, then one atomic per column and workgroup)
  if (db) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v = dbacc[g];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (q4 == 0) atomicAdd(db + g * W + u, v);
    }
  }
}

template <bool XIN>
__global__ __launch_bounds__(NT8, 1) void lstm_scan_bwd_w128_kernel(const KlScanBwd a) {
  bwd_w128_body<XIN, false, false, 0>(a, blockIdx.x, a.db);
}

// all layers of a window in one launch: workgroup = (layer = L - 1 - blockIdx / n_rb, row block = blockIdx % n_rb) -- the top
// layer's workgroups first, every workgroup resident at once (the launcher checks layers x row blocks <= CUs)
__global__ __launch_bounds__(NT8, 1) void lstm_scan_bwd_w128_multi_kernel(const KlScanBwd a) {
  const int L = a.L, n_rb = a.n_rb;
  const int layer = L - 1 - blockIdx.x / n_rb, rb = blockIdx.x % n_rb;
  if (layer == L - 1) {       // (takes the softmax side's dH rows, all steps there before the launch)
    if (layer == 1) bwd_w128_body<false, false, true, 1>(a, rb, a.db_l[1]);
    else if (layer == 2) bwd_w128_body<false, false, true, 2>(a, rb, a.db_l[2]);
    else bwd_w128_body<false, false, true, 3>(a, rb, a.db_l[3]);
  } else if (layer == 0) {
    bwd_w128_body<true, true, false, 0>(a, rb, a.db_l[0]);
  } else if (layer == 1) {
    bwd_w128_body<true, true, true, 1>(a, rb, a.db_l[1]);
  } else {
    bwd_w128_body<true, true, true, 2>(a, rb, a.db_l[2]);
  }
}


// ---------------------------------------------------------------- output layer of a training window at width 128
// logits = X . E^T, softmax, cross-entropy, its gradient AND the gradient into the top layer, dH = dlogits . E, in one pass over
// the rows: the embedding (<= 256 characters x 128 bf16 = 64 KiB) is stationary in registers in both orientations, 32 rows at
// a time go through LDS, and out go the bf16 gradient rows dlogits (the embedding's gradient reads them), the f32 rows dH and
// the per-row (loss, hit) pair.  The GEMM + softmax + GEMM this replaces wrote the logits as f32 (268 MB at 1024 streams), read
// them back and read dlogits once more.  Same rules as softmax_ce_v256_kernel / logits_ce_ws_kernel (rating.py:255-258 through
// Keras: probabilities clipped to [1e-7, 1 - 1e-7] -- no gradient outside --, padded positions count in the mean but carry no
// target, accuracy = first maximum equals the target; targets < -1 = a dummy stream of the caller's padding: counts for nothing).
struct KlCeW128 {
  const bf16_t* X;       // [M][128] (masked) outputs of the top layer, time-major rows r = t B + b
  const bf16_t* E;       // [Vp][128], rows beyond V zero
  const bf16_t* ET;      // [128][Vp]
  const int* tgt;        // [B][T]
  bf16_t* dlogits;       // [M][Vp]
  float* dH;             // [M][128]
  float* rowstat;        // [M][2]
  int M, B, T, V, Vp, n_wg, last_only;
  float inv_count;
};
constexpr int CE_ROWS = 64;
constexpr int CE_X_LD = 272, CE_Z_LD = 260, CE_G_LD = 528, CE_D_LD = 132;      // bytes / floats / bytes / floats
constexpr int CE_XT = 0, CE_ZL = CE_XT + 2 * CE_ROWS * CE_X_LD, CE_GL = CE_ZL + CE_ROWS * CE_Z_LD * 4, CE_LDS = CE_GL + CE_ROWS * CE_G_LD;

// 64 rows per tile.  Softmax pass: a wave takes FOUR rows at once, 16 lanes per row and 16 characters per lane (characters
// 64 k + 4 (l & 15) .. + 3, k = 0 .. 3): the reductions are four DPP steps inside the 16-lane rows and serve four rows each --
// with a row per pass and wave-wide reductions (the first cut, as logits_ce_ws_kernel) the pass was 150 vector instructions per
// row, 0.66 ms per window at 4096 streams against 0.17 ms of memory traffic.
__global__ __launch_bounds__(1024, 1) void logits_ce_w128_kernel(const KlCeW128 a) {
  constexpr int W = W8, ROWS = CE_ROWS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wg = blockIdx.x;
  const int n_tiles_all = (a.M + ROWS - 1) / ROWS;
  const int my_tiles = wg < n_tiles_all ? (n_tiles_all - wg + a.n_wg - 1) / a.n_wg : 0;
  if (my_tiles == 0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* zl = reinterpret_cast<float*>(smem + CE_ZL);
  const int V = a.V, Vp = a.Vp, nk = Vp >> 5;      // (k-steps of the dH contraction)

  // E for the logits: this wave's 16 characters, K = 128; E^T for dH: the 16 units of column tile wave & 7, K = Vp
  u32x4 bu[4], be[8];
  {
    const int ch = 16 * wave + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bu[j] = ch < Vp ? *reinterpret_cast<const u32x4*>(a.E + (long)ch * W + (lane >> 4) * 8 + j * 32) : u32x4{0u, 0u, 0u, 0u};
    const int un = 16 * (wave & 7) + (lane & 15);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      be[j] = j < nk ? *reinterpret_cast<const u32x4*>(a.ET + (long)un * Vp + (lane >> 4) * 8 + j * 32) : u32x4{0u, 0u, 0u, 0u};
  }
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(a.X, (long)a.M * W * 2);
  const __amdgpu_buffer_rsrc_t rs_dl = make_rsrc(a.dlogits, (long)a.M * Vp * 2);
  const __amdgpu_buffer_rsrc_t rs_dh = make_rsrc(a.dH, (long)a.M * W * 4);
  const __amdgpu_buffer_rsrc_t rs_rs = make_rsrc(a.rowstat, (long)a.M * 8);
  // a tile's rows: 64 x 256 bytes = one 16-byte piece per thread (rows beyond M read as zeros)
  const int x_row = tid >> 4, x_seg = tid & 15;
  auto fetch = [&](int i) __attribute__((always_inline)) {
    const long row = (long)(wg + (long)i * a.n_wg) * ROWS + x_row;
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, x_seg * 16, (int)(unsigned)(row * W * 2), 0));
  };
  auto put = [&](int buf, u32x4 v) __attribute__((always_inline)) {
    *reinterpret_cast<u32x4*>(smem + CE_XT + (buf * ROWS + x_row) * CE_X_LD + x_seg * 16) = v;
  };
  const int rs4 = lane >> 4, cl = lane & 15;       // softmax pass: row 4 wave + rs4, characters 64 k + 4 cl ..
  u32x4 ra = fetch(0);
  put(0, ra);
  if (my_tiles > 1) ra = fetch(1);
#pragma unroll 1
  for (int i = 0; i < my_tiles; ++i) {
    const int buf = i & 1;
    const long row0 = (long)(wg + (long)i * a.n_wg) * ROWS;
    __syncthreads();                       // tile i is complete in LDS; every wave has left tile i - 1, its logits and its gradient rows
    if (i + 1 < my_tiles) put(buf ^ 1, ra);
    if (i + 2 < my_tiles) ra = fetch(i + 2);
    // (the target of this lane's row, asked for before the contraction)
    const int lr = 4 * wave + rs4;
    const long row = row0 + lr;
    const bool there = row < a.M;
    const long rr = there ? row : 0;
    const int tt = (int)(rr / a.B);
    int t = a.tgt[(rr - (long)tt * a.B) * a.T + tt];
    // ---- logits of 64 rows x this wave's 16 characters
    {
      f32x4 acc[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) acc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned char* tb = smem + CE_XT + (buf * ROWS + (lane & 15)) * CE_X_LD + (lane >> 4) * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int h = 0; h < 4; ++h)
          acc[h] = mfma16(__builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(tb + h * 16 * CE_X_LD + j * 64)), __builtin_bit_cast(bf16x8, bu[j]), acc[h]);
      }
#pragma unroll
      for (int h = 0; h < 4; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) zl[(h * 16 + 4 * (lane >> 4) + r) * CE_Z_LD + 16 * wave + (lane & 15)] = acc[h][r];
    }
    __syncthreads();
    // ---- softmax, cross-entropy and its gradient of rows 4 wave .. 4 wave + 3
    {
      float e[16];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 z = *reinterpret_cast<const f32x4*>(zl + lr * CE_Z_LD + 64 * k + 4 * cl);
#pragma unroll
        for (int j = 0; j < 4; ++j) e[4 * k + j] = 64 * k + 4 * cl + j < V ? z[j] : -INFINITY;      // (characters beyond the vocabulary do not exist)
      }
      float mloc = e[0];
#pragma unroll
      for (int j = 1; j < 16; ++j) mloc = fmaxf(mloc, e[j]);
      const float mx = row16_max(mloc);
      // the first character that reaches the maximum (Keras' argmax)
      int first = 0x7fffffff;
#pragma unroll
      for (int j = 15; j >= 0; --j) first = e[j] == mx ? 64 * (j >> 2) + 4 * cl + (j & 3) : first;
      const int amax = row16_min_i(first);      // (a lane's characters ascend with j: `first` is its smallest hit)
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        e[j] = __expf(e[j] - mx);      // (exp2-based: the arguments are <= 0)
        sum += e[j];
      }
      const float inv = 1.f / row16_sum(sum);
      bool counts = true;
      if (a.last_only && tt != a.T - 1) { t = -1; counts = false; }
      if (t < -1) { t = -1; counts = false; }      // (a dummy stream added by the caller's padding: no accuracy either)
      float ploc = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        e[j] *= inv;
        ploc = (64 * (j >> 2) + 4 * cl + (j & 3)) == t ? e[j] : ploc;
      }
      const float pt = row16_sum(ploc);              // (one lane of the row holds the target's probability)
      const bool valid = t >= 0;
      const bool active = there && valid && pt >= 1e-7f && pt <= 1.f - 1e-7f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        unsigned short g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float gk = active ? e[4 * k + j] : 0.f;
          if (active && 64 * k + 4 * cl + j == t) gk -= 1.f;
          g[j] = f2bf(gk * a.inv_count);
        }
        const u32x2 gp = u32x2{(unsigned)g[0] | ((unsigned)g[1] << 16), (unsigned)g[2] | ((unsigned)g[3] << 16)};
        *reinterpret_cast<u32x2*>(smem + CE_GL + lr * CE_G_LD + (64 * k + 4 * cl) * 2) = gp;
        // (rows beyond M: out of the buffer's range, dropped; the row differs from lane to lane: all of the address in the vector offset)
        if (64 * k + 4 * cl < Vp) __builtin_amdgcn_raw_buffer_store_b64(gp, rs_dl, (int)(unsigned)(row * Vp * 2 + (64 * k + 4 * cl) * 2), 0, 0);
      }
      if (cl == 0) {
        float l = 0.f;
        if (valid) {
          const float pc = fminf(fmaxf(pt, 1e-7f), 1.f - 1e-7f);
          l = -__logf(pc) * a.inv_count;
        }
        const int tsafe = valid ? t : 0;
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, l), __builtin_bit_cast(unsigned, (counts && amax == tsafe) ? a.inv_count : 0.f)},
                                              rs_rs, (int)(unsigned)(row * 8), 0, 0);
      }
    }
    __syncthreads();
    // ---- dH of the tile: wave = (16 units of column tile wave & 7, row tiles 2 (wave >> 3) and the next), K = Vp characters
    {
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
      const int h0 = 2 * (wave >> 3);
      const unsigned char* gb = smem + CE_GL + (h0 * 16 + (lane & 15)) * CE_G_LD + (lane >> 4) * 16;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < nk) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
            acc[h] = mfma16(__builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(gb + h * 16 * CE_G_LD + j * 64)), __builtin_bit_cast(bf16x8, be[j]), acc[h]);
        }
      // (out as whole rows: through the logits' LDS, free since the barrier above)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) zl[((h0 + h) * 16 + 4 * (lane >> 4) + r) * CE_D_LD + 16 * (wave & 7) + (lane & 15)] = acc[h][r];
    }
    __syncthreads();
    {
      const int d_row = tid >> 4, d_seg = tid & 15;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(zl + d_row * CE_D_LD + (d_seg + 16 * k) * 4);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs_dh, (int)(unsigned)(((row0 + d_row) * W + (d_seg + 16 * k) * 4) * 4), 0, 0);
      }
    }
  }
}

}  // namespace

namespace {
int w128_cus() {
  static int v = 0;
  if (!v) {
    int dev = 0;
    hipDeviceProp_t prop;
    v = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 1;
  }
  return v;
}
}  // namespace

// the gate inputs of layer 0 straight from the look-up tables: up to two context variables (kl_scan_w128_tables_max_ctx)
namespace {
bool w128_tables_ok(const KlScanFwdWide& a) {
  if (!a.EK || !a.idx || !a.bias || a.n_ctx < 0 || a.n_ctx > 2 || (a.n_ctx > 0 && !a.ctx)) return false;
  for (int n = 0; n < a.n_ctx; ++n)
    if (!a.CtxK[n]) return false;
  return true;
}
}  // namespace
int kl_scan_w128_tables_max_ctx() { return 2; }

bool kl_scan_w128_applicable(int B, int T, int W) {
  return W == W8 && B >= 1 && T >= 1 && (long)(T + 1) * B * 4 * W * 4 < 0x7fffffffffffL;
}

// KL_ERR_SHAPE = not applicable.  Gate inputs: f32 rows P [T*B][4W] (gate-major, bias included); or -- a.P == null, layer 0 -- the
// look-up tables a.EK / a.CtxK[0 .. n_ctx) (f32 rows of 4W) with a.idx [B][T] / a.ctx [B][T][n_ctx] and a.bias; or -- a.KT != null -- the input
// rows a.X [T*B][W] bf16 with the input kernel a.KT [4W][W] and a.bias [4W] (the contraction then happens inside the scan).
int kl_launch_scan_fwd_w128(KlScanFwdWide a, hipStream_t stream) {
  if (!kl_scan_w128_applicable(a.B, a.T, a.W) || a.p_bf16 || a.HT || a.HdT || !a.H || !a.C || !a.UT) return KL_ERR_SHAPE;
  const bool kin = a.KT != nullptr;
  const int mode = kin ? 1 : (a.P ? 0 : 2 + a.n_ctx);
  if (kin ? (!a.X || !a.bias) : (!a.P && !w128_tables_ok(a))) return KL_ERR_SHAPE;
  static KlLdsGrant grant[5];
  const void* fn = mode == 0   ? reinterpret_cast<const void*>(&lstm_scan_fwd_w128_kernel<0>)
                   : mode == 1 ? reinterpret_cast<const void*>(&lstm_scan_fwd_w128_kernel<1>)
                   : mode == 2 ? reinterpret_cast<const void*>(&lstm_scan_fwd_w128_kernel<2>)
                   : mode == 3 ? reinterpret_cast<const void*>(&lstm_scan_fwd_w128_kernel<3>)
                               : reinterpret_cast<const void*>(&lstm_scan_fwd_w128_kernel<4>);
  if (kl_grant_lds(grant[mode], fn, (size_t)F_LDS)) return KL_ERR_LAUNCH;
  const dim3 grid((a.B + 15) / 16), block(NT8);
  switch (mode) {
    case 0: hipLaunchKernelGGL(lstm_scan_fwd_w128_kernel<0>, grid, block, (size_t)F_LDS, stream, a); break;
    case 1: hipLaunchKernelGGL(lstm_scan_fwd_w128_kernel<1>, grid, block, (size_t)F_LDS, stream, a); break;
    case 2: hipLaunchKernelGGL(lstm_scan_fwd_w128_kernel<2>, grid, block, (size_t)F_LDS, stream, a); break;
    case 3: hipLaunchKernelGGL(lstm_scan_fwd_w128_kernel<3>, grid, block, (size_t)F_LDS, stream, a); break;
    default: hipLaunchKernelGGL(lstm_scan_fwd_w128_kernel<4>, grid, block, (size_t)F_LDS, stream, a); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

bool kl_scan_w128_multi_fits(int B, int L) { return L >= 2 && L <= 4 && B >= 1 && (long)L * ((B + 15) / 16) <= w128_cus(); }

// Backward, all layers in one launch: a.L layers in the slots 0 .. L - 1 (Un, Kn, G, C, dZ, mask, db_l), a.dH = the gradient from
// the softmax side (f32, for the top layer); the CALLER pre-fills dZ[1 .. L - 1] with 0xFFFF halfwords (all steps; a.sentinel == 2:
// only steps T - 1 and T - 2, the publishing layer arms step t - 2 while it publishes step t).  KL_ERR_SHAPE = not applicable.
int kl_launch_scan_bwd_w128_multi(KlScanBwd a, hipStream_t stream) {
  if (!kl_scan_w128_applicable(a.B, a.T, a.W) || !kl_scan_w128_multi_fits(a.B, a.L) || a.dZT || !a.dH || !a.status) return KL_ERR_SHAPE;
  for (int l = 0; l < a.L; ++l)
    if (!a.Un[l] || !a.G[l] || !a.C[l] || !a.dZ[l] || (l > 0 && !a.Kn[l])) return KL_ERR_SHAPE;
  a.n_rb = (a.B + 15) / 16;
  static KlLdsGrant grant;
  if (kl_grant_lds(grant, reinterpret_cast<const void*>(&lstm_scan_bwd_w128_multi_kernel), (size_t)B_LDS_XIN)) return KL_ERR_LAUNCH;
  hipLaunchKernelGGL(lstm_scan_bwd_w128_multi_kernel, dim3(a.L * a.n_rb), dim3(NT8), (size_t)B_LDS_XIN, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

// The layers of a window in one launch (layers[0]: gate inputs P; layers[l > 0]: KT / X / bias with X = the rows layer l - 1
// publishes -- the CALLER pre-fills those rows with 0xFFFF halfwords: all T steps, or with layers[l - 1].sentinel == 2 only the
// first two, the publishing layer then arms step t + 2 while it publishes step t).  Only where every workgroup finds a CU at once
// (layers x row blocks <= CUs): the point is to use CUs a single layer leaves idle.  KL_ERR_SHAPE = not applicable.
int kl_launch_scan_fwd_w128_multi(const KlScanFwdWide* layers, int L, hipStream_t stream) {
  if (!kl_scan_w128_multi_fits(layers[0].B, L)) return KL_ERR_SHAPE;
  const int n_rb = (layers[0].B + 15) / 16;
  for (int l = 0; l < L; ++l) {
    const KlScanFwdWide& a = layers[l];
    if (!kl_scan_w128_applicable(a.B, a.T, a.W) || a.B != layers[0].B || a.T != layers[0].T || a.p_bf16 || a.HT || a.HdT || !a.H || !a.C || !a.UT || !a.status)
      return KL_ERR_SHAPE;
    if (l == 0 ? (a.KT || (!a.P && !w128_tables_ok(a))) : (!a.KT || !a.X || !a.bias)) return KL_ERR_SHAPE;
    if (l > 0 && a.X != (layers[l - 1].Hd ? layers[l - 1].Hd : layers[l - 1].H + (size_t)a.B * a.W)) return KL_ERR_SHAPE;
  }
  static KlLdsGrant grant;
  if (kl_grant_lds(grant, reinterpret_cast<const void*>(&lstm_scan_fwd_w128_multi_kernel), (size_t)F_LDS)) return KL_ERR_LAUNCH;
  const KlScanFwdWide& z = layers[L - 1];
  hipLaunchKernelGGL(lstm_scan_fwd_w128_multi_kernel, dim3(L * n_rb), dim3(NT8), (size_t)F_LDS, stream, layers[0], layers[1], L > 2 ? layers[2] : z,
                     L > 3 ? layers[3] : z, L, n_rb);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

// one layer (a.L == 1; Un[0], G[0], C[0], dZ[0], mask[0], db); the gradient from above: dH f32 [T*B][W], or -- a.Kn[1] != null --
// the layer above's dZ rows a.dZ[1] [T*B][4W] with its input kernel a.Kn[1] [W][4W] (contracted inside the scan)
int kl_launch_scan_bwd_w128(KlScanBwd a, hipStream_t stream) {
  if (!kl_scan_w128_applicable(a.B, a.T, a.W) || a.L != 1 || a.dZT || !a.Un[0] || !a.G[0] || !a.C[0] || !a.dZ[0]) return KL_ERR_SHAPE;
  const bool xin = a.Kn[1] != nullptr;
  if (xin ? !a.dZ[1] : !a.dH) return KL_ERR_SHAPE;
  static KlLdsGrant grant[2];
  const void* fn = xin ? reinterpret_cast<const void*>(&lstm_scan_bwd_w128_kernel<true>) : reinterpret_cast<const void*>(&lstm_scan_bwd_w128_kernel<false>);
  const size_t lds = xin ? (size_t)B_LDS_XIN : (size_t)B_LDS;
  if (kl_grant_lds(grant[xin], fn, lds)) return KL_ERR_LAUNCH;
  if (xin) hipLaunchKernelGGL(lstm_scan_bwd_w128_kernel<true>, dim3((a.B + 15) / 16), dim3(NT8), lds, stream, a);
  else hipLaunchKernelGGL(lstm_scan_bwd_w128_kernel<false>, dim3((a.B + 15) / 16), dim3(NT8), lds, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

// Output layer of a training window at width 128 in one pass (logits_ce_w128_kernel): dlogits [M][Vp] bf16, dH [M][128] f32,
// rowstat [M][2]; E [Vp][128] / ET [128][Vp] bf16 with zeros beyond V; the caller still reduces rowstat (kl_launch_rowstat_reduce).
// KL_ERR_SHAPE = not applicable.
int kl_launch_logits_ce_w128(const bf16_t* X, const bf16_t* E, const bf16_t* ET, const int* tgt, bf16_t* dlogits, float* dH, float* rowstat, int B,
                             int T, int W, int V, int Vp, float inv_count, int last_only, hipStream_t stream) {
  const long M = (long)B * T;
  if (W != W8 || V < 1 || V > Vp || Vp > 256 || (Vp & 31) || M < 1 || M * 256L * 2 > 0xfffffff0L) return KL_ERR_SHAPE;
  KlCeW128 a;
  a.X = X; a.E = E; a.ET = ET; a.tgt = tgt; a.dlogits = dlogits; a.dH = dH; a.rowstat = rowstat;
  a.M = (int)M; a.B = B; a.T = T; a.V = V; a.Vp = Vp; a.last_only = last_only; a.inv_count = inv_count;
  const long n_tiles = (M + CE_ROWS - 1) / CE_ROWS;
  a.n_wg = (int)(n_tiles < w128_cus() ? n_tiles : w128_cus());
  static KlLdsGrant grant;
  if (kl_grant_lds(grant, reinterpret_cast<const void*>(&logits_ce_w128_kernel), (size_t)CE_LDS)) return KL_ERR_LAUNCH;
  hipLaunchKernelGGL(logits_ce_w128_kernel, dim3(a.n_wg), dim3(1024), (size_t)CE_LDS, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}
