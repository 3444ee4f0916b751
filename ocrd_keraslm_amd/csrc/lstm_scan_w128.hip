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
template <bool KIN>
__global__ __launch_bounds__(NT8, 1) void lstm_scan_fwd_w128_kernel(const KlScanFwdWide a) {
  constexpr int W = W8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int B = a.B, T = a.T;
  const int row0 = blockIdx.x * 16;
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
  float bias_c[4] = {0.f, 0.f, 0.f, 0.f};      // (KIN: the bias of this lane's column in each tile -- the accumulators start from it)
  if (KIN) {
    const int col = lane & 15;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const long wrow = ((long)(col & 3) * W + 16 * wave + 4 * (col >> 2) + c) * W + (lane >> 4) * 8;
#pragma unroll
      for (int j = 0; j < 4; ++j) bk[c][j] = *reinterpret_cast<const u32x4*>(a.KT + wrow + j * 32);
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
    const int i = tid + NT8 * k, row = i >> 7, seg = i & 127;
    *reinterpret_cast<float4*>(smem + F_PB + (buf * 16 + row) * F_P_LD + seg * 16) = v;
  };
  // (KIN: the input rows X[t] -- 16 x 256 bytes, threads 0 .. 255 -- take the place of the gate-input rows; tile [2][16][272] at F_PB)
  auto x_load = [&](int t) __attribute__((always_inline)) {
    uint4 v = uint4{0u, 0u, 0u, 0u};
    if (tid < 256 && h_row < nrow) v = *reinterpret_cast<const uint4*>(a.X + ((long)t * B + row0 + h_row) * W + h_seg * 8);
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
  if (KIN) {
    x_put(0, x_load(0));
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
    uint4 xn = uint4{0u, 0u, 0u, 0u};
    if (t + 1 < T) {
      if (KIN) {
        xn = x_load(t + 1);
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
        x_put(p ^ 1, xn);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) p_put(p ^ 1, k, pn[k]);
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
      if (tid < 256)
        *reinterpret_cast<uint4*>(a.H + (trow + B + h_row) * W + h_seg * 8) = *reinterpret_cast<const uint4*>(smem + F_HT + ((p ^ 1) * 16 + h_row) * F_H_LD + h_seg * 16);
      else if (a.Hd)
        *reinterpret_cast<uint4*>(a.Hd + (trow + h_row) * W + h_seg * 8) = *reinterpret_cast<const uint4*>(smem + F_HS + (p * 16 + h_row) * F_H_LD + h_seg * 16);
    }
  }
}

// dh[t] = dH[t] (from above, all steps at once: the softmax side or the layer above's dX product) * mask + dZ[t+1] . U^T;
// dZ[t] from the gate derivatives; db summed on the way
// XIN: the gradient from above is contracted HERE -- dh[t] = mask * (dZ_above[t] . K_above^T) + dZ[t+1] . U^T with K_above
// resident beside U and the 16 x 512 rows of the layer above's dZ[t] staged like this layer's own tile -- instead of f32 dH rows
// from a product over all steps (a.Kn[1] / a.dZ[1] = the layer above's input kernel [W][4W] and dZ [T*B][4W]).
template <bool XIN>
__global__ __launch_bounds__(NT8, 1) void lstm_scan_bwd_w128_kernel(const KlScanBwd a) {
  constexpr int W = W8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int B = a.B, T = a.T;
  const int row0 = blockIdx.x * 16;
  const int nrow = min(16, B - row0);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int col = lane & 15, q4 = lane >> 4;
  const int u = 16 * wave + col;                      // this lane's unit; its rows: 4 q4 + r

  // resident weights: B fragments of this wave's 16 output units, K = 4W = 512 (U natural layout [W][4W]: row = unit)
  u32x4 bu[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) bu[j] = *reinterpret_cast<const u32x4*>(a.Un[0] + (long)u * 4 * W + j * 32 + q4 * 8);
  u32x4 bk[XIN ? 16 : 1];
  if (XIN) {
#pragma unroll
    for (int j = 0; j < 16; ++j) bk[j] = *reinterpret_cast<const u32x4*>(a.Kn[1] + (long)u * 4 * W + j * 32 + q4 * 8);
  }
  float ccur[4], dc[4], mk[4], dbacc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = min(4 * q4 + r, nrow - 1);
    ccur[r] = a.C[0][((long)T * B + row0 + row) * W + u];       // c_{T-1} = block T
    mk[r] = a.mask[0] ? a.mask[0][((long)row0 + row) * W + u] : 1.f;
    dc[r] = 0.f;
    dbacc[r] = 0.f;
  }
  // rows between memory and LDS: gates (2 pieces of 16 bytes per thread and step), c_{t-1} (1), dH (1), dZ out (2)
  const int c_row = tid >> 5, c_seg = tid & 31;
  struct In { uint4 g[2]; float4 c, d; uint4 za[2]; };
  auto in_load = [&](int t) __attribute__((always_inline)) {
    In v;
    const long trow = (long)t * B + row0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + NT8 * k, row = i >> 6, seg = i & 63;
      v.g[k] = uint4{0u, 0u, 0u, 0u};
      v.za[k] = uint4{0u, 0u, 0u, 0u};
      if (row < nrow) {
        v.g[k] = *reinterpret_cast<const uint4*>(a.G[0] + (trow + row) * 4 * W + seg * 8);
        if (XIN) v.za[k] = *reinterpret_cast<const uint4*>(a.dZ[1] + (trow + row) * 4 * W + seg * 8);
      }
    }
    v.c = float4{0.f, 0.f, 0.f, 0.f};
    v.d = float4{0.f, 0.f, 0.f, 0.f};
    if (c_row < nrow) {
      v.c = *reinterpret_cast<const float4*>(a.C[0] + (trow + c_row) * W + c_seg * 4);      // block t = c_{t-1}
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
  in_put((T - 1) & 1, in_load(T - 1));
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
    if (t > 0) in_put(p ^ 1, nxt);
    __syncthreads();
    // ---- dZ[t] to memory, whole rows
    const long trow = (long)t * B + row0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int i = tid + NT8 * k, row = i >> 6, seg = i & 63;
      if (row < nrow)
        *reinterpret_cast<uint4*>(a.dZ[0] + (trow + row) * 4 * W + seg * 8) = *reinterpret_cast<const uint4*>(smem + B_ZT + (p * 16 + row) * B_Z_LD + seg * 16);
    }
  }
  // bias gradient: column sums of what was written (the four row groups of a wave, then one atomic per column and workgroup)
  if (a.db) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v = dbacc[g];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (q4 == 0) atomicAdd(a.db + g * W + u, v);
    }
  }
}

}  // namespace

bool kl_scan_w128_applicable(int B, int T, int W) {
  return W == W8 && B >= 1 && T >= 1 && (long)(T + 1) * B * 4 * W * 4 < 0x7fffffffffffL;
}

// KL_ERR_SHAPE = not applicable.  Gate inputs: f32 rows P [T*B][4W] (gate-major, bias included), or -- a.KT != null -- the input
// rows a.X [T*B][W] bf16 with the input kernel a.KT [4W][W] and a.bias [4W] (the contraction then happens inside the scan).
int kl_launch_scan_fwd_w128(KlScanFwdWide a, hipStream_t stream) {
  if (!kl_scan_w128_applicable(a.B, a.T, a.W) || a.p_bf16 || a.HT || a.HdT || !a.H || !a.C || !a.UT) return KL_ERR_SHAPE;
  const bool kin = a.KT != nullptr;
  if (kin ? (!a.X || !a.bias) : !a.P) return KL_ERR_SHAPE;
  static KlLdsGrant grant[2];
  const void* fn = kin ? reinterpret_cast<const void*>(&lstm_scan_fwd_w128_kernel<true>) : reinterpret_cast<const void*>(&lstm_scan_fwd_w128_kernel<false>);
  if (kl_grant_lds(grant[kin], fn, (size_t)F_LDS)) return KL_ERR_LAUNCH;
  if (kin) hipLaunchKernelGGL(lstm_scan_fwd_w128_kernel<true>, dim3((a.B + 15) / 16), dim3(NT8), (size_t)F_LDS, stream, a);
  else hipLaunchKernelGGL(lstm_scan_fwd_w128_kernel<false>, dim3((a.B + 15) / 16), dim3(NT8), (size_t)F_LDS, stream, a);
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
