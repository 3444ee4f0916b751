// Incremental step for many hypotheses (n >= 256; Rater.predict / rate_best batches, rating.py:578-639): ONE launch per
// layer, a workgroup = TR hypotheses x 32 hidden units x 4 gates, the contraction in split precision over operands
// that are each read ONCE per tile.
//
// What the [hi | lo | hi] . [w_hi | w_hi | w_lo] form of step_big.hip + gemm.hip cost (round 2-3: 2 x 23 us of a 60 us
// step at 1024 hypotheses): 6 bytes per element on both sides (hi twice), all of it by LDS-DMA (1 KiB per ~32 cycles of
// the CU's address unit: ~70 GB/s per CU whatever is in flight), behind a separate gather launch that splits the f32 state
// rows, in front of an epilogue whose inputs were requested only after the last k-step.  Here:
//  * the state rows are read as they are (f32, through the pool slots: slot_out for the layer below's new h, slot_in for
//    this layer's previous h), split into bf16 hi + lo in registers and laid into the stage's two A planes -- no gather
//    launch, no activation copy in HBM, 4 bytes per element;
//  * the weights come from the [4W][W] hi / lo arrays kl_prepare keeps anyway (the tile's 128 columns are four row groups
//    of 32, one per gate: no permuted copy), 4 bytes per element, as 16-byte register loads (half the address-unit time
//    of LDS-DMA) two k-steps ahead;
//  * three MFMAs per fragment pair: hi.hi + lo.hi + hi.lo;
//  * the epilogue's inputs (c_prev, the table rows of layer 0, the bias) are requested before the main loop;
//  * tiles are dealt to the XCDs as rectangles of the tile grid (4 unit blocks x 8 row tiles at 1024 x 512), so that an
//    XCD's L2 holds what its 32 tiles share (speed only: any placement is correct).
// The arithmetic restates rating.py:578-639 (one LSTM step per layer with explicit states); rows = hypotheses.
#include <string.h>

#include <type_traits>

#include "kl_common.h"
#include "kl_kernels.h"

namespace {

// byte offset of 16-byte chunk `chunk` (0..7) of row `row` in a [rows][64] bf16 plane (the same image as gemm.hip's)
__device__ __forceinline__ int pl_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

struct IncTile {
  int n, W;
  float* pool; long slot_ld;
  const int* slot_in; const int* slot_out;
  int h_off, c_off, x_off;          // float offsets inside a slot: this layer's h and c, the layer below's h (-1: layer 0)
  const bf16_t* UT_hi; const bf16_t* UT_lo; const bf16_t* KT_hi; const bf16_t* KT_lo;      // [4W][W]
  const float* T1; const int* i1; const float* T2; const int* i2; const float* bias;       // z init (tables [.][4W], bias [4W])
  int nx, ny, px, py;               // tile grid (unit blocks x row tiles) and its partition over the XCDs (px * py == 8, or 0)
};

constexpr int TC = 128;             // tile columns: 4 gates x 32 units
constexpr int BK = 64;              // k-step

// LDS stores done and visible, then the barrier; nothing moves across it at compile time either.  (Not __syncthreads():
// its fences may drain the register loads that are meant to stay in flight across the barrier.)
__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// The split of a state value into bf16 hi + lo, on single registers: written with plain float arithmetic hipcc pairs the
// subtractions of two pieces into packed instructions, whose operands must be even-aligned register pairs -- it then
// shuffles the freshly loaded registers into pairs right behind the loads, i.e. waits for loads that are meant to
// stay in flight for two k-steps.
__device__ __forceinline__ unsigned cvt_pk(float lo_half, float hi_half) {      // two f32 -> two bf16 (RNE) in one register
  unsigned d;
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(lo_half), "v"(hi_half));      // (volatile: stays behind the barrier in front of it -- hoisted, it would wait for its load a contraction early)
  return d;
}
__device__ __forceinline__ float sub_f32(float x, float y) {
  float d;
  asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y));
  return d;
}

// VAR: timing builds (KL_TILE_VAR; 0 in production): 1 = no epilogue, 2 = no main loop
// Gate non-linearities from the hardware's exp2 and reciprocal (1 ulp each; absolute error of a gate ~2e-7): the ocml
// expf / tanhf / IEEE division of kl_common.h cost ~250 instructions per cell, 4 us of a 15 us launch at 1024 x 512.
__device__ __forceinline__ float gate_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float gate_tanh(float x) {
  return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.f;
}

template <int TR, bool LO, int VAR>
__global__ __launch_bounds__(512, 1) void inc_tile_kernel(const IncTile a) {
  constexpr int WM = 2, WN = 4;                       // 8 waves: wave = (TR / 2) rows x 32 columns
  constexpr int RF = TR / 16 / WM, NT = 2;            // 16 x 16 fragments per wave
  constexpr int APC = TR * 16 / 512;                  // float4 pieces of A per thread and k-step
  constexpr int BPC = LO ? 4 : 2;                     // 16-byte pieces of B per thread and k-step
  constexpr int NPL = LO ? 2 : 1;
  constexpr int A_PLANE = TR * 128, B_PLANE = TC * 128;
  constexpr int STAGE = NPL * (A_PLANE + B_PLANE);
  static_assert(TR == 64, "the epilogue's thread = (row, four units) map takes 64 rows");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int W = a.W;

  // ---- tile of this workgroup
  int bx, by;
  {
    const int lin = blockIdx.x;
    if (a.px) {
      const int xcd = lin & 7, j = lin >> 3;
      const int sx = a.nx / a.px, sy = a.ny / a.py;
      bx = (xcd % a.px) * sx + j % sx;
      by = (xcd / a.px) * sy + j / sx;
    } else {
      bx = lin % a.nx;
      by = lin / a.nx;
    }
  }
  const int m0 = by * TR, u0 = bx * 32;
  const int Kx = a.x_off >= 0 ? W : 0;                // [x | h] or h alone
  const int nkt = (Kx + W) / BK, nkx = Kx / BK;

  // ---- every index this thread will need, in one round trip: the slots of its A rows (piece j: row j * 32 + tid / 16, floats
  // 4 * (tid % 16) .. + 3 of the k-step) and of its epilogue cells (row tid / 8, units u0 + 4 * (tid % 8) .. + 3: 16-byte loads and stores -- a
  // dword access costs the CU's address unit as much as a 16-byte one, and there are 13 loads per thread this way, not 44)
  int a_si[APC], a_so[APC];
#pragma unroll
  for (int j = 0; j < APC; ++j) {
    const int row = min(m0 + j * 32 + (tid >> 4), a.n - 1);
    a_si[j] = a.slot_in[row];
    a_so[j] = a.slot_out[row];
  }
  const int erow = min(m0 + (tid >> 3), a.n - 1);
  const int e_si = a.slot_in[erow], e_out = a.slot_out[erow];
  const int e_i1 = a.i1 ? a.i1[erow] : erow, e_i2 = a.i2 ? a.i2[erow] : erow;
  // B piece j: plane j / 2, column (j % 2) * 64 + tid / 8 = gate * 32 + unit, chunk tid % 8
  unsigned bo[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = j * 64 + (tid >> 3);
    bo[j] = (unsigned)(((col >> 5) * W + u0 + (col & 31)) * W + (tid & 7) * 8);
  }
  struct Regs {
    float4 av[APC];
    uint4 bv[BPC];
  };
  auto load_b = [&](Regs& r, int kt) __attribute__((always_inline)) {
    const bool is_x = kt < nkx;
    const bf16_t* wh = is_x ? a.KT_hi : a.UT_hi;
    const bf16_t* wl = is_x ? a.KT_lo : a.UT_lo;
    const int kw = is_x ? kt * BK : kt * BK - Kx;
#pragma unroll
    for (int j = 0; j < 2; ++j) r.bv[j] = *reinterpret_cast<const uint4*>(wh + bo[j] + kw);
    if (LO) {
#pragma unroll
      for (int j = 0; j < 2; ++j) r.bv[2 + j] = *reinterpret_cast<const uint4*>(wl + bo[j] + kw);
    }
  };
  const float* ax[APC];
  const float* ah[APC];
  auto load_a = [&](Regs& r, int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < APC; ++j) r.av[j] = *reinterpret_cast<const float4*>((kt < nkx ? ax[j] : ah[j]) + kt * BK);
  };
  auto load = [&](Regs& r, int kt) __attribute__((always_inline)) {
    load_b(r, kt);
    load_a(r, kt);
  };
  auto stage_write = [&](const Regs& r, int stage) __attribute__((always_inline)) {
    unsigned char* const st = smem + stage * STAGE;
#pragma unroll
    for (int j = 0; j < APC; ++j) {
      const float4 v = r.av[j];
      const unsigned h01 = cvt_pk(v.x, v.y), h23 = cvt_pk(v.z, v.w);
      const int off = pl_off(j * 32 + (tid >> 4), (tid & 15) >> 1) + (tid & 1) * 8;
      *reinterpret_cast<uint2*>(st + off) = uint2{h01, h23};
      if (LO) {
        const unsigned l01 = cvt_pk(sub_f32(v.x, __builtin_bit_cast(float, h01 << 16)), sub_f32(v.y, __builtin_bit_cast(float, h01 & 0xffff0000u)));
        const unsigned l23 = cvt_pk(sub_f32(v.z, __builtin_bit_cast(float, h23 << 16)), sub_f32(v.w, __builtin_bit_cast(float, h23 & 0xffff0000u)));
        *reinterpret_cast<uint2*>(st + A_PLANE + off) = uint2{l01, l23};
      }
    }
#pragma unroll
    for (int j = 0; j < BPC; ++j) {
      const int col = (j & 1) * 64 + (tid >> 3);
      *reinterpret_cast<uint4*>(st + NPL * A_PLANE + (j >> 1) * B_PLANE + pl_off(col, tid & 7)) = r.bv[j];
    }
  };

  // ---- the weights of the first two k-steps need no index: they go out behind the index loads; then ONE wait for the
  // indices, and everything that hangs on them -- the state rows of the two k-steps, the epilogue's inputs -- goes out
  Regs r0, r1;
  constexpr bool run_main = VAR != 2;
  if (run_main) {
    load_b(r0, 0);
    load_b(r1, 1);
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * BPC) : "memory");
#pragma unroll
  for (int j = 0; j < APC; ++j) {
    ax[j] = a.pool + (long)a_so[j] * a.slot_ld + (a.x_off >= 0 ? a.x_off : 0) + (tid & 15) * 4;
    ah[j] = a.pool + (long)a_si[j] * a.slot_ld + a.h_off + (tid & 15) * 4 - Kx;
  }
  if (run_main) {
    load_a(r0, 0);
    load_a(r1, 1);
  }

  // ---- epilogue inputs: requested now, used after the loop (whole groups under uniform branches: no wait in between)
  const int eu = u0 + 4 * (tid & 7);
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 ecp = *reinterpret_cast<const f32x4*>(a.pool + (long)e_si * a.slot_ld + a.c_off + eu);
  f32x4 et1[4], et2[4], eb[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) eb[g] = et1[g] = et2[g] = zero4;
  if (a.bias) {
#pragma unroll
    for (int g = 0; g < 4; ++g) eb[g] = *reinterpret_cast<const f32x4*>(a.bias + g * W + eu);
  }
  if (a.T1) {
#pragma unroll
    for (int g = 0; g < 4; ++g) et1[g] = *reinterpret_cast<const f32x4*>(a.T1 + (long)e_i1 * 4 * W + g * W + eu);
  }
  if (a.T2) {
#pragma unroll
    for (int g = 0; g < 4; ++g) et2[g] = *reinterpret_cast<const f32x4*>(a.T2 + (long)e_i2 * 4 * W + g * W + eu);
  }

  f32x4 acc[RF][NT];
#pragma unroll
  for (int i = 0; i < RF; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  auto contract = [&](int stage) __attribute__((always_inline)) {
    const unsigned char* const st = smem + stage * STAGE;
    const unsigned char* const a_hi = st;
    const unsigned char* const b_hi = st + NPL * A_PLANE;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      frag16 fah[RF], fal[RF], fbh[NT], fbl[NT];
#pragma unroll
      for (int i = 0; i < RF; ++i) {
        const int off = pl_off(wm * (TR / WM) + i * 16 + fr, s * 4 + fq);
        fah[i].u = *reinterpret_cast<const uint4*>(a_hi + off);
        if (LO) fal[i].u = *reinterpret_cast<const uint4*>(a_hi + A_PLANE + off);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int off = pl_off(wn * 32 + j * 16 + fr, s * 4 + fq);
        fbh[j].u = *reinterpret_cast<const uint4*>(b_hi + off);
        if (LO) fbl[j].u = *reinterpret_cast<const uint4*>(b_hi + B_PLANE + off);
      }
#pragma unroll
      for (int i = 0; i < RF; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          acc[i][j] = mfma16(fah[i].v, fbh[j].v, acc[i][j]);
          if (LO) {
            acc[i][j] = mfma16(fal[i].v, fbh[j].v, acc[i][j]);
            acc[i][j] = mfma16(fah[i].v, fbl[j].v, acc[i][j]);
          }
        }
    }
  };

  // ---- main loop, two k-steps per turn: registers -> stage, the k-step two ahead requested into the registers just
  // emptied, one barrier, MFMAs.  A stage is rewritten two k-steps later: every wave has passed the barrier in between,
  // which it reaches with that stage's fragment reads retired.  (The last turn is peeled so that the compiler counts its
  // waits over a fixed sequence of loads, the first one so that the epilogue's inputs stay in flight behind it; nkt >= 4.)
  if (run_main) {
    auto turn = [&](int kt, auto more) __attribute__((always_inline)) {
      constexpr bool MORE = decltype(more)::value;
      stage_write(r0, 0);
      if (MORE) load(r0, kt + 2);
      wg_barrier();
      contract(0);
      stage_write(r1, 1);
      if (MORE) load(r1, kt + 3);
      wg_barrier();
      contract(1);
    };
    turn(0, std::true_type{});
    int kt = 2;
    for (; kt + 2 < nkt; kt += 2) turn(kt, std::true_type{});
    turn(kt, std::false_type{});
  }
  if (VAR == 1) {
    if (acc[0][0][0] == 12345.678f) a.pool[0] = 0.f;
    return;
  }

  // ---- cell update: the tile goes through LDS once, so that a thread holds the four gates of its cells
  constexpr int LDP = TC + 4;
  float* const ct = reinterpret_cast<float*>(smem);
  wg_barrier();
#pragma unroll
  for (int i = 0; i < RF; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) ct[(wm * (TR / WM) + i * 16 + fq * 4 + r) * LDP + wn * 32 + j * 16 + fr] = acc[i][j][r];
  wg_barrier();
  if (m0 + (tid >> 3) < a.n) {
    f32x4 z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
      z[g] = *reinterpret_cast<const f32x4*>(ct + (tid >> 3) * LDP + g * 32 + 4 * (tid & 7)) + eb[g] + et1[g] + et2[g];
    f32x4 c, hv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gi = gate_sigmoid(z[0][k]), gf = gate_sigmoid(z[1][k]), gg = gate_tanh(z[2][k]), go = gate_sigmoid(z[3][k]);
      c[k] = gf * ecp[k] + gi * gg;
      hv[k] = go * gate_tanh(c[k]);
    }
    float* out = a.pool + (long)e_out * a.slot_ld;
    *reinterpret_cast<f32x4*>(out + a.c_off + eu) = c;
    *reinterpret_cast<f32x4*>(out + a.h_off + eu) = hv;
  }
}

}  // namespace

// one LSTM cell step of a layer for n hypotheses with pool slots; KL_ERR_SHAPE = not applicable (the caller takes
// step_big.hip's gather + GEMM path)
int kl_launch_inc_tile(const KlIncCellArgs& p, int variant, hipStream_t stream) {
  const int W = p.W;
  if (p.n < 1 || (W & 127) || W < 256 || !p.pool || !p.slot_in || !p.slot_out || !p.UT_hi) return KL_ERR_SHAPE;
  if ((long)4 * W * W >= (1L << 31)) return KL_ERR_SHAPE;
  const bool lo = p.split == 3;
  if (lo && (!p.UT_lo || (p.x_off >= 0 && !p.KT_lo))) return KL_ERR_ARG;
  if (p.x_off >= 0 && !p.KT_hi) return KL_ERR_ARG;
  constexpr int TR = 64;
  IncTile a;
  memset(&a, 0, sizeof(a));
  a.n = p.n; a.W = W; a.pool = p.pool; a.slot_ld = p.slot_ld; a.slot_in = p.slot_in; a.slot_out = p.slot_out;
  a.h_off = p.h_off; a.c_off = p.c_off; a.x_off = p.x_off;
  a.UT_hi = p.UT_hi; a.UT_lo = p.UT_lo; a.KT_hi = p.KT_hi; a.KT_lo = p.KT_lo;
  a.T1 = p.T1; a.i1 = p.i1; a.T2 = p.T2; a.i2 = p.i2; a.bias = p.bias;
  a.nx = W / 32;
  a.ny = (p.n + TR - 1) / TR;
  // XCD partition of the tile grid: the split (px unit-block groups x py row-tile groups) with the fewest operand bytes per
  // XCD (a unit block's weights are 128 rows of K, a row tile's states TR rows)
  if (((a.nx * a.ny) & 7) == 0) {
    long best = -1;
    for (int px = 1; px <= 8; px *= 2) {
      const int py = 8 / px;
      if (a.nx % px || a.ny % py) continue;
      const long cost = (long)(a.nx / px) * TC + (long)(a.ny / py) * TR;
      if (best < 0 || cost < best) { best = cost; a.px = px; a.py = py; }
    }
  }
  const size_t lds = (size_t)2 * (lo ? 2 : 1) * (TR + TC) * 128;
  static_assert((size_t)TR * (TC + 4) * 4 <= (size_t)2 * (TR + TC) * 128, "the epilogue's tile fits in the stages");
#define KL_IT_CASE(LO_, VAR_)                                                                                               \
  do {                                                                                                                      \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&inc_tile_kernel<TR, LO_, VAR_>),                                \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return KL_ERR_LAUNCH;     \
    hipLaunchKernelGGL((inc_tile_kernel<TR, LO_, VAR_>), dim3(a.nx * a.ny), dim3(512), lds, stream, a);                     \
  } while (0)
  if (!lo) KL_IT_CASE(false, 0);
  else if (variant == 1) KL_IT_CASE(true, 1);
  else if (variant == 2) KL_IT_CASE(true, 2);
  else KL_IT_CASE(true, 0);
#undef KL_IT_CASE
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}
