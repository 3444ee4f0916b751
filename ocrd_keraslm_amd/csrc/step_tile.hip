// Incremental step for many hypotheses (n >= 256; Rater.predict / rate_best batches, rating.py:578-639): ONE launch per
// layer, a workgroup = 64 hypotheses x 32 hidden units x 4 gates, the contraction in split precision over operands
// that are each read ONCE per tile; and the output layer (logits over the tied embedding + softmax) in one launch.
//
// What the [hi | lo | hi] . [w_hi | w_hi | w_lo] form of step_big.hip + gemm.hip cost (round 2-3: 2 x 23 us of a 60 us
// step at 1024 hypotheses): 6 bytes per element on both sides (hi twice), all of it by LDS-DMA, behind a separate gather
// launch that splits the f32 state rows, in front of an epilogue whose inputs were requested only after the last k-step
// and whose gates cost ~250 instructions per cell.  Here:
//  * the state rows are read as they are (f32, through the pool slots: slot_out for the layer below's new h, slot_in for
//    this layer's previous h), split into bf16 hi + lo in registers and laid into the stage's two planes -- no gather
//    launch, no activation copy in HBM, 4 bytes per element;
//  * the weights never touch LDS: wave w owns columns 16 w .. 16 w + 15 of the tile, and its fragments come straight from
//    FRAGMENT-MAJOR copies of the [4W][W] hi / lo arrays (frag_major_kernel: block (16 rows, 32-deep k block, plane) = 1 KiB
//    in the order the MFMA's lanes take it, so a wave instruction reads 1 KiB of contiguous memory), four k-steps ahead.
//    Read from the [4W][W] arrays themselves, a fragment is 16 rows x 64 bytes with consecutive lanes in different rows:
//    every lane a cache line of its own -- measured 27 GB/s per CU, slower than staging the weights through LDS;
//  * three MFMAs per fragment pair: hi.hi + lo.hi + hi.lo;
//  * the epilogue's inputs (c_prev, the table rows of layer 0, the bias) are requested before the main loop, as 16-byte
//    accesses (a dword access costs the CU's address unit as much), the gates from the hardware's exp2 and reciprocal;
//  * tiles are dealt to the XCDs as rectangles of the tile grid (4 unit blocks x 8 row tiles at 1024 x 512), so that an
//    XCD's L2 holds what its 32 tiles share (speed only: any placement is correct).
// Measured (round 3, depth 2, width 512, split precision): 1024 hypotheses 60.7 -> 40-42 us per step (the two cell launches
// 2 x 23 -> 12.2 + 19.8 us, gather 5.6 -> none, thin GEMM 8.9 + softmax 4.8 -> 9.6 us); 2048: 96 -> 70; depth 4 / width 1024
// at 1024 hypotheses 453 -> 233 us.  The main loop runs at the L2's rate: 256 workgroups x (64 + 128) rows x K x 4 bytes =
// 196 MB per layer-1 launch at ~18 TB/s (64 B/clk per L2 channel, 16 channels per XCD).
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
  const bf16_t* UF; const bf16_t* KF;                 // fragment-major weights (frag_major_kernel): [4W / 16][W / 32][planes][64][8]
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
  // 8 waves, wave w = all TR rows x columns 16 w .. 16 w + 15 of the tile (gate w / 2, units 16 (w % 2) ..): every weight
  // element is used by exactly one wave, so the weight fragments go straight from global memory into MFMA operands (lane
  // (column c, quarter q) holds k = 8 q .. 8 q + 7 of a 32-deep block: 16 bytes of row c of the [4W][W] array) -- no LDS
  // store, no LDS read, no barrier on their way; four k-steps of them are in flight per wave.  Only the state rows, which
  // every wave needs and which have to be split first, go through LDS (two stages of hi + lo planes, 32 KiB).
  constexpr int RF = TR / 16;                         // 16-row fragments
  constexpr int APC = TR * 16 / 512;                  // float4 pieces of A per thread and k-step
  constexpr int NPL = LO ? 2 : 1;
  constexpr int A_PLANE = TR * 128;
  constexpr int STAGE = NPL * A_PLANE;
  constexpr int EP = TR / 64;                         // epilogue passes: thread = (row p * 64 + tid / 8, four units)
  constexpr bool EARLY = TR == 64;                    // the epilogue's inputs requested before the main loop (TR = 128: behind it --
                                                      // 8 row fragments and their accumulators leave no registers to park them in)
  static_assert(TR == 64 || TR == 128, "tile rows");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
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

  // ---- the indices of the tile's 64 rows (pool slots, table rows): requested by ONE wave and handed to the others through
  // LDS.  (Every workgroup of a row tile -- W / 32 of them, eight waves each -- would ask for the same few cache lines at the
  // same moment, and an L2 channel answers the requests for one line one after the other: step_small.hip's stamps showed
  // 2 500 - 6 000 clocks from request to arrival that way.)
  __shared__ int idx_lds[4][TR];
  int i_si = 0, i_so = 0, i_1 = 0, i_2 = 0;
  if (wave < TR / 64) {
    const int row = min(m0 + wave * 64 + lane, a.n - 1);
    i_si = a.slot_in[row];
    i_so = a.slot_out[row];
    i_1 = a.i1 ? a.i1[row] : row;
    i_2 = a.i2 ? a.i2[row] : row;
  }

  // this wave's 16 weight rows are row tile (gate * W + u0 + 16 * (wave % 2)) / 16 of the fragment-major arrays: block
  // (row tile, 32-deep block kb, plane) is 1 KiB in lane order -- a wave instruction reads 1 KiB of contiguous memory.
  // (Read from the [4W][W] arrays instead, the same fragment is 16 rows x 64 bytes, every lane a cache line of its own:
  // measured 27 GB/s per CU that way, the whole step slower than with the weights staged through LDS.)
  const int nkb = W >> 5;
  const unsigned bo = (unsigned)((((wave >> 1) * W + u0 + (wave & 1) * 16) >> 4) * nkb * NPL * 512 + lane * 8);
  struct BRegs {
    uint4 h[2], l[2];                                 // the two 32-deep blocks of a k-step, hi and lo
  };
  struct ARegs {
    float4 v[APC];
  };
  auto load_b = [&](BRegs& r, int kt) __attribute__((always_inline)) {
    const bool is_x = kt < nkx;
    const bf16_t* wf = (is_x ? a.KF : a.UF) + bo + (is_x ? kt * 2 : kt * 2 - 2 * nkx) * NPL * 512;
#pragma unroll
    for (int s = 0; s < 2; ++s) r.h[s] = *reinterpret_cast<const uint4*>(wf + s * NPL * 512);
    if (LO) {
#pragma unroll
      for (int s = 0; s < 2; ++s) r.l[s] = *reinterpret_cast<const uint4*>(wf + s * NPL * 512 + 512);
    }
  };
  const float* ax[APC];
  const float* ah[APC];
  auto load_a = [&](ARegs& r, int kt) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < APC; ++j) r.v[j] = *reinterpret_cast<const float4*>((kt < nkx ? ax[j] : ah[j]) + kt * BK);
  };
  auto stage_write = [&](const ARegs& r, int stage) __attribute__((always_inline)) {
    unsigned char* const st = smem + stage * STAGE;
#pragma unroll
    for (int j = 0; j < APC; ++j) {
      const float4 v = r.v[j];
      const unsigned h01 = cvt_pk(v.x, v.y), h23 = cvt_pk(v.z, v.w);
      const int off = pl_off(j * 32 + (tid >> 4), (tid & 15) >> 1) + (tid & 1) * 8;
      *reinterpret_cast<uint2*>(st + off) = uint2{h01, h23};
      if (LO) {
        const unsigned l01 = cvt_pk(sub_f32(v.x, __builtin_bit_cast(float, h01 << 16)), sub_f32(v.y, __builtin_bit_cast(float, h01 & 0xffff0000u)));
        const unsigned l23 = cvt_pk(sub_f32(v.z, __builtin_bit_cast(float, h23 << 16)), sub_f32(v.w, __builtin_bit_cast(float, h23 & 0xffff0000u)));
        *reinterpret_cast<uint2*>(st + A_PLANE + off) = uint2{l01, l23};
      }
    }
  };

  // ---- the weights of the first four k-steps need no index: they go out behind the index loads; then ONE wait for the
  // indices, and everything that hangs on them -- the state rows of two k-steps, the epilogue's inputs -- goes out
  ARegs a0, a1;
  BRegs b0, b1, b2, b3;
  constexpr bool run_main = VAR != 2;
  if (run_main) {
    load_b(b0, 0);
    load_b(b1, 1);
    load_b(b2, 2);
    load_b(b3, 3);
  }
  if (wave < TR / 64) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(run_main ? 4 * 2 * NPL : 0) : "memory");
    idx_lds[0][wave * 64 + lane] = i_si;
    idx_lds[1][wave * 64 + lane] = i_so;
    idx_lds[2][wave * 64 + lane] = i_1;
    idx_lds[3][wave * 64 + lane] = i_2;
  }
  wg_barrier();
  // A piece j: row j * 32 + tid / 16, floats 4 * (tid % 16) .. + 3 of the k-step; epilogue cells: row tid / 8, units u0 + 4 * (tid % 8) .. + 3
  // (16-byte loads and stores -- a dword access costs the CU's address unit as much as a 16-byte one, and there are 13 loads
  // per thread this way, not 44)
#pragma unroll
  for (int j = 0; j < APC; ++j) {
    const int lr = j * 32 + (tid >> 4);
    ax[j] = a.pool + (long)idx_lds[1][lr] * a.slot_ld + (a.x_off >= 0 ? a.x_off : 0) + (tid & 15) * 4;
    ah[j] = a.pool + (long)idx_lds[0][lr] * a.slot_ld + a.h_off + (tid & 15) * 4 - Kx;
  }
  if (run_main) {
    load_a(a0, 0);
    load_a(a1, 1);
  }

  // ---- epilogue inputs (whole groups under uniform branches: no wait in between); EARLY: requested now, used after the loop
  const int eu = u0 + 4 * (tid & 7);
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 ecp[EP], et1[EP][4], et2[EP][4], eb[4];
  auto epi_load = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < 4; ++g) eb[g] = zero4;
    if (a.bias) {
#pragma unroll
      for (int g = 0; g < 4; ++g) eb[g] = *reinterpret_cast<const f32x4*>(a.bias + g * W + eu);
    }
#pragma unroll
    for (int p = 0; p < EP; ++p) {
      const int lr = p * 64 + (tid >> 3);
      ecp[p] = *reinterpret_cast<const f32x4*>(a.pool + (long)idx_lds[0][lr] * a.slot_ld + a.c_off + eu);
#pragma unroll
      for (int g = 0; g < 4; ++g) et1[p][g] = et2[p][g] = zero4;
      if (a.T1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) et1[p][g] = *reinterpret_cast<const f32x4*>(a.T1 + (long)idx_lds[2][lr] * 4 * W + g * W + eu);
      }
      if (a.T2) {
#pragma unroll
        for (int g = 0; g < 4; ++g) et2[p][g] = *reinterpret_cast<const f32x4*>(a.T2 + (long)idx_lds[3][lr] * 4 * W + g * W + eu);
      }
    }
  };
  if (EARLY) epi_load();

  f32x4 acc[RF];
#pragma unroll
  for (int i = 0; i < RF; ++i) acc[i] = zero4;

  auto contract = [&](int stage, const BRegs& b) __attribute__((always_inline)) {
    const unsigned char* const a_hi = smem + stage * STAGE;
    constexpr int RH = RF > 4 ? 4 : RF;               // row fragments in registers at a time
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      frag16 fbh, fbl;
      fbh.u = b.h[s];
      if (LO) fbl.u = b.l[s];
#pragma unroll
      for (int i0 = 0; i0 < RF; i0 += RH) {
        frag16 fah[RH], fal[RH];
#pragma unroll
        for (int i = 0; i < RH; ++i) {
          const int off = pl_off((i0 + i) * 16 + fr, s * 4 + fq);
          fah[i].u = *reinterpret_cast<const uint4*>(a_hi + off);
          if (LO) fal[i].u = *reinterpret_cast<const uint4*>(a_hi + A_PLANE + off);
        }
#pragma unroll
        for (int i = 0; i < RH; ++i) {
          acc[i0 + i] = mfma16(fah[i].v, fbh.v, acc[i0 + i]);
          if (LO) {
            acc[i0 + i] = mfma16(fal[i].v, fbh.v, acc[i0 + i]);
            acc[i0 + i] = mfma16(fah[i].v, fbl.v, acc[i0 + i]);
          }
        }
      }
    }
  };

  // ---- main loop, four k-steps per turn (nkt is a multiple of 4).  Per k-step: state registers -> stage, the state rows two
  // k-steps ahead requested into the registers just emptied, one barrier, MFMAs, the weight fragments four k-steps ahead
  // requested into the registers just used.  A stage is rewritten two k-steps later: every wave has passed the barrier in
  // between, which it reaches with that stage's fragment reads retired.  (The last turn is peeled so that the compiler
  // counts its waits over a fixed sequence of loads, the first one so that the epilogue's inputs stay in flight behind it.)
  if (run_main) {
    auto turn = [&](int kt, auto more) __attribute__((always_inline)) {
      constexpr bool MORE = decltype(more)::value;
      stage_write(a0, 0);
      load_a(a0, kt + 2);
      wg_barrier();
      contract(0, b0);
      if (MORE) load_b(b0, kt + 4);
      stage_write(a1, 1);
      load_a(a1, kt + 3);
      wg_barrier();
      contract(1, b1);
      if (MORE) load_b(b1, kt + 5);
      stage_write(a0, 0);
      if (MORE) load_a(a0, kt + 4);
      wg_barrier();
      contract(0, b2);
      if (MORE) load_b(b2, kt + 6);
      stage_write(a1, 1);
      if (MORE) load_a(a1, kt + 5);
      wg_barrier();
      contract(1, b3);
      if (MORE) load_b(b3, kt + 7);
    };
    int kt = 0;
    if (nkt > 4) {
      turn(0, std::true_type{});
      for (kt = 4; kt + 4 < nkt; kt += 4) turn(kt, std::true_type{});
    }
    turn(kt, std::false_type{});
  }
  if (VAR == 1) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < RF; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (t == 12345.678f) a.pool[0] = 0.f;
    return;
  }

  // ---- cell update: the tile goes through LDS once, so that a thread holds the four gates of its cells
  constexpr int LDP = TC + 4;
  float* const ct = reinterpret_cast<float*>(smem);
  wg_barrier();
#pragma unroll
  for (int i = 0; i < RF; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) ct[(i * 16 + fq * 4 + r) * LDP + wave * 16 + fr] = acc[i][r];
  wg_barrier();
  if (!EARLY) epi_load();
#pragma unroll
  for (int p = 0; p < EP; ++p) {
    const int lr = p * 64 + (tid >> 3);
    if (m0 + lr >= a.n) continue;
    f32x4 z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
      z[g] = *reinterpret_cast<const f32x4*>(ct + lr * LDP + g * 32 + 4 * (tid & 7)) + eb[g] + et1[p][g] + et2[p][g];
    f32x4 c, hv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gi = gate_sigmoid(z[0][k]), gf = gate_sigmoid(z[1][k]), gg = gate_tanh(z[2][k]), go = gate_sigmoid(z[3][k]);
      c[k] = gf * ecp[p][k] + gi * gg;
      hv[k] = go * gate_tanh(c[k]);
    }
    float* out = a.pool + (long)idx_lds[1][lr] * a.slot_ld;
    *reinterpret_cast<f32x4*>(out + a.c_off + eu) = c;
    *reinterpret_cast<f32x4*>(out + a.h_off + eu) = hv;
  }
}

// ---------------------------------------------------------------- output layer: logits over the tied embedding + softmax
// One launch instead of a thin GEMM and a softmax kernel: a workgroup = 16 hypotheses x all characters (V <= 256), so that a
// row's maximum and sum never leave the CU.  The 16 state rows (the top layer's new h, through slot_out) come in coalesced,
// are split and laid into LDS once; wave w contracts characters 32 w .. 32 w + 31 over all of K against embedding fragments
// it loads straight into MFMA operands (each element is used by exactly one wave: no staging, no barrier in the loop), four
// 32-deep blocks in flight per wave.
struct OutSoftmax {
  int n, W, V;
  const float* pool; long slot_ld; const int* slot_out; int h_off;
  const bf16_t* EF; int lo;                           // fragment-major embedding [round_up(V, 32) / 16][W / 32][planes][64][8], pad rows zero
  float* probs; long ldp;
};

// byte offset inside a [16][K] bf16 plane of element k (k % 4 == 0) of row `row`: 16-byte chunks XOR-swizzled by row inside
// groups of 16 chunks (K a multiple of 128), so that the 16 rows of a fragment read fall on different banks
__device__ __forceinline__ unsigned o_off(int row, int k, int K) {
  const int chunk = k >> 3;
  const int pos = (chunk & ~15) | ((chunk ^ row) & 15);
  return (unsigned)((row * K + pos * 8 + (k & 7)) * 2);
}

template <bool LO>
__global__ __launch_bounds__(512, 1) void out_softmax_kernel(const OutSoftmax a) {
  constexpr int LDC = 260;
  const int W = a.W;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const int r0 = blockIdx.x * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const plane_hi = smem;
  unsigned char* const plane_lo = smem + (size_t)16 * W * 2;
  float* const ct = reinterpret_cast<float*>(smem + (size_t)(LO ? 2 : 1) * 16 * W * 2);

  const int my_row = min(r0 + (lane & 15), a.n - 1);
  const int my_slot = a.slot_out[my_row];             // lanes 0..15 of every wave: the 16 rows' slots

  // ---- embedding fragments of the first four blocks (they need no index)
  const bool has_cols = wave * 32 < a.V;
  constexpr int NPL = LO ? 2 : 1;
  const int nkb = W >> 5;
  const unsigned eo = (unsigned)(wave * 2 * nkb * NPL * 512 + lane * 8);      // column tile 2 w; tile 2 w + 1 = + nkb blocks
  struct ERegs {
    uint4 h[2], l[2];
  };
  auto load_e = [&](ERegs& r, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j) r.h[j] = *reinterpret_cast<const uint4*>(a.EF + eo + (j * nkb + ks) * NPL * 512);
    if (LO) {
#pragma unroll
      for (int j = 0; j < 2; ++j) r.l[j] = *reinterpret_cast<const uint4*>(a.EF + eo + (j * nkb + ks) * NPL * 512 + 512);
    }
  };
  // (eight blocks in flight per wave = 256 KiB per CU: with only n / 16 workgroups on the chip nothing but a CU's own
  // requests in flight sets its rate -- four blocks: 9.6 us per launch at width 512, 53 GB/s per CU)
  ERegs e0, e1, e2, e3, e4, e5, e6, e7;
  const int nks = W >> 5;
  if (has_cols) {
    load_e(e0, 0);
    load_e(e1, 1);
    load_e(e2, 2);
    load_e(e3, 3);
    if (nks > 4) {
      load_e(e4, 4);
      load_e(e5, 5);
      load_e(e6, 6);
      load_e(e7, 7);
    }
  }

  // ---- state rows: coalesced (a wave instruction = 1 KiB of one row), split, into LDS; four pieces per thread in flight
  {
    const int per_row = W >> 2;                       // float4 pieces per row (a multiple of 32)
    const int total = 16 * per_row;
    for (int p0 = 0; p0 < total; p0 += 4 * 512) {
      float4 v[4];
      int rr[4], kk[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int e = min(p0 + i * 512 + tid, total - 1);
        rr[i] = e / per_row;
        kk[i] = (e - rr[i] * per_row) * 4;
        // (per_row >= 64 or the wave's 64 pieces split over two rows at a 32-lane boundary: the slot comes per lane)
        const int sl = __shfl(my_slot, rr[i] & 15);
        v[i] = *reinterpret_cast<const float4*>(a.pool + (long)sl * a.slot_ld + a.h_off + kk[i]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (p0 + i * 512 + tid < total) {
          const unsigned h01 = cvt_pk(v[i].x, v[i].y), h23 = cvt_pk(v[i].z, v[i].w);
          const unsigned off = o_off(rr[i], kk[i], W);
          *reinterpret_cast<uint2*>(plane_hi + off) = uint2{h01, h23};
          if (LO) {
            const unsigned l01 = cvt_pk(sub_f32(v[i].x, __builtin_bit_cast(float, h01 << 16)), sub_f32(v[i].y, __builtin_bit_cast(float, h01 & 0xffff0000u)));
            const unsigned l23 = cvt_pk(sub_f32(v[i].z, __builtin_bit_cast(float, h23 << 16)), sub_f32(v[i].w, __builtin_bit_cast(float, h23 & 0xffff0000u)));
            *reinterpret_cast<uint2*>(plane_lo + off) = uint2{l01, l23};
          }
        }
      }
    }
  }
  wg_barrier();

  // ---- contraction, no barrier: turns of eight blocks, each set of fragments reloaded eight blocks ahead right behind its
  // use; the last turn peeled (its waits are counted over a fixed sequence of loads)
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  if (has_cols) {
    auto contract = [&](int ks, const ERegs& e) __attribute__((always_inline)) {
      const unsigned off = o_off(fr, ks * 32 + fq * 8, W);
      frag16 ah, al, bh, bl;
      ah.u = *reinterpret_cast<const uint4*>(plane_hi + off);
      if (LO) al.u = *reinterpret_cast<const uint4*>(plane_lo + off);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh.u = e.h[j];
        acc[j] = mfma16(ah.v, bh.v, acc[j]);
        if (LO) {
          bl.u = e.l[j];
          acc[j] = mfma16(al.v, bh.v, acc[j]);
          acc[j] = mfma16(ah.v, bl.v, acc[j]);
        }
      }
    };
    auto turn = [&](int ks, auto more) __attribute__((always_inline)) {
      constexpr bool MORE = decltype(more)::value;
      contract(ks, e0);
      if (MORE) load_e(e0, ks + 8);
      contract(ks + 1, e1);
      if (MORE) load_e(e1, ks + 9);
      contract(ks + 2, e2);
      if (MORE) load_e(e2, ks + 10);
      contract(ks + 3, e3);
      if (MORE) load_e(e3, ks + 11);
      contract(ks + 4, e4);
      if (MORE) load_e(e4, ks + 12);
      contract(ks + 5, e5);
      if (MORE) load_e(e5, ks + 13);
      contract(ks + 6, e6);
      if (MORE) load_e(e6, ks + 14);
      contract(ks + 7, e7);
      if (MORE) load_e(e7, ks + 15);
    };
    if (nks == 4) {      // width 128
      contract(0, e0);
      contract(1, e1);
      contract(2, e2);
      contract(3, e3);
    } else {             // widths of 256, 512, ...: W / 32 a multiple of 8
      int ks = 0;
      for (; ks + 8 < nks; ks += 8) turn(ks, std::true_type{});
      turn(ks, std::false_type{});
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) ct[(fq * 4 + r) * LDC + wave * 32 + j * 16 + fr] = acc[j][r];
  }
  wg_barrier();

  // ---- softmax: wave w takes rows 2 w and 2 w + 1, a lane four characters
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int lr = wave * 2 + q;
    const int row = r0 + lr;
    if (row >= a.n) continue;
    const int v0 = lane * 4;
    float e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) e[k] = v0 + k < a.V ? ct[lr * LDC + v0 + k] : -INFINITY;
    float mx = fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      e[k] = v0 + k < a.V ? expf(e[k] - mx) : 0.f;
      sum += e[k];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    const float inv = 1.f / sum;
    float* out = a.probs + (long)row * a.ldp + v0;
    if (v0 + 3 < a.V && (a.ldp & 3) == 0 && ((size_t)a.probs & 15) == 0) {
      *reinterpret_cast<float4*>(out) = float4{e[0] * inv, e[1] * inv, e[2] * inv, e[3] * inv};
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (v0 + k < a.V) out[k] = e[k] * inv;
    }
  }
}


// in [rows][K] (row stride ld) bf16, hi and (lo or null) -> fragment-major out [rows / 16][K / 32][planes][64 lanes][8]: lane
// (column c = lane % 16, quarter q = lane / 16) of block (row tile rt, kb) holds in[rt * 16 + c][kb * 32 + 8 q .. + 7], the
// B operand of one MFMA 16x16x32 in the order its lanes take it
__global__ void frag_major_kernel(const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo, int rows, int K, long ld,
                                  bf16_t* __restrict__ out) {
  const int npl = lo ? 2 : 1;
  const long total = (long)(rows >> 4) * (K >> 5) * npl * 64;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(e & 63);
    const long blk = e >> 6;
    const int plane = (int)(blk % npl);
    const long t = blk / npl;
    const int kb = (int)(t % (K >> 5));
    const int rt = (int)(t / (K >> 5));
    const bf16_t* src = (plane ? lo : hi) + (long)(rt * 16 + (lane & 15)) * ld + kb * 32 + (lane >> 4) * 8;
    *reinterpret_cast<uint4*>(out + e * 8) = *reinterpret_cast<const uint4*>(src);
  }
}

}  // namespace

// one LSTM cell step of a layer for n hypotheses with pool slots; KL_ERR_SHAPE = not applicable (the caller takes
// step_big.hip's gather + GEMM path)
int kl_launch_frag_major(const bf16_t* hi, const bf16_t* lo, int rows, int K, long ld, bf16_t* out, hipStream_t stream) {
  if ((rows & 15) || (K & 31) || (ld & 7) || !hi || !out) return KL_ERR_SHAPE;
  long g = ((long)(rows >> 4) * (K >> 5) * (lo ? 2 : 1) * 64 + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(frag_major_kernel, dim3((unsigned)g), dim3(256), 0, stream, hi, lo, rows, K, ld, out);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

template <int TR>
static int launch_inc_tile(const KlIncCellArgs& p, int variant, hipStream_t stream) {
  const int W = p.W;
  const bool lo = p.split == 3;
  IncTile a;
  memset(&a, 0, sizeof(a));
  a.n = p.n; a.W = W; a.pool = p.pool; a.slot_ld = p.slot_ld; a.slot_in = p.slot_in; a.slot_out = p.slot_out;
  a.h_off = p.h_off; a.c_off = p.c_off; a.x_off = p.x_off;
  a.UF = p.UF; a.KF = p.KF;
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
  size_t lds = (size_t)2 * (lo ? 2 : 1) * TR * 128;      // two stages of the state rows' planes ...
  if (lds < (size_t)TR * (TC + 4) * 4) lds = (size_t)TR * (TC + 4) * 4;      // ... or the epilogue's f32 tile
#define KL_IT_CASE(LO_, VAR_)                                                                                               \
  do {                                                                                                                      \
    static KlLdsGrant grant;        /* (per instantiation and device: the attribute is set when a launch needs more than any before) */ \
    if (kl_grant_lds(grant, reinterpret_cast<const void*>(&inc_tile_kernel<TR, LO_, VAR_>), lds)) return KL_ERR_LAUNCH;     \
    hipLaunchKernelGGL((inc_tile_kernel<TR, LO_, VAR_>), dim3(a.nx * a.ny), dim3(512), lds, stream, a);                     \
  } while (0)
  if (!lo) KL_IT_CASE(false, 0);
  else if (variant == 1) KL_IT_CASE(true, 1);
  else if (variant == 2) KL_IT_CASE(true, 2);
  else KL_IT_CASE(true, 0);
#undef KL_IT_CASE
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

// one LSTM cell step of a layer for n hypotheses with pool slots; KL_ERR_SHAPE = not applicable (the caller takes
// step_small.hip's kernel or step_big.hip's gather + GEMM path).  Tiles of 128 rows once 64-row tiles would be two or more
// per CU (width 1024 from 512 hypotheses, width 512 from 2048): a unit block's weights are then read by half as many tiles
// -- the launches run at the L2's rate, so the bytes are the time.  (rows = -1: by that rule; KL_TILE_ROWS forces 64 / 128.)
int kl_launch_inc_tile(const KlIncCellArgs& p, int variant, hipStream_t stream, int rows) {
  const int W = p.W;
  if (p.n < 1 || (W & 255) || !p.pool || !p.slot_in || !p.slot_out || !p.UF) return KL_ERR_SHAPE;
  if ((long)8 * W * W >= (1L << 31)) return KL_ERR_SHAPE;
  if (p.x_off >= 0 && !p.KF) return KL_ERR_ARG;
  if (rows != 64 && rows != 128) rows = (long)(W / 32) * ((p.n + 63) / 64) >= 512 ? 128 : 64;
  return rows == 128 ? launch_inc_tile<128>(p, variant, stream) : launch_inc_tile<64>(p, variant, stream);
}

// probs[n][V] = softmax(h_top . E^T) for the top layer's new h rows (pool slots slot_out); KL_ERR_SHAPE = not applicable (the
// caller takes the thin GEMM + the softmax kernel)
int kl_launch_out_softmax(const float* pool, long slot_ld, const int* slot_out, int h_off, const bf16_t* EF, int split,
                          int n, int W, int V, float* probs, long ldp, hipStream_t stream) {
  if (n < 1 || V < 1 || V > 256 || (W != 128 && (W & 255)) || !pool || !slot_out || !EF || !probs) return KL_ERR_SHAPE;
  OutSoftmax a;
  memset(&a, 0, sizeof(a));
  a.n = n; a.W = W; a.V = V; a.pool = pool; a.slot_ld = slot_ld; a.slot_out = slot_out; a.h_off = h_off;
  a.EF = EF; a.probs = probs; a.ldp = ldp;
  const bool lo = split == 3;
  a.lo = lo;
  const size_t lds = (size_t)(lo ? 2 : 1) * 16 * W * 2 + (size_t)16 * 260 * 4;
  if (lds > 150 * 1024) return KL_ERR_SHAPE;
  const dim3 grid((n + 15) / 16);
  static KlLdsGrant grant[2];      // (per instantiation and device: the attribute is set when a launch needs more than any before)
  const void* fn = lo ? reinterpret_cast<const void*>(&out_softmax_kernel<true>) : reinterpret_cast<const void*>(&out_softmax_kernel<false>);
  if (kl_grant_lds(grant[lo], fn, lds)) return KL_ERR_LAUNCH;
  if (lo) hipLaunchKernelGGL(out_softmax_kernel<true>, grid, dim3(512), lds, stream, a);
  else hipLaunchKernelGGL(out_softmax_kernel<false>, grid, dim3(512), lds, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}
