// Incremental step for 16..255 hypotheses (the reference's callers: rate_best feeds at most 128 rows per call, generate
// at most 256; rating.py:49, 704, 809).
//
// inc_cell_kernel: one workgroup = 16 hidden units x 4 gates (four MFMA column tiles) x 16 or 32 hypotheses, all of K, eight
// waves.  The state rows come in COALESCED (a wave stages whole rows, 1 KiB per wave instruction, through the pool slots),
// are split into bf16 hi + lo once and laid into LDS (16-byte chunks XOR-swizzled by row, so that the 16 rows of an MFMA
// fragment fall on different banks); every wave then contracts an eighth of K against weight fragments it loads straight
// into registers from the FRAGMENT-MAJOR copies of the weights (step_tile.hip's frag_major_kernel: a fragment = 1 KiB of
// contiguous memory; each weight element is used by exactly one wave: no staging), the eight partial tiles meet in LDS and
// the cell update runs on the reduced tile.
// Measured at width 512, depth 2, split precision (round 3), us per step of the whole incremental step:
//   128 hypotheses: 43.6 (round 2: four-unit thin workgroups) -> 35.8 (first cut of this kernel: 4 waves, fragments from the
//   [4W][W] arrays) -> 29.0 (fragment-major weights) -> 24.2 (8 waves, scalar base addresses, exp2 / rcp gates; the kernel
//   itself 10.9 -> 8.3 us per layer); 16 / 32 / 64 / 80 hypotheses 22.3 / 22.6 / 22.9 / 23.2 against 23.7 / 27.6 / 28.6 / 39.8
//   with the thin kernels, which now only serve fewer than 16 rows and widths above 1024 that are not multiples of 256.
//   Widths 64 and 128 (the reference's README example and its published model; GEN instantiation): 27 -> 21 us per step at
//   32 / 128 hypotheses, 1024 hypotheses 37.4 -> 19.6 us at width 128 (two row tiles per workgroup, any number of rows).
// What the stamps say about the rest (tools/probe_inc_stamps.py): of a launch's ~11 000 (layer 0) / ~18 000 clocks, 2 500-6 000
// pass between the request of the slot indices and their arrival, ~1 500 more until the rows are in LDS -- the chain
// indices -> rows -> LDS -> MFMA -> exchange -> cell, every link a memory round trip, is the launch.
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
// position c ^ (r & 15) of its 128-byte-aligned group of 16 chunks (K a multiple of 128 elements = 16 chunks, or K = 64)
__device__ __forceinline__ unsigned a_off(int row, int k, int K) {      // byte offset inside a plane of element k of row `row` (k % 4 == 0)
  const int chunk = k >> 3;
  const int m = K >= 128 ? 15 : 7;                 // (K = 64: eight chunks per row, rows r and r + 8 share positions)
  const int pos = (chunk & ~m) | ((chunk ^ row) & m);
  return (unsigned)((row * K + pos * 8 + (k & 7)) * 2);
}

// diagnostic build only (-DKL_STAMP, tools/probe_inc_stamps.py): shader clock at the phases of one workgroup's wave 0,
// layer 0 in [0, 16), the layers above in [16, 32)
#ifdef KL_STAMP
__device__ unsigned long long kl_inc_stamps[32];
#define ISTAMP(i)                                                                              \
  do {                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    if (blockIdx.x == 5 && blockIdx.y == 3 && threadIdx.x == 0) kl_inc_stamps[(a.x_off >= 0 ? 16 : 0) + (i)] = clock64(); \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  } while (0)
#define IWAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define ISTAMP(i)
#define IWAIT()
#endif

struct IncCell {
  int n, W;
  float* pool; long slot_ld;
  const int* slot_in; const int* slot_out;
  int h_off, c_off, x_off;          // float offsets inside a slot: this layer's h and c, the layer below's h (-1: layer 0)
  const bf16_t* UF; const bf16_t* KF;      // fragment-major [4W / 16][W / 32][planes][64][8]
  const float* T1; const int* i1; const float* T2; const int* i2; const float* bias;       // z init (tables [.][4W], bias [4W])
  int* slots_copy;                  // HX kernels, layer 0: slot_out of every row is also left here for the launches behind
};

// HX kernels (kl_step_batch_host): the per-hypothesis indices arrive IN THE KERNEL ARGUMENTS (KlHostIdx, kl_kernels.h) -- the
// dispatch packet's own memory, no copy to the device in front of the launch and no index array to chase through L2.  They are
// read straight from the kernarg segment (constant address space: scalar loads where the row is uniform); the by-value
// parameter itself is never named, so the compiler keeps no private copy of it.
typedef const int __attribute__((address_space(4))) * kx_int_p;
typedef const unsigned short __attribute__((address_space(4))) * kx_u16_p;
constexpr int KX_OFF = (int)((sizeof(IncCell) + 7) & ~(size_t)7);      // offset of the KlHostIdx parameter behind IncCell

// hardware exp2 / reciprocal gates (1 ulp each; step_tile.hip's): the ocml forms cost ~250 instructions per cell
__device__ __forceinline__ float gate_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float gate_tanh(float x) {
  return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.f;
}

// grid (W / 16, ceil(n / (16 NMT))), 512 threads = 8 waves (K split).
// What the stamps of the 4-wave form showed (tools/probe_inc_stamps.py, 128 hypotheses, width 512: 19 000 - 24 000 clocks per
// launch): the launch is bound by INSTRUCTION ISSUE, not by memory -- ~4 000 straight-line instructions per wave at one wave
// per SIMD and 4-5 clocks each; a third of them 64-bit address arithmetic on the vector unit, because the wave index was not
// known to be uniform.  Hence: eight waves (two per SIMD, half the k-steps and half the staging each), the wave index
// through readfirstlane so that every base address is computed on the scalar unit (loads take the SGPR-base form), no
// division in the staging (a wave stages whole rows), hardware exp2 / rcp gates.
// GEN: widths 64 and 128 (the reference's README example and its published model), K = 64, 128 or 256 -- a wave instruction of the
// staging covers several rows, or the x and the h part of one; else W is a multiple of 256 and a wave stages whole rows.
template <int NMT, bool LO, bool GEN, bool HX>
__device__ __forceinline__ void inc_cell_body(const IncCell& a) {
  constexpr int ROWS = 16 * NMT, NW = 8, RPW = ROWS / NW, NPL = LO ? 2 : 1;
  const char __attribute__((address_space(4))) * const kx =
      (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + KX_OFF;
  auto slot_in_of = [&](int row) __attribute__((always_inline)) {
    return HX ? ((kx_int_p)(kx + offsetof(KlHostIdx, slot_in)))[row] : a.slot_in[row];
  };
  auto slot_out_of = [&](int row) __attribute__((always_inline)) {
    return HX ? ((kx_int_p)(kx + offsetof(KlHostIdx, slot_out)))[row] : a.slot_out[row];
  };
  const int W = a.W;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int u0 = blockIdx.x * 16, r0 = blockIdx.y * ROWS;
  const int Kx = a.x_off >= 0 ? W : 0;
  const int K = Kx + W;                            // [x | h] or h alone
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const plane_hi = smem;
  unsigned char* const plane_lo = smem + (size_t)ROWS * K * 2;
  // ---- order of the requests (loads return in order, and a CU serves its requests in order): the index arrays; ONE k-step
  // of weights (it depends on nothing); -- one wait for the indices --; the state rows of the first staging round (the head of
  // the chain rows -> LDS -> barrier -> MFMA: whatever is issued ahead of them delays them); the rest of the first weight group;
  // the epilogue's inputs (table rows, bias, c_prev: needed last).
  ISTAMP(0);
  // (Every workgroup of a row tile -- W / 16 of them, eight waves each -- asks for the same few cache lines of the index
  // arrays at the same moment, and an L2 channel answers the requests for one line one after the other: the stamps showed
  // 2 500 - 6 000 clocks between the request of the indices and their arrival.  So a wave asks only for what it needs: the
  // slots of the rows it stages by SCALAR loads, the epilogue's indices in waves 0 .. 3 alone.)
  const int ej = tid & 15, er = (tid >> 4) & 15;   // epilogue cell of threads 0 .. 255: (row er of each row tile, unit u0 + ej)
  int e_in[NMT], e_out[NMT], e_i1[NMT], e_i2[NMT];
#pragma unroll
  for (int m = 0; m < NMT; ++m) e_in[m] = e_out[m] = e_i1[m] = e_i2[m] = 0;
  if (wave < 4) {
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
      const int row = min(r0 + m * 16 + er, a.n - 1);
      e_in[m] = slot_in_of(row);
      e_out[m] = slot_out_of(row);
      if (HX) {      // (a.i1 / a.i2 non-null only say "by index": the values are the kernel arguments')
        e_i1[m] = a.i1 ? (int)((kx_u16_p)(kx + offsetof(KlHostIdx, idx)))[row] : row;
        e_i2[m] = a.i2 ? (int)((kx_u16_p)(kx + offsetof(KlHostIdx, ctx)))[row] : row;
      } else {
        e_i1[m] = a.i1 ? a.i1[row] : row;
        e_i2[m] = a.i2 ? a.i2[row] : row;
      }
    }
  }
  if (HX && a.slots_copy && blockIdx.x == 0 && wave == 7 && lane < ROWS && r0 + lane < a.n)
    a.slots_copy[r0 + lane] = slot_out_of(r0 + lane);
  // the slots of this wave's RPW staging rows (uniform addresses: scalar loads); GEN: lane r < ROWS holds the slots of row r
  int sl_in[RPW], sl_out[RPW];
  int sl_all_in = 0, sl_all_out = 0;
  if (GEN) {
    const int row = min(r0 + (lane < ROWS ? lane : 0), a.n - 1);
    sl_all_in = slot_in_of(row);
    sl_all_out = slot_out_of(row);
  }
#pragma unroll
  for (int rr = 0; rr < RPW; ++rr) {
    const int row = min(r0 + wave * RPW + rr, a.n - 1);
    sl_in[rr] = GEN ? 0 : slot_in_of(row);
    sl_out[rr] = GEN ? 0 : slot_out_of(row);
  }

  const int nks = K >> 5, nks_x = Kx >> 5;         // k-steps of 32; the first nks_x belong to x . K
  const int col = lane & 15, kg = lane >> 4;
  // (fragment-major arrays: block (row tile, 32-deep block, plane) = 1 KiB in lane order, see step_tile.hip's frag_major_kernel
  // -- read from the [4W][W] arrays a fragment is 16 rows x 64 bytes, every lane a cache line of its own: 20-27 GB/s per CU)
  const int nkb = W >> 5;
  auto wfrag = [&](int ks, int g, int plane) -> u32x4_t {
    const bool is_x = ks < nks_x;
    const bf16_t* base = (is_x ? a.KF : a.UF) + ((long)(((g * W + u0) >> 4) * nkb + (is_x ? ks : ks - nks_x)) * NPL + plane) * 512;
    return *reinterpret_cast<const u32x4_t*>(base + lane * 8);
  };
  // ALL weight fragments of up to GS k-steps of this wave go out at once (GS x 8 fragments = 128 registers in split precision)
  constexpr int GS = 4;
  u32x4_t bh[GS][4], bl[GS][4];
  const int my_steps = (nks - wave + NW - 1) / NW;      // k-steps wave, wave + 8, ...
  auto load_group = [&](int s0, int j0, int j1) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < GS; ++j) {
      if (j >= j0 && j < j1 && s0 + j < my_steps) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bh[j][g] = wfrag(wave + NW * (s0 + j), g, 0);
          if (LO) bl[j][g] = wfrag(wave + NW * (s0 + j), g, 1);
        }
      }
    }
  };
  load_group(0, 0, 1);
  ISTAMP(1);

  // ---- activation rows: wave w stages rows RPW w .. RPW w + RPW - 1, whole rows (a wave instruction = 256 consecutive floats
  // of one row: K / 256 pieces per lane and row), split, into LDS; AP pieces per thread and round
  constexpr int AP = 8;
  const int cpr = GEN ? 1 : K >> 8;
  const bool cpr_pow2 = (cpr & (cpr - 1)) == 0;
  const int cshift = __builtin_ctz(cpr);
  // GEN: the workgroup's ROWS x K floats as one run of 256-float pieces, piece p to wave p % 8; lane: row = e / K, k = e % K
  const int kshift = __builtin_ctz(K);
  const int npc = GEN ? ((ROWS * K >> 8) - wave + NW - 1) / NW : RPW * cpr;
  float4 av[AP];
  auto piece = [&](int p, int& rr, int& c) __attribute__((always_inline)) {
    rr = cpr_pow2 ? p >> cshift : p / cpr;
    c = p - rr * cpr;
  };
  auto pick = [&](const int (&v)[RPW], int rr) __attribute__((always_inline)) {      // v[rr] for a uniform rr without indexing registers
    int x = v[0];
#pragma unroll
    for (int q = 1; q < RPW; ++q) x = rr == q ? v[q] : x;
    return x;
  };
  auto rows_load = [&](int p0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      if (p0 + i < npc) {
        if (GEN) {
          const int e = (wave + NW * (p0 + i)) * 256 + lane * 4;
          const int row = e >> kshift, k = e & (K - 1);
          const int so = __shfl(sl_all_out, row), si = __shfl(sl_all_in, row);
          const float* src = k < Kx ? a.pool + (long)so * a.slot_ld + a.x_off + k : a.pool + (long)si * a.slot_ld + a.h_off + (k - Kx);
          av[i] = *reinterpret_cast<const float4*>(src);
        } else {
          int rr, c;
          piece(p0 + i, rr, c);
          const int so = pick(sl_out, rr), si = pick(sl_in, rr);
          const int k0 = c * 256;                    // (W is a multiple of 256: a piece lies in x or in h)
          const float* src = k0 < Kx ? a.pool + (long)so * a.slot_ld + a.x_off + k0 : a.pool + (long)si * a.slot_ld + a.h_off + (k0 - Kx);
          av[i] = *reinterpret_cast<const float4*>(src + lane * 4);
        }
      }
    }
  };
  auto rows_store = [&](int p0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < AP; ++i) {
      if (p0 + i < npc) {
        uint2 hi, lo;
        split4<LO>(av[i], hi, lo);
        unsigned off;
        if (GEN) {
          const int e = (wave + NW * (p0 + i)) * 256 + lane * 4;
          off = a_off(e >> kshift, e & (K - 1), K);
        } else {
          int rr, c;
          piece(p0 + i, rr, c);
          off = a_off(wave * RPW + rr, c * 256 + lane * 4, K);
        }
        *reinterpret_cast<uint2*>(plane_hi + off) = hi;
        if (LO) *reinterpret_cast<uint2*>(plane_lo + off) = lo;
      }
    }
  };
  rows_load(0);
  ISTAMP(2);
  load_group(0, 1, GS);

  // ---- the epilogue's inputs (threads 0 .. 255)
  float eb[4], et1[NMT][4], et2[NMT][4], ecp[NMT];
#pragma unroll
  for (int g = 0; g < 4; ++g) eb[g] = 0.f;
#pragma unroll
  for (int m = 0; m < NMT; ++m) {
    ecp[m] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) et1[m][g] = et2[m][g] = 0.f;
  }
  if (wave < 4) {
    if (a.bias) {
#pragma unroll
      for (int g = 0; g < 4; ++g) eb[g] = a.bias[g * W + u0 + ej];
    }
#pragma unroll
    for (int m = 0; m < NMT; ++m) ecp[m] = a.pool[(long)e_in[m] * a.slot_ld + a.c_off + u0 + ej];
    if (a.T1) {
#pragma unroll
      for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) et1[m][g] = a.T1[(long)e_i1[m] * 4 * W + g * W + u0 + ej];
    }
    if (a.T2) {
#pragma unroll
      for (int m = 0; m < NMT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) et2[m][g] = a.T2[(long)e_i2[m] * 4 * W + g * W + u0 + ej];
    }
  }
  ISTAMP(3);
  rows_store(0);
  ISTAMP(5);
  for (int p0 = AP; p0 < npc; p0 += AP) {
    rows_load(p0);
    rows_store(p0);
  }
  __syncthreads();
  ISTAMP(6);
#ifdef KL_STAMP
  IWAIT();
  ISTAMP(7);
#endif

  // ---- contraction: wave w takes k-steps w, w + 8, ...; 3 MFMAs per (row tile, gate) and k-step in split precision
  f32x4 acc[NMT][4];
#pragma unroll
  for (int m = 0; m < NMT; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[m][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int s0 = 0; s0 < my_steps; s0 += GS) {
    if (s0 > 0) load_group(s0, 0, GS);
#pragma unroll
    for (int j = 0; j < GS; ++j) {
      if (s0 + j < my_steps) {
        const int ks = wave + NW * (s0 + j);
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
  ISTAMP(8);
  __syncthreads();      // (every wave has read its last fragments: the planes make room for the partial tiles)
  float* const part = reinterpret_cast<float*>(smem);      // [8 waves][NMT][4 gates][16 rows][17]
#pragma unroll
  for (int m = 0; m < NMT; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[(((wave * NMT + m) * 4 + g) * 16 + kg * 4 + r) * 17 + col] = acc[m][g][r];
  __syncthreads();

  ISTAMP(9);
  // ---- cell update on the reduced tile: thread = (row, unit), NMT cells each, threads 0 .. 255
  if (wave < 4) {
#pragma unroll
    for (int m = 0; m < NMT; ++m) {
      const int row = r0 + m * 16 + er;
      if (row >= a.n) continue;
      const int u = u0 + ej;
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v = eb[g] + et1[m][g] + et2[m][g];
#pragma unroll
        for (int w = 0; w < NW; ++w) v += part[(((w * NMT + m) * 4 + g) * 16 + er) * 17 + ej];
        z[g] = v;
      }
      const float gi = gate_sigmoid(z[0]), gf = gate_sigmoid(z[1]), gg = gate_tanh(z[2]), go = gate_sigmoid(z[3]);
      const float c = gf * ecp[m] + gi * gg;
      const float h = go * gate_tanh(c);
      float* out = a.pool + (long)e_out[m] * a.slot_ld;
      out[a.c_off + u] = c;
      out[a.h_off + u] = h;
    }
  }
  ISTAMP(10);
#ifdef KL_STAMP
  IWAIT();
  ISTAMP(11);
#endif
}

template <int NMT, bool LO, bool GEN>
__global__ __launch_bounds__(512) void inc_cell_kernel(const IncCell a) {
  inc_cell_body<NMT, LO, GEN, false>(a);
}

// ... with the indices in the kernel arguments (hx is read through the kernarg segment pointer, see KX_OFF)
template <int NMT, bool LO, bool GEN>
__global__ __launch_bounds__(512) void inc_cell_hx_kernel(const IncCell a, const KlHostIdx hx) {
  inc_cell_body<NMT, LO, GEN, true>(a);
}

constexpr size_t LDS_LIMIT = 150 * 1024;

// ---- last launch of a host-driven step (kl_step_batch_host): softmax over the logits rows (the arithmetic of
// softmax_ce_kernel: first maximum, expf, one division) and delivery INTO HOST MEMORY -- the probabilities (whole rows, or per
// row the one character the caller will look at), optionally the first head_k state vectors of every new state (history
// clustering, rating.py:887-916) --, then one word that tells the waiting host thread that everything has arrived: every
// workgroup fences its stores at system scope and takes a ticket, the last one resets the counter and writes `ticket` to
// *done (the host spins on that word instead of synchronising with the stream: kl_step_wait).
constexpr int FX_OFF = (int)((sizeof(KlStepFinish) + 7) & ~(size_t)7);      // offset of the KlHostTargets parameter
__global__ __launch_bounds__(256) void step_finish_kernel(const KlStepFinish a, const KlHostTargets tx) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row < a.n) {
    const float* x = a.logits + (long)row * a.ld;
    int t = -1;
    if (a.by_target) {
      if (a.target) t = a.target[row];
      else t = ((kx_u16_p)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + FX_OFF))[row];
    }
    float mx = 0.f, inv = 1.f;
    if (a.softmax) {
      mx = -INFINITY;
      for (int v = lane; v < a.V; v += 64) mx = fmaxf(mx, x[v]);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
      float sum = 0.f;
      for (int v = lane; v < a.V; v += 64) sum += expf(x[v] - mx);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
      inv = 1.f / sum;
    }
    if (a.by_target) {
      if (lane == 0) a.probs_host[row] = (t >= 0 && t < a.V) ? (a.softmax ? expf(x[t] - mx) * inv : x[t]) : 0.f;
    } else {
      float* out = a.probs_host + (long)row * a.V;
      for (int v = lane; v < a.V; v += 64) out[v] = a.softmax ? expf(x[v] - mx) * inv : x[v];
    }
    if (a.head_k > 0) {
      const float4* src = reinterpret_cast<const float4*>(a.pool + (long)a.slot_out[row] * a.slot_ld);
      float4* dst = reinterpret_cast<float4*>(a.heads_host + (long)row * a.head_k * a.W);
      const int n4 = a.head_k * a.W / 4;      // (the head vectors are the first head_k * W floats of a slot; W % 4 == 0)
      for (int i = lane; i < n4; i += 64) dst[i] = src[i];
    }
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned got = atomicAdd(a.counter, 1u);
    if (got == gridDim.x - 1) {
      atomicExch(a.counter, 0u);
      __threadfence_system();
      __hip_atomic_store(a.done_host, a.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

}  // namespace

#ifdef KL_STAMP
extern "C" int kl_test_read_inc_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(kl_inc_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : KL_ERR_LAUNCH;
}
#endif

// one LSTM cell step of layer `l` for n hypotheses with pool slots; KL_ERR_SHAPE = not applicable (the caller takes the
// launch-per-layer kernels of lstm_step.hip).
int kl_launch_inc_cell(const KlIncCellArgs& p, hipStream_t stream, const KlHostIdx* hx, int* slots_copy) {
  const int W = p.W;
  if (p.n < 1 || ((W & 255) && W != 64 && W != 128) || !p.pool || !p.UF) return KL_ERR_SHAPE;
  if (!hx && (!p.slot_in || !p.slot_out)) return KL_ERR_SHAPE;
  if (hx && p.n > KL_HOST_STEP_MAX) return KL_ERR_SHAPE;
  const bool lo = p.split == 3;
  if (p.x_off >= 0 && !p.KF) return KL_ERR_ARG;
  const int K = p.x_off >= 0 ? 2 * W : W;
  const bool gen = (W & 255) != 0;                 // widths 64 and 128: K = 64, 128 or 256, a staging piece spans rows or x and h
  int nmt = p.n > 128 ? 2 : 1;
  if ((size_t)16 * nmt * K * 4 > LDS_LIMIT) nmt = 1;
  size_t lds = (size_t)16 * nmt * K * 4;
  if (lds < (size_t)8 * nmt * 4 * 16 * 17 * 4) lds = (size_t)8 * nmt * 4 * 16 * 17 * 4;      // (the partial tiles re-use the planes' room)
  if (lds > LDS_LIMIT) return KL_ERR_SHAPE;
  IncCell a;
  a.n = p.n; a.W = W; a.pool = p.pool; a.slot_ld = p.slot_ld; a.slot_in = p.slot_in; a.slot_out = p.slot_out;
  a.h_off = p.h_off; a.c_off = p.c_off; a.x_off = p.x_off;
  a.UF = p.UF; a.KF = p.KF;
  a.T1 = p.T1; a.i1 = p.i1; a.T2 = p.T2; a.i2 = p.i2; a.bias = p.bias;
  a.slots_copy = slots_copy;
  dim3 grid(W / 16, (p.n + 16 * nmt - 1) / (16 * nmt));
#define KL_IC_CASE(NMT_, LO_, GEN_)                                                                                         \
  do {                                                                                                                      \
    if (hx) {                                                                                                               \
      static KlLdsGrant grant_hx;                                                                                           \
      if (kl_grant_lds(grant_hx, reinterpret_cast<const void*>(&inc_cell_hx_kernel<NMT_, LO_, GEN_>), lds)) return KL_ERR_LAUNCH; \
      hipLaunchKernelGGL((inc_cell_hx_kernel<NMT_, LO_, GEN_>), grid, dim3(512), lds, stream, a, *hx);                      \
      break;                                                                                                                \
    }                                                                                                                       \
    static KlLdsGrant grant;                                                                                                \
    if (kl_grant_lds(grant, reinterpret_cast<const void*>(&inc_cell_kernel<NMT_, LO_, GEN_>), lds)) return KL_ERR_LAUNCH;   \
    hipLaunchKernelGGL((inc_cell_kernel<NMT_, LO_, GEN_>), grid, dim3(512), lds, stream, a);                                \
  } while (0)
#define KL_IC_CASE2(NMT_, LO_) do { if (gen) KL_IC_CASE(NMT_, LO_, true); else KL_IC_CASE(NMT_, LO_, false); } while (0)
  if (nmt == 2) { if (lo) KL_IC_CASE2(2, true); else KL_IC_CASE2(2, false); }
  else { if (lo) KL_IC_CASE2(1, true); else KL_IC_CASE2(1, false); }
#undef KL_IC_CASE2
#undef KL_IC_CASE
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

int kl_launch_step_finish(const KlStepFinish& a, const KlHostTargets* tx, hipStream_t stream) {
  if (a.n < 1 || !a.logits || !a.probs_host || !a.done_host || !a.counter) return KL_ERR_ARG;
  if (a.head_k > 0 && (!a.pool || !a.slot_out || !a.heads_host || (a.W & 3))) return KL_ERR_ARG;
  if (a.by_target && !a.target && (!tx || a.n > KL_HOST_STEP_MAX)) return KL_ERR_ARG;
  static const KlHostTargets none = {};
  hipLaunchKernelGGL(step_finish_kernel, dim3((a.n + 3) / 4), dim3(256), 0, stream, a, tx ? *tx : none);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}
