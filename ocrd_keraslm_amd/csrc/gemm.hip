// Big-tile bf16 MFMA GEMM for the non-recurrent contractions (F2 input contraction,
// F5 logits, B* weight/input gradients, the big-n incremental step).
//
//   C[M,N] (op)= A[M,K] . B[N,K]^T (+ bias[N])
//
// Both operands are K-contiguous bf16 ("TN"): every producer in this library
// writes the layout its consumer contracts over, so one kernel serves all.
// Tile BM x 128 x 64 (BM = 128 or 64), 256 threads = 4 waves, each wave 64 rows x
// (64 or 32) columns of MFMA 16x16x32 tiles.  LDS: 2 stages x (A + B tiles),
// 128-byte rows XOR-swizzled (chunk ^= (row>>1)&7) so that the ds_read_b128
// fragment reads are bank-conflict free.  Register-staged double buffering:
// the global loads of k-tile t+1 are in flight while k-tile t is in the MFMAs;
// one barrier per k-tile.  BM = 64 is chosen when the 128-row grid would leave
// most CUs idle (M = 1024 hypotheses x N = 2048: 128 vs 256 workgroups).
// Split-K over gridDim.z with f32 atomics for the K = B*T weight-gradient
// shapes (M,N small, K huge).
#include <stdlib.h>
#include <string.h>

#include "kl_common.h"
#include "kl_kernels.h"

namespace {

constexpr int BN = 128, BK = 64;
constexpr int NTHREADS = 256;

__device__ __forceinline__ int lds_off(int row, int chunk) {
  // byte offset of 16-byte chunk `chunk` (0..7) of row `row` in a [rows][64] bf16 tile
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <int OUT, int BM>   // OUT: 0 = f32 store, 1 = bf16 store, 2 = f32 atomic add
__global__ __launch_bounds__(NTHREADS) void gemm_tn_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, void* __restrict__ Cv,
    const float* __restrict__ bias, int M, int N, int K, long lda, long ldb, long ldc,
    int k_per_split, float alpha) {
  constexpr int WM = BM / 64;          // wave rows (1 or 2)
  constexpr int WN = 4 / WM;           // wave columns (4 or 2)
  constexpr int NT = BN / WN / 16;     // 16-wide column tiles per wave (2 or 4)
  constexpr int ALOADS = BM * 8 / NTHREADS;   // 16-byte chunks of the A tile per thread (2 or 4)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                 // 2 stages x BM rows x 128 B
  unsigned char* sB = smem + 2 * BM * 128;  // 2 stages x 128 rows x 128 B

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.y * BM;
  const int n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * k_per_split;
  const int kend = min(K, kbeg + k_per_split);
  const int nkt = (kend - kbeg + BK - 1) / BK;

  f32x4 acc[4][NT];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 ra[ALOADS], rb[4];
  auto gload = [&](int kt) {
    const int k0 = kbeg + kt * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = tid + NTHREADS * j;
      const int row = q >> 3, c = q & 7;
      const int k = k0 + c * 8;
      const bool kok = k < kend;
      const int br = n0 + row;
      rb[j] = (kok && br < N) ? *reinterpret_cast<const uint4*>(B + (long)br * ldb + k) : uint4{0, 0, 0, 0};
      if (j < ALOADS) {
        const int ar = m0 + row;
        ra[j] = (kok && ar < M) ? *reinterpret_cast<const uint4*>(A + (long)ar * lda + k) : uint4{0, 0, 0, 0};
      }
    }
  };
  auto sstore = [&](int stage) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = tid + NTHREADS * j;
      const int row = q >> 3, c = q & 7;
      *reinterpret_cast<uint4*>(sB + stage * BN * 128 + lds_off(row, c)) = rb[j];
      if (j < ALOADS) *reinterpret_cast<uint4*>(sA + stage * BM * 128 + lds_off(row, c)) = ra[j];
    }
  };

  if (nkt > 0) {
    gload(0);
    sstore(0);
  }
  __syncthreads();
  const int fr = lane & 15, fq = lane >> 4;
  for (int kt = 0; kt < nkt; ++kt) {
    const int stage = kt & 1;
    if (kt + 1 < nkt) gload(kt + 1);
    const unsigned char* a_base = sA + stage * BM * 128;
    const unsigned char* b_base = sB + stage * BN * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      frag16 fa[4], fb[NT];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        fa[i].u = *reinterpret_cast<const uint4*>(a_base + lds_off(wm * 64 + i * 16 + fr, s * 4 + fq));
#pragma unroll
      for (int j = 0; j < NT; ++j)
        fb[j].u = *reinterpret_cast<const uint4*>(b_base + lds_off(wn * (BN / WN) + j * 16 + fr, s * 4 + fq));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(fa[i].v, fb[j].v, acc[i][j]);
    }
    if (kt + 1 < nkt) sstore(stage ^ 1);
    __syncthreads();
  }

  // epilogue: D col = lane&15, row = 4*(lane>>4)+r
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + wn * (BN / WN) + j * 16 + fr;
      if (col >= N) continue;
      const float bv = (bias != nullptr && blockIdx.z == 0) ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 64 + i * 16 + fq * 4 + r;
        if (row >= M) continue;
        const float v = acc[i][j][r] * alpha + bv;
        if (OUT == 0) {
          reinterpret_cast<float*>(Cv)[(long)row * ldc + col] = v;
        } else if (OUT == 1) {
          reinterpret_cast<bf16_t*>(Cv)[(long)row * ldc + col] = f2bf(v);
        } else {
          atomicAdd(reinterpret_cast<float*>(Cv) + (long)row * ldc + col, v);
        }
      }
    }
  }
}

template <int BM>
void launch_bm(int out_mode, dim3 grid, hipStream_t stream, const bf16_t* A, const bf16_t* B, void* C, const float* bias,
               int M, int N, int K, long lda, long ldb, long ldc, int k_per_split, float alpha) {
  const size_t lds = 2 * (BM + BN) * 128;
  switch (out_mode) {
    case 0:
      hipLaunchKernelGGL((gemm_tn_kernel<0, BM>), grid, dim3(NTHREADS), lds, stream, A, B, C, bias, M, N, K, lda, ldb, ldc, k_per_split, alpha);
      break;
    case 1:
      hipLaunchKernelGGL((gemm_tn_kernel<1, BM>), grid, dim3(NTHREADS), lds, stream, A, B, C, bias, M, N, K, lda, ldb, ldc, k_per_split, alpha);
      break;
    default:
      hipLaunchKernelGGL((gemm_tn_kernel<2, BM>), grid, dim3(NTHREADS), lds, stream, A, B, C, bias, M, N, K, lda, ldb, ldc, k_per_split, alpha);
      break;
  }
}


// ---------------------------------------------------------------- long-K variant
// For the K = B*T contractions (weight gradients: M, N <= 2048, K ~ 1e5) one workgroup
// per CU stays in its main loop for thousands of k-steps, so the loop itself is what
// counts: 256 x 128 x 64 tile, 512 threads = 8 waves as 4 (M) x 2 (N), each 64 x 64;
// both operand tiles arrive by LDS-DMA (buffer_load ... lds, 1 KiB = 8 rows per wave
// instruction; no staging registers, no ds_write pass) into a 3-deep ring (144 KiB), two
// tiles in flight across ONE raw barrier per k-step and a counted vmcnt.  The LDS image
// is the same XOR-swizzled [rows][64] layout as above; an LDS-DMA writes lane-linear, so
// the swizzle is applied to each lane's SOURCE chunk instead.
// Ordering (MI355X_MICROARCH.md, LDS-DMA): a stage is read only after its issuing waves'
// counted vmcnt AND a barrier the reader has passed; it is restaged only after a barrier
// that every reader reaches with its ds_reads retired (they feed MFMAs issued before it).
#ifndef KL_GEMM_PRIO
#define KL_GEMM_PRIO 1
#endif
constexpr int LBM = 256, LBN = 128, LSTAGES = 3;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4g;

// One LDS-DMA piece: 64 lanes x 16 bytes from rsrc[voff + soff] to LDS [lds_addr, +1 KiB).
// Inline asm on purpose: behind the builtin hipcc drains vmcnt(0) in front of the next
// ds_read (it cannot tell the ring's stages apart), which would serialise the pipeline;
// the waits are counted by hand below.  M0 is compiler-reserved: saved and restored.
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff)
      : "memory");
}

// WM x WN waves, each RF*16 rows x (128 / WN) columns: (RF, WM, WN) = (4, 4, 2) is the 256 x 128
// tile for big grids; (2, 2, 4) a 64 x 128 tile of 8 waves (72 KiB) for M ~ 1e3 shapes (the
// incremental step's GEMMs), where 256-row tiles would leave 3/4 of the CUs idle and two waves
// per SIMD are needed to hide the ds_read -> MFMA latency of the short k-loop.
//
// ATR: the A operand is given K-major, A[k][m] with row stride lda (the backward scan's dZ rows as they
// are: no transposed copy).  Its stage is the [64 k][256 m] image the DMA lays down (512-byte rows,
// chunks XOR-swizzled at the source) and the fragments come from two ds_read_b64_tr_b16 each -- the
// hardware transpose read: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4 x 16
// block and lane i receives column i (cdna_hip_programming.md T10).  BTR: the same for B, B[k][n] with
// row stride ldb ([64 k][128 n] image, 256-byte rows).  c_t (OUT == 2): accumulate C^T.
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// Two products over ONE A operand in one launch: C = A.B^T for the column tiles below n_first, C2 = A.B2^T for the
// tiles from n_first on (n_first a multiple of the 128-column tile).  The weight-gradient contractions of a layer all read
// the same T*B rows of dZ -- 3.2 GB at the benchmark shape, which is what bounds each of them -- so the recurrent and
// the input kernel's gradients (and the two look-up tables' of layer 0) share one pass over it.
struct KlGemmSecond {
  const bf16_t* B;      // null: a single product
  void* C;
  int n_first, N;
  long ldb, ldc;
  int c_transposed;
};

template <int OUT, bool ILV, int RF, int WM, int WN, bool ATR = false, bool BTR = false>
__global__ __launch_bounds__(64 * WM * WN, 1) void gemm_tn_long_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B_, void* __restrict__ Cv_,
    const float* __restrict__ bias, int M, int N_, int K, long lda, long ldb_, long ldc_,
    int k_per_split, float alpha, const KlGateEpi epi, int xcd_remap, int c_t_, const KlGemmSecond second) {
  // (second.B: column tiles from second.n_first on belong to a second product over the same A -- see KlGemmSecond)
  const bf16_t* __restrict__ B = B_;
  void* __restrict__ Cv = Cv_;
  int N = N_, c_t = c_t_;
  long ldb = ldb_, ldc = ldc_;
  static_assert(!(ATR || BTR) || (RF == 4 && WM == 4 && WN == 2), "K-major operands: 256 x 128 tiles only");
  constexpr bool PRIO = KL_GEMM_PRIO;
  constexpr int NW = WM * WN;
  constexpr int WROWS = 16 * RF;               // rows per wave
  constexpr int TBM = WROWS * WM;              // tile rows
  constexpr int WCOLS = LBN / WN;              // columns per wave
  constexpr int NT = WCOLS / 16;               // column fragments per wave
  constexpr int PA = TBM / 8 / NW;             // A pieces (8 rows each) per wave and k-step
  constexpr int PB = LBN / 8 / NW;             // B pieces
  constexpr int NP = PA + PB;                  // pieces per wave and k-step = the counted vmcnt
  constexpr int NP0 = (NP + 1) / 2;            // ... issued with the first MFMA group
  static_assert(PA >= 1 && PB >= 1 && NP <= 6, "piece split");
  constexpr int STAGE_BYTES = (TBM + LBN) * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order (speed only, any placement stays correct): workgroups are dealt round-robin
  // over the 8 XCDs, each with its own L2, so consecutive ids land on different XCDs and the tiles
  // that share an operand (all n-tiles of an m-tile; all tiles of one K split) would each pull it
  // through a different L2.  Re-number so that id % 8 picks a contiguous eighth of the tile sequence.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  {
    const unsigned nx = gridDim.x, ny = gridDim.y, total = nx * ny * gridDim.z;
    if (xcd_remap && (total & 7) == 0) {
      const unsigned lin = bx + nx * (by + ny * bz);
      const unsigned re = (lin & 7) * (total >> 3) + (lin >> 3);
      bx = re % nx;
      by = (re / nx) % ny;
      bz = re / (nx * ny);
    }
  }
  const int m0 = by * TBM;
  int n0 = bx * LBN;
  if (second.B != nullptr) {
    if (n0 >= second.n_first) {
      n0 -= second.n_first;
      B = second.B;
      Cv = second.C;
      N = second.N;
      ldb = second.ldb;
      ldc = second.ldc;
      c_t = second.c_transposed;
    } else {
      N = second.n_first;
    }
  }
  const int kbeg = bz * k_per_split;
  const int kend = min(K, kbeg + k_per_split);
  const int nkt = (kend - kbeg) / BK;     // host guarantees whole k-tiles

  // buffer resources based at this workgroup's first rows; rows past the matrix read as zero
  auto clamp31 = [](long v) { return (int)(v > 0x7fffffffL ? 0x7fffffffL : (v < 0 ? 0 : v)); };
  const __amdgpu_buffer_rsrc_t rsA = ATR
      ? __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A + (long)kbeg * lda + m0), 0, clamp31(((long)(kend - kbeg - 1) * lda + (M - m0)) * 2), 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A + (long)m0 * lda), 0, clamp31(((long)(M - m0 - 1) * lda + K) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = BTR
      ? __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B + (long)kbeg * ldb + n0), 0, clamp31(((long)(kend - kbeg - 1) * ldb + (N - n0)) * 2), 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B + (long)n0 * ldb), 0, clamp31(((long)(N - n0 - 1) * ldb + K) * 2), 0x00020000);
  // per-lane source offsets of this wave's pieces (rows wave*8*P + j*8 + (lane>>3)), swizzled chunk
  unsigned vo[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    if (ATR && j < PA) {
      // piece p = two k-rows of 512 bytes: lanes 0..31 row 2p, lanes 32..63 row 2p + 1
      const int krow = 2 * (wave * PA + j) + (lane >> 5);
      const int c = (lane & 31) ^ (2 * ((krow & 3) | (((krow >> 3) & 1) << 2)));
      vo[j] = (unsigned)((long)krow * lda * 2 + c * 16);
      continue;
    }
    if (BTR && j >= PA) {
      // piece p = four k-rows of 256 bytes, 16 lanes each
      const int krow = 4 * (wave * PB + (j - PA)) + (lane >> 4);
      const int c = (lane & 15) ^ (2 * ((krow & 3) | (((krow >> 3) & 1) << 2)));
      vo[j] = (unsigned)((long)krow * ldb * 2 + c * 16);
      continue;
    }
    const int row = j < PA ? wave * 8 * PA + j * 8 + (lane >> 3) : wave * 8 * PB + (j - PA) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    vo[j] = (unsigned)((long)row * (j < PA ? lda : ldb) * 2 + c * 16);
  }
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
  // the pieces are issued in two halves between the two MFMA groups of a k-step
  auto issue_half = [&](int kt, int stage, int half) {
    const int soff = (kbeg + kt * BK) * 2;
    const int soff_a = ATR ? (int)((long)kt * BK * lda * 2) : soff;     // (K-major A: based at kbeg already)
    const int soff_b = BTR ? (int)((long)kt * BK * ldb * 2) : soff;
    const unsigned sa = lds0 + stage * STAGE_BYTES + wave * 8 * PA * 128;
    const unsigned sb = lds0 + stage * STAGE_BYTES + TBM * 128 + wave * 8 * PB * 128;
#pragma unroll
    for (int j = half ? NP0 : 0; j < (half ? NP : NP0); ++j) {
      if (j < PA) glds16(rsA, vo[j], soff_a, sa + j * 1024);
      else glds16(rsB, vo[j], soff_b, sb + (j - PA) * 1024);
    }
  };
  auto issue = [&](int kt, int stage) {
    issue_half(kt, stage, 0);
    issue_half(kt, stage, 1);
  };

  f32x4 acc[RF][NT];
#pragma unroll
  for (int i = 0; i < RF; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (nkt > 0) issue(0, 0);
  if (nkt > 1) issue(1, 1);
  const int fr = lane & 15, fq = lane >> 4;
  // K-major A: this lane's part of the transposed-read addresses, one per row fragment (the swizzle
  // term is a lane constant: k-rows s*32 + fq*8 + half*4 + q have (krow & 3) = q, bit 3 = fq & 1)
  unsigned atr_off[RF];
  if (ATR) {
    const int q = fr >> 2, pp = fr & 3;
    const int fl = 2 * (q | ((fq & 1) << 2));
#pragma unroll
    for (int i = 0; i < RF; ++i)
      atr_off[i] = (unsigned)((fq * 8 + q) * 512 + ((((wm * WROWS + i * 16) >> 3) ^ fl) + (pp >> 1)) * 16 + 8 * (pp & 1));
  }
  unsigned btr_off[NT];
  if (BTR) {
    const int q = fr >> 2, pp = fr & 3;
    const int fl = 2 * (q | ((fq & 1) << 2));
#pragma unroll
    for (int j = 0; j < NT; ++j)
      btr_off[j] = (unsigned)((fq * 8 + q) * 256 + ((((wn * WCOLS + j * 16) >> 3) ^ fl) + (pp >> 1)) * 16 + 8 * (pp & 1));
  }
  int stage = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool more = kt + 2 < nkt;
    const int nstage = stage >= 1 ? stage - 1 : LSTAGES - 1;   // (kt+2) % 3 == (stage+2) % 3
    if (!ILV && more) issue(kt + 2, nstage);
    const unsigned char* a_base = smem + stage * STAGE_BYTES;
    const unsigned char* b_base = a_base + TBM * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      frag16 fa[RF], fb[NT];
#pragma unroll
      for (int i = 0; i < RF; ++i) {
        if (ATR) {
          const unsigned char* ap = a_base + atr_off[i] + s * 32 * 512;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(ap + 4 * 512));
          fa[i].s[0] = (bf16_t)lo.x; fa[i].s[1] = (bf16_t)lo.y; fa[i].s[2] = (bf16_t)lo.z; fa[i].s[3] = (bf16_t)lo.w;
          fa[i].s[4] = (bf16_t)hi.x; fa[i].s[5] = (bf16_t)hi.y; fa[i].s[6] = (bf16_t)hi.z; fa[i].s[7] = (bf16_t)hi.w;
        } else {
          fa[i].u = *reinterpret_cast<const uint4*>(a_base + lds_off(wm * WROWS + i * 16 + fr, s * 4 + fq));
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (BTR) {
          const unsigned char* bp = b_base + btr_off[j] + s * 32 * 256;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(bp));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(bp + 4 * 256));
          fb[j].s[0] = (bf16_t)lo.x; fb[j].s[1] = (bf16_t)lo.y; fb[j].s[2] = (bf16_t)lo.z; fb[j].s[3] = (bf16_t)lo.w;
          fb[j].s[4] = (bf16_t)hi.x; fb[j].s[5] = (bf16_t)hi.y; fb[j].s[6] = (bf16_t)hi.z; fb[j].s[7] = (bf16_t)hi.w;
        } else {
          fb[j].u = *reinterpret_cast<const uint4*>(b_base + lds_off(wn * WCOLS + j * 16 + fr, s * 4 + fq));
        }
      }
      if (ILV && more) issue_half(kt + 2, nstage, s);
      if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < RF; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(fa[i].v, fb[j].v, acc[i][j]);
      if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    stage = stage + 1 < LSTAGES ? stage + 1 : 0;
  }

  if (OUT == 3) {
    // LSTM cell epilogue (incremental step, step_big.hip): the weight rows are permuted so that
    // this 128-column tile holds the four gates of 32 hidden units; the tile goes through LDS
    // and every thread finishes (row, unit) pairs: z + table rows + bias -> gates -> c', h'
    // written to the output pool slots, and h' as bf16 (hi | lo | hi) into the next layer's
    // activation rows.  No z round trip through HBM, no separate gate kernel.
    constexpr int LDP = LBN + 4;
    float* ct = reinterpret_cast<float*>(smem);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < RF; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ct[(wm * WROWS + i * 16 + fq * 4 + r) * LDP + wn * WCOLS + j * 16 + fr] = acc[i][j][r];
    __syncthreads();
    const int u = tid & 31;
    const int U = bx * 32 + u;
    const int W = epi.W;
    float bg[4] = {0.f, 0.f, 0.f, 0.f};
    if (epi.bias) {
#pragma unroll
      for (int g = 0; g < 4; ++g) bg[g] = epi.bias[(long)g * W + U];
    }
    constexpr int RPP = 64 * NW / 32;
    for (int p = 0; p < TBM / RPP; ++p) {
      const int lrow = p * RPP + (tid >> 5);
      const int row = m0 + lrow;
      if (row >= M) continue;
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) z[g] = ct[lrow * LDP + g * 32 + u] + bg[g];
      if (epi.T1) {
        const float* t1 = epi.T1 + (long)(epi.i1 ? epi.i1[row] : row) * 4 * W + U;
#pragma unroll
        for (int g = 0; g < 4; ++g) z[g] += t1[(long)g * W];
      }
      if (epi.T2) {
        const float* t2 = epi.T2 + (long)(epi.i2 ? epi.i2[row] : row) * 4 * W + U;
#pragma unroll
        for (int g = 0; g < 4; ++g) z[g] += t2[(long)g * W];
      }
      const float gi = sigmoidf_(z[0]), gf = sigmoidf_(z[1]), gg = tanhf_(z[2]), go = sigmoidf_(z[3]);
      const float cp = epi.c_prev[(long)epi.slot_in[row] * epi.c_ld + U];
      const float c = gf * cp + gi * gg;
      const float hv = go * tanhf_(c);
      const long o = (long)epi.slot_out[row] * epi.out_ld + U;
      epi.c_out[o] = c;
      epi.h_out[o] = hv;
      if (epi.xn) {
        bf16_t hi, lo;
        split_bf16(hv, hi, lo);
        bf16_t* x = epi.xn + (long)row * epi.ldn + U;
        x[0] = hi;
        if (epi.nbn == 3) {
          x[epi.kn] = lo;
          x[2 * epi.kn] = hi;
        }
      }
    }
    return;
  }
  if (OUT == 1 && (ldc & 3) == 0 && (n0 + LBN <= N) && ((size_t)Cv & 7) == 0) {
    // bf16 stores in whole rows (dH / dX of the second-generation backward scan): as the f32 form below, 8 bytes per lane,
    // 32 lanes = one 256-byte row segment
    constexpr int LDP = LBN + 4;
    float* ct = reinterpret_cast<float*>(smem);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < RF; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ct[(wm * WROWS + i * 16 + fq * 4 + r) * LDP + wn * WCOLS + j * 16 + fr] = acc[i][j][r] * alpha;
    __syncthreads();
    const int c4 = (tid & 31) * 4;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) bv = f32x4{bias[n0 + c4], bias[n0 + c4 + 1], bias[n0 + c4 + 2], bias[n0 + c4 + 3]};
    bf16_t* Cb = reinterpret_cast<bf16_t*>(Cv);
    constexpr int RPP = 64 * NW / 32;
#pragma unroll 4
    for (int p = 0; p < TBM / RPP; ++p) {
      const int row = p * RPP + (tid >> 5);
      if (m0 + row < M) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ct + row * LDP + c4) + bv;
        uint2 pk;
        pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
        pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
        *reinterpret_cast<uint2*>(Cb + (long)(m0 + row) * ldc + n0 + c4) = pk;
      }
    }
    return;
  }
  if (OUT == 0 && (ldc & 3) == 0 && (n0 + LBN <= N) && ((size_t)Cv & 15) == 0) {
    // f32 stores in whole rows: the accumulator layout (4 rows x 1 column per lane) would
    // write 64-byte row segments; turn the tile through LDS (the ring is free now) so that
    // 32 lanes write one 512-byte row
    constexpr int LDP = LBN + 4;                       // padded row, floats
    float* ct = reinterpret_cast<float*>(smem);
    __builtin_amdgcn_s_barrier();                      // every wave is done reading the last stage
#pragma unroll
    for (int i = 0; i < RF; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          ct[(wm * WROWS + i * 16 + fq * 4 + r) * LDP + wn * WCOLS + j * 16 + fr] = acc[i][j][r] * alpha;
    __syncthreads();
    const int c4 = (tid & 31) * 4;
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) bv = f32x4{bias[n0 + c4], bias[n0 + c4 + 1], bias[n0 + c4 + 2], bias[n0 + c4 + 3]};
    float* Cf = reinterpret_cast<float*>(Cv);
    constexpr int RPP = 64 * NW / 32;                  // rows per pass
#pragma unroll 4
    for (int p = 0; p < TBM / RPP; ++p) {
      const int row = p * RPP + (tid >> 5);
      if (m0 + row < M) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ct + row * LDP + c4);
        *reinterpret_cast<f32x4*>(Cf + (long)(m0 + row) * ldc + n0 + c4) = v + bv;
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < RF; ++i) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + wn * WCOLS + j * 16 + fr;
      if (col >= N) continue;
      const float bv = (bias != nullptr && bz == 0) ? bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * WROWS + i * 16 + fq * 4 + r;
        if (row >= M) continue;
        const float v = acc[i][j][r] * alpha + bv;
        if (OUT == 0) {
          reinterpret_cast<float*>(Cv)[(long)row * ldc + col] = v;
        } else if (OUT == 1) {
          reinterpret_cast<bf16_t*>(Cv)[(long)row * ldc + col] = f2bf(v);
        } else {
          if (c_t) atomicAdd(reinterpret_cast<float*>(Cv) + (long)col * ldc + row, v);
          else atomicAdd(reinterpret_cast<float*>(Cv) + (long)row * ldc + col, v);
        }
      }
    }
  }
}

// ---------------------------------------------------------------- persistent ring GEMM, plain f32 stores
// The many-row products of a training window (P = H K^T, dX = dZ Kn^T, logits, dH: M = T*B rows, K of a
// few hundred) have 8-32 k-steps per 256 x 128 tile: with one workgroup per tile every tile pays the ring's
// start-up latency, a 128 KiB turn of the tile through LDS and the launch of its workgroup, none of it
// overlapped (diagnostic at M = 262144, N = 2048, K = 512: 0.19 ms of 1.1 ms with neither main loop nor stores,
// 0.48 ms main loop, 0.45 ms stores).  Here a workgroup stays on its CU and walks through its tiles: the
// ring runs across tile boundaries (the next tile's first stages are in flight during the last k-steps),
// and the accumulators leave straight from registers -- a 4 x 4 transpose inside each lane quad (DPP)
// turns "4 rows x 1 column" per lane into 16 contiguous bytes of one row, so the ring's LDS is never
// borrowed and the stores (buffer stores, out-of-range lanes dropped by the bounds check: always 16 per
// wave, which the counted vmcnt below relies on) overlap the next tile's main loop.
__device__ __forceinline__ float dpp_xor1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float dpp_xor2(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
}
// lane k of a quad holds m[r] = M[k][r]; afterwards m[r] = M[r][k]
__device__ __forceinline__ void quad_transpose(f32x4& m, int k) {
  const bool odd = k & 1, hi = k & 2;
  {
    const float s0 = odd ? m[0] : m[1], s1 = odd ? m[2] : m[3];
    const float r0 = dpp_xor1(s0), r1 = dpp_xor1(s1);
    if (odd) { m[0] = r0; m[2] = r1; } else { m[1] = r0; m[3] = r1; }
  }
  {
    const float s0 = hi ? m[0] : m[2], s1 = hi ? m[1] : m[3];
    const float r0 = dpp_xor2(s0), r1 = dpp_xor2(s1);
    if (hi) { m[0] = r0; m[1] = r1; } else { m[2] = r0; m[3] = r1; }
  }
}

__global__ __launch_bounds__(512, 1) void gemm_tn_pers_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, float* __restrict__ C, const float* __restrict__ bias,
    int M, int N, int K, long lda, long ldb, long ldc, float alpha, int tiles_n, int total, int out_bf16) {
  constexpr int RF = 4, WN = 2;               // 8 waves as 4 x 2, each 64 x 64
  constexpr int WROWS = 64, TBM = 256, WCOLS = 64, NT = 4;
  constexpr int PA = 4, PB = 2, NP = 6, NP0 = 3;
  constexpr int STAGE_BYTES = (TBM + LBN) * 128;
  constexpr int NST = RF * NT;                 // stores per lane and tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int nkt = K / BK;
  // tiles of this workgroup: XCD x (= blockIdx % 8) takes a contiguous run of the n-fastest tile order in
  // every round, so the n-tiles of an m-tile (same A rows) meet in one L2
  const int per_xcd = gridDim.x >> 3;
  const int first = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int my_n = first < total ? (total - first + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  if (my_n == 0) return;
  const int G = my_n * nkt;                    // flat k-steps of this workgroup

  auto clamp31 = [](long v) { return (int)(v > 0x7fffffffL ? 0x7fffffffL : (v < 0 ? 0 : v)); };
  unsigned vo[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    const int row = j < PA ? wave * 8 * PA + j * 8 + (lane >> 3) : wave * 8 * PB + (j - PA) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    vo[j] = (unsigned)((long)row * (j < PA ? lda : ldb) * 2 + c * 16);
  }
  const unsigned lds0 = (unsigned)(size_t)(lds_void_t*)smem;
  // the bias row lives in LDS behind the ring: a global load in the epilogue would sit in the in-order
  // vmcnt queue behind the next tile's DMA
  float* bias_l = reinterpret_cast<float*>(smem + LSTAGES * STAGE_BYTES);
  if (bias != nullptr) {
    for (int c = tid; c < N; c += 512) bias_l[c] = bias[c];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // issue side: tile and k-step the next DMA belongs to
  int it_i = 0, kt_i = 0;
  __amdgpu_buffer_rsrc_t rsA, rsB;
  auto set_issue_tile = [&](int it) {
    const int L = first + it * (int)gridDim.x;
    const int m0 = (L / tiles_n) * TBM, n0 = (L % tiles_n) * LBN;
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A + (long)m0 * lda), 0, clamp31(((long)(M - m0 - 1) * lda + K) * 2), 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B + (long)n0 * ldb), 0, clamp31(((long)(N - n0 - 1) * ldb + K) * 2), 0x00020000);
  };
  set_issue_tile(0);
  auto issue_half = [&](int stage, int half) {
    const int soff = kt_i * BK * 2;
    const unsigned sa = lds0 + stage * STAGE_BYTES + wave * 8 * PA * 128;
    const unsigned sb = lds0 + stage * STAGE_BYTES + TBM * 128 + wave * 8 * PB * 128;
#pragma unroll
    for (int j = half ? NP0 : 0; j < (half ? NP : NP0); ++j) {
      if (j < PA) glds16(rsA, vo[j], soff, sa + j * 1024);
      else glds16(rsB, vo[j], soff, sb + (j - PA) * 1024);
    }
    if (half) {     // the step is issued: advance
      if (++kt_i == nkt) {
        kt_i = 0;
        if (++it_i < my_n) set_issue_tile(it_i);
      }
    }
  };

  f32x4 acc[RF][NT];
#pragma unroll
  for (int i = 0; i < RF; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue_half(0, 0);
  issue_half(0, 1);
  if (G > 1) {
    issue_half(1, 0);
    issue_half(1, 1);
  }
  const int fr = lane & 15, fq = lane >> 4;
  int stage = 0, it_c = 0, kt_c = 0;
  // A tile that ends in step e issues its NST stores behind the DMA of steps e+1 and e+2 (both already in
  // flight), so the waits of those two steps may leave them outstanding: vmcnt is in order and only counts.
  int stored = 0;
  for (int g = 0; g < G; ++g) {
    if (g + 1 < G) {
      if (stored) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP + NST) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    } else {
      if (stored) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (stored) --stored;
    __builtin_amdgcn_s_barrier();
    const bool more = g + 2 < G;
    const int nstage = stage >= 1 ? stage - 1 : LSTAGES - 1;
    const unsigned char* a_base = smem + stage * STAGE_BYTES;
    const unsigned char* b_base = a_base + TBM * 128;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      frag16 fa[RF], fb[NT];
#pragma unroll
      for (int i = 0; i < RF; ++i)
        fa[i].u = *reinterpret_cast<const uint4*>(a_base + lds_off(wm * WROWS + i * 16 + fr, s * 4 + fq));
#pragma unroll
      for (int j = 0; j < NT; ++j)
        fb[j].u = *reinterpret_cast<const uint4*>(b_base + lds_off(wn * WCOLS + j * 16 + fr, s * 4 + fq));
      if (more) issue_half(nstage, s);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < RF; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = mfma16(fa[i].v, fb[j].v, acc[i][j]);
      __builtin_amdgcn_s_setprio(0);
    }
    stage = stage + 1 < LSTAGES ? stage + 1 : 0;
    if (++kt_c == nkt) {
      // the tile is complete: out it goes, straight from the accumulators
      const int L = first + it_c * (int)gridDim.x;
      const int m0 = (L / tiles_n) * TBM, n0 = (L % tiles_n) * LBN;
      // (out_bf16: C is a bf16 matrix -- the gate-input rows P of the second-generation wide scans: half the bytes)
      const int esz = out_bf16 ? 2 : 4;
      const long bytes = ((long)(M - m0 - 1) * ldc + (N - n0)) * esz;
      const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(
          reinterpret_cast<char*>(C) + ((long)m0 * ldc + n0) * esz, 0, (int)(unsigned)(bytes > 0xffffffffL ? 0xffffffffL : bytes), 0x00020000);
      const int k = fr & 3, c4 = fr & ~3;
#pragma unroll
      for (int i = 0; i < RF; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          f32x4 v = acc[i][j];
          quad_transpose(v, k);
          const int row = wm * WROWS + i * 16 + fq * 4 + k;
          const int col = wn * WCOLS + j * 16 + c4;
          f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
          if (bias != nullptr && n0 + col + 3 < N) bv = *reinterpret_cast<const f32x4*>(bias_l + n0 + col);
          v = v * alpha + bv;
          // (a column past N would land in the next row's bytes: push it out of the buffer instead)
          const unsigned off = (n0 + col < N) ? (unsigned)(((long)row * ldc + col) * esz) : 0xfffffff0u;
          if (out_bf16) {
            typedef __attribute__((ext_vector_type(2))) unsigned int u32x2g;
            const u32x2g pk = u32x2g{(unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16), (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16)};
            __builtin_amdgcn_raw_buffer_store_b64(pk, rsC, (int)off, 0, 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4g, v), rsC, (int)off, 0, 0);
          }
          acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      kt_c = 0;
      ++it_c;
      stored = 2;
    }
  }
}

template <int RF, int WM, int WN>
int launch_long_t(int out_mode, dim3 grid, hipStream_t stream, const bf16_t* A, const bf16_t* B, void* C, const float* bias,
                  int M, int N, int K, long lda, long ldb, long ldc, int k_per_split, float alpha,
                  const KlGateEpi* gate = nullptr) {
  KlGateEpi epi;
  if (gate) epi = *gate;
  else memset(&epi, 0, sizeof(epi));
  const size_t lds = (size_t)LSTAGES * (16 * RF * WM + LBN) * 128;
  static const bool ilv = !(getenv("KL_GEMM_ILV") && getenv("KL_GEMM_ILV")[0] == '0');
  static const int remap = !(getenv("KL_GEMM_XCD") && getenv("KL_GEMM_XCD")[0] == '0');
#define KL_LONG_CASE(O)                                                                                                  \
  do {                                                                                                                   \
    static KlLdsGrant grant_a, grant_b;      /* (per device: kl_kernels.h) */                                            \
    if (kl_grant_lds(grant_a, reinterpret_cast<const void*>(&gemm_tn_long_kernel<O, true, RF, WM, WN>), lds)) return KL_ERR_LAUNCH;  \
    if (kl_grant_lds(grant_b, reinterpret_cast<const void*>(&gemm_tn_long_kernel<O, false, RF, WM, WN>), lds)) return KL_ERR_LAUNCH; \
    if (ilv) hipLaunchKernelGGL((gemm_tn_long_kernel<O, true, RF, WM, WN>), grid, dim3(64 * WM * WN), lds, stream, A, B, C,  \
                                bias, M, N, K, lda, ldb, ldc, k_per_split, alpha, epi, remap, 0, KlGemmSecond{});                        \
    else hipLaunchKernelGGL((gemm_tn_long_kernel<O, false, RF, WM, WN>), grid, dim3(64 * WM * WN), lds, stream, A, B, C,     \
                            bias, M, N, K, lda, ldb, ldc, k_per_split, alpha, epi, remap, 0, KlGemmSecond{});                            \
  } while (0)
  if (out_mode == 0) KL_LONG_CASE(0);
  else if (out_mode == 1) KL_LONG_CASE(1);
  else if (out_mode == 3) KL_LONG_CASE(3);
  else KL_LONG_CASE(2);
#undef KL_LONG_CASE
  return 0;
}

int launch_long(int out_mode, dim3 grid, hipStream_t stream, const bf16_t* A, const bf16_t* B, void* C, const float* bias,
                int M, int N, int K, long lda, long ldb, long ldc, int k_per_split, float alpha) {
  return launch_long_t<4, 4, 2>(out_mode, grid, stream, A, B, C, bias, M, N, K, lda, ldb, ldc, k_per_split, alpha);
}

}  // namespace

// out_mode: 0 f32 store, 1 bf16 store, 2 f32 atomic accumulate (split-K allowed)
int kl_launch_gemm_tn(const bf16_t* A, const bf16_t* B, void* C, const float* bias, int M, int N, int K,
                      long lda, long ldb, long ldc, int out_mode, int splits, float alpha, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if ((K & 7) || (lda & 7) || (ldb & 7)) return KL_ERR_SHAPE;
  if (splits < 1) splits = 1;
  if (out_mode != 2) splits = 1;
  // long-K shapes (weight gradients): one 256 x 128 tile per CU, K split to fill the chip
  static const int long_mode = getenv("KL_GEMM_LONG") ? atoi(getenv("KL_GEMM_LONG")) : 2;   // 0 off, 1 split-K shapes only, 2 all
  const bool long_ok = (K % BK) == 0 && lda < (1L << 22) && ldb < (1L << 22);
  if (long_mode >= 1 && long_ok && out_mode == 2 && K >= 8192) {
    const int tiles = ((M + LBM - 1) / LBM) * ((N + LBN - 1) / LBN);
    int sp = (256 + tiles - 1) / tiles;          // ~ one workgroup per CU
    const int nk = K / BK;
    if (sp > nk / 16) sp = nk / 16;              // at least 16 k-steps per workgroup
    if (sp < 1) sp = 1;
    const int kps = ((nk + sp - 1) / sp) * BK;
    sp = (K + kps - 1) / kps;
    dim3 grid((N + LBN - 1) / LBN, (M + LBM - 1) / LBM, sp);
    const int e = launch_long(out_mode, grid, stream, A, B, C, bias, M, N, K, lda, ldb, ldc, kps, alpha);
    if (e != 0) return e;
    return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
  }
  // many-row shapes (activations x weights): the same ring, one tile per workgroup, no split
  static const bool small_all = getenv("KL_GEMM_SMALL_ALL") && getenv("KL_GEMM_SMALL_ALL")[0] == '1';   // experiment
  static const bool pers = !(getenv("KL_GEMM_PERS") && getenv("KL_GEMM_PERS")[0] == '0');
  // (measured at M = 262144: K = 512 -- P, logits -- 8-10 % faster than one workgroup per tile; K = 2048 and K = 256 not)
  if (pers && !small_all && long_mode >= 2 && long_ok && (out_mode == 0 || out_mode == 1) && K >= 6 * BK && K <= 16 * BK && (N & 3) == 0 && (ldc & 3) == 0 &&
      ((size_t)C & 15) == 0 && (bias == nullptr || N <= 2048) &&
      (long)((M + LBM - 1) / LBM) * ((N + LBN - 1) / LBN) >= 512) {
    // many-row shapes with several tiles per CU: persistent workgroups (see gemm_tn_pers_kernel)
    static int n_wg = 0;
    if (!n_wg) {
      int dev = 0;
      hipDeviceProp_t prop;
      n_wg = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8)
                 ? (prop.multiProcessorCount & ~7) : 256;
    }
    const int tiles_n = (N + LBN - 1) / LBN, total = ((M + LBM - 1) / LBM) * tiles_n;
    const size_t lds = (size_t)LSTAGES * (LBM + LBN) * 128 + (bias ? (size_t)N * 4 : 0);
    static KlLdsGrant grant;
    if (kl_grant_lds(grant, reinterpret_cast<const void*>(&gemm_tn_pers_kernel), (size_t)LSTAGES * (LBM + LBN) * 128 + 2048 * 4)) return KL_ERR_LAUNCH;
    hipLaunchKernelGGL(gemm_tn_pers_kernel, dim3(n_wg), dim3(512), lds, stream, A, B, (float*)C, bias, M, N, K, lda, ldb, ldc,
                       alpha, tiles_n, total, out_mode == 1 ? 1 : 0);
    return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
  }
  if (!small_all && long_mode >= 2 && long_ok && splits == 1 && K >= 256 && (long)((M + LBM - 1) / LBM) * ((N + LBN - 1) / LBN) >= 256) {
    dim3 grid((N + LBN - 1) / LBN, (M + LBM - 1) / LBM, 1);
    const int e = launch_long(out_mode, grid, stream, A, B, C, bias, M, N, K, lda, ldb, ldc, K, alpha);
    if (e != 0) return e;
    return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
  }
  // M ~ 1e3 rows (the incremental step's GEMMs): 64 x 128 tiles of the same ring
  if (long_mode >= 2 && long_ok && splits == 1 && K >= 512 && M >= 256 &&
      (long)((M + 63) / 64) * ((N + LBN - 1) / LBN) >= 64) {
    dim3 grid((N + LBN - 1) / LBN, (M + 63) / 64, 1);
    const int e = launch_long_t<2, 2, 4>(out_mode, grid, stream, A, B, C, bias, M, N, K, lda, ldb, ldc, K, alpha);
    if (e != 0) return e;
    return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
  }
  int k_per_split = (K + splits - 1) / splits;
  k_per_split = ((k_per_split + BK - 1) / BK) * BK;
  splits = (K + k_per_split - 1) / k_per_split;
  const int nbx = (N + BN - 1) / BN;
  // 64-row tiles when 128-row tiles cannot give every CU a workgroup
  const bool small = (long)nbx * ((M + 127) / 128) * splits < 256 && M > 64;
  if (small) {
    dim3 grid(nbx, (M + 63) / 64, splits);
    launch_bm<64>(out_mode, grid, stream, A, B, C, bias, M, N, K, lda, ldb, ldc, k_per_split, alpha);
  } else {
    dim3 grid(nbx, (M + 127) / 128, splits);
    launch_bm<128>(out_mode, grid, stream, A, B, C, bias, M, N, K, lda, ldb, ldc, k_per_split, alpha);
  }
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

// C (+)= A^T-view . B^T for a K-major A: A_km [K][M] (row stride lda_km), B [N][K] or -- b_km -- [K][N] (row stride ldb), both
// bf16; C f32 accumulated with atomics over the K splits, as C[m][n] (c_transposed 0) or C[n][m] (1), row
// stride ldc.  This is the weight-gradient contraction over the T*B rows taken straight from the backward
// scan's row-major dZ.  KL_ERR_SHAPE = not applicable (the caller transposes and uses kl_launch_gemm_tn).
// shapes kl_launch_gemm_an serves (N, ldb: any the tn kernel takes)
// K split of the K-major products: about one workgroup per CU, at least 16 k-steps per workgroup, and -- the kernels
// address one split with 32-bit byte offsets -- more splits where a split's rows would not fit in 2 GiB (the paired
// launches have twice the column tiles, so half the splits and twice the rows per split: depth 4 / width 1024 /
// T*B = 262144 fits as a single product and did not as a pair).  Returns the rows per split, 0 = not applicable.
static int an_rows_per_split(int M, int n_all, int K, long ld_max) {
  if (M <= 0 || n_all <= 0 || K <= 0 || (M % LBM) || (K % BK)) return 0;
  const int tiles = (M / LBM) * ((n_all + LBN - 1) / LBN);
  int sp = (256 + tiles - 1) / tiles;          // ~ one workgroup per CU
  const int nk = K / BK;
  if (sp > nk / 16) sp = nk / 16;              // at least 16 k-steps per workgroup
  if (sp < 1) sp = 1;
  int kps = ((nk + sp - 1) / sp) * BK;
  while ((long)kps * ld_max * 2 >= 0x7fffffffL && kps > BK) {      // 32-bit offsets inside one split
    ++sp;
    kps = ((nk + sp - 1) / sp) * BK;
  }
  return (long)kps * ld_max * 2 < 0x7fffffffL ? kps : 0;
}

bool kl_gemm_an_applicable(int M, int N, int K, long lda_km) {
  if (lda_km & 7) return false;
  return an_rows_per_split(M, N, K, lda_km) > 0;
}

int kl_launch_gemm_an2(const bf16_t* A_km, const bf16_t* B, float* C, int M, int N, int K, long lda_km, long ldb, long ldc,
                       int c_transposed, const bf16_t* B2, float* C2, int N2, long ldb2, long ldc2, int c_transposed2,
                       hipStream_t stream, int b_km) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (B2 != nullptr && ((N % LBN) || N2 <= 0 || (ldb2 & 7) || (!b_km && ldb2 >= (1L << 22)))) return KL_ERR_SHAPE;
  const int n_all = B2 != nullptr ? N + N2 : N;
  if ((lda_km & 7) || (ldb & 7) || (!b_km && ldb >= (1L << 22))) return KL_ERR_SHAPE;
  long ld_max = lda_km;
  if (b_km && ldb > ld_max) ld_max = ldb;
  if (b_km && B2 != nullptr && ldb2 > ld_max) ld_max = ldb2;
  const int kps = an_rows_per_split(M, n_all, K, ld_max);
  if (kps <= 0) return KL_ERR_SHAPE;
  const int sp = (K + kps - 1) / kps;
  dim3 grid((n_all + LBN - 1) / LBN, M / LBM, sp);
  KlGateEpi epi;
  memset(&epi, 0, sizeof(epi));
  KlGemmSecond second;
  memset(&second, 0, sizeof(second));
  if (B2 != nullptr) {
    second.B = B2;
    second.C = C2;
    second.n_first = N;
    second.N = N2;
    second.ldb = ldb2;
    second.ldc = ldc2;
    second.c_transposed = c_transposed2;
  }
  const size_t lds = (size_t)LSTAGES * (LBM + LBN) * 128;
  static const int remap = !(getenv("KL_GEMM_XCD") && getenv("KL_GEMM_XCD")[0] == '0');
  static KlLdsGrant grant_a, grant_b;
  if (kl_grant_lds(grant_a, reinterpret_cast<const void*>(&gemm_tn_long_kernel<2, true, 4, 4, 2, true, false>), lds)) return KL_ERR_LAUNCH;
  if (kl_grant_lds(grant_b, reinterpret_cast<const void*>(&gemm_tn_long_kernel<2, true, 4, 4, 2, true, true>), lds)) return KL_ERR_LAUNCH;
  if (b_km)
    hipLaunchKernelGGL((gemm_tn_long_kernel<2, true, 4, 4, 2, true, true>), grid, dim3(512), lds, stream, A_km, B, (void*)C,
                       (const float*)nullptr, M, N, K, lda_km, ldb, ldc, kps, 1.f, epi, remap, c_transposed, second);
  else
    hipLaunchKernelGGL((gemm_tn_long_kernel<2, true, 4, 4, 2, true, false>), grid, dim3(512), lds, stream, A_km, B, (void*)C,
                       (const float*)nullptr, M, N, K, lda_km, ldb, ldc, kps, 1.f, epi, remap, c_transposed, second);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

int kl_launch_gemm_an(const bf16_t* A_km, const bf16_t* B, float* C, int M, int N, int K, long lda_km, long ldb, long ldc,
                      int c_transposed, hipStream_t stream, int b_km) {
  return kl_launch_gemm_an2(A_km, B, C, M, N, K, lda_km, ldb, ldc, c_transposed, nullptr, nullptr, 0, 0, 0, 0, stream, b_km);
}

// z = A3 . WTperm^T with the LSTM cell as epilogue (see OUT == 3 above).  A3 [n][lda] bf16 rows,
// WTperm [4W][lda] with rows in (unit block of 32, gate, unit) order, Kc = contracted length.
// KL_ERR_SHAPE = not applicable (caller uses the unfused path).
int kl_launch_gemm_gates(const bf16_t* A3, const bf16_t* WTperm, int n, int W, int Kc, long lda, const KlGateEpi* epi,
                         hipStream_t stream) {
  if (n < 1 || (W & 31) || (Kc % BK) != 0 || (lda & 7) || lda >= (1L << 22) || !epi) return KL_ERR_SHAPE;
  dim3 grid(4 * W / LBN, (n + 63) / 64, 1);
  const int e = launch_long_t<2, 2, 4>(3, grid, stream, A3, WTperm, nullptr, nullptr, n, 4 * W, Kc, lda, lda, 0, Kc, 1.f, epi);
  if (e != 0) return e;
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}
