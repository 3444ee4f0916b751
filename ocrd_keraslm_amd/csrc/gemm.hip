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

}  // namespace

// out_mode: 0 f32 store, 1 bf16 store, 2 f32 atomic accumulate (split-K allowed)
int kl_launch_gemm_tn(const bf16_t* A, const bf16_t* B, void* C, const float* bias, int M, int N, int K,
                      long lda, long ldb, long ldc, int out_mode, int splits, float alpha, hipStream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if ((K & 7) || (lda & 7) || (ldb & 7)) return KL_ERR_SHAPE;
  if (splits < 1) splits = 1;
  if (out_mode != 2) splits = 1;
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
