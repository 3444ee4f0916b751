// Big-tile bf16 MFMA GEMM for the non-recurrent contractions of training
// (F2 input contraction, F5 logits, B* weight/input gradients).
//
//   C[M,N] (op)= A[M,K] . B[N,K]^T (+ bias[N])
//
// Both operands are K-contiguous bf16 ("TN"): every producer in this library
// writes the layout its consumer contracts over, so one kernel serves all.
// Tile 128x128x64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA
// 16x16x32 tiles (64 accumulator VGPRs).  LDS: 2 stages x (A 16 KiB + B 16 KiB),
// 128-byte rows XOR-swizzled (chunk ^= (row>>1)&7) so that the ds_read_b128
// fragment reads are bank-conflict free.  Register-staged double buffering:
// the global loads of k-tile t+1 are in flight while k-tile t is in the MFMAs;
// one barrier per k-tile.
// Split-K over gridDim.z with f32 atomics for the K = B*T weight-gradient
// shapes (M,N small, K huge).
#include "kl_common.h"
#include "kl_kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int NTHREADS = 256;

__device__ __forceinline__ int lds_off(int row, int chunk) {
  // byte offset of 16-byte chunk `chunk` (0..7) of row `row` in a [rows][64] bf16 tile
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <int OUT>   // 0 = f32 store, 1 = bf16 store, 2 = f32 atomic add
__global__ __launch_bounds__(NTHREADS) void gemm_tn_kernel(
    const bf16_t* __restrict__ A, const bf16_t* __restrict__ B, void* __restrict__ Cv,
    const float* __restrict__ bias, int M, int N, int K, long lda, long ldb, long ldc,
    int k_per_split, float alpha) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sA = smem;                 // 2 x 16 KiB
  unsigned char* sB = smem + 2 * BM * 128;  // 2 x 16 KiB

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.y * BM;
  const int n0 = blockIdx.x * BN;
  const int kbeg = blockIdx.z * k_per_split;
  const int kend = min(K, kbeg + k_per_split);
  const int nkt = (kend - kbeg + BK - 1) / BK;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 ra[4], rb[4];
  auto gload = [&](int kt) {
    const int k0 = kbeg + kt * BK;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = tid + NTHREADS * j;
      const int row = q >> 3, c = q & 7;
      const int k = k0 + c * 8;
      const bool kok = k < kend;
      const int ar = m0 + row, br = n0 + row;
      ra[j] = (kok && ar < M) ? *reinterpret_cast<const uint4*>(A + (long)ar * lda + k) : uint4{0, 0, 0, 0};
      rb[j] = (kok && br < N) ? *reinterpret_cast<const uint4*>(B + (long)br * ldb + k) : uint4{0, 0, 0, 0};
    }
  };
  auto sstore = [&](int stage) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int q = tid + NTHREADS * j;
      const int row = q >> 3, c = q & 7;
      *reinterpret_cast<uint4*>(sA + stage * BM * 128 + lds_off(row, c)) = ra[j];
      *reinterpret_cast<uint4*>(sB + stage * BN * 128 + lds_off(row, c)) = rb[j];
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
      frag16 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        fa[i].u = *reinterpret_cast<const uint4*>(a_base + lds_off(wm * 64 + i * 16 + fr, s * 4 + fq));
        fb[i].u = *reinterpret_cast<const uint4*>(b_base + lds_off(wn * 64 + i * 16 + fr, s * 4 + fq));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16(fa[i].v, fb[j].v, acc[i][j]);
    }
    if (kt + 1 < nkt) sstore(stage ^ 1);
    __syncthreads();
  }

  // epilogue: D col = lane&15, row = 4*(lane>>4)+r
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = n0 + wn * 64 + j * 16 + fr;
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
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, splits);
  const size_t lds = 2 * (BM + BN) * 128;
  switch (out_mode) {
    case 0:
      hipLaunchKernelGGL(gemm_tn_kernel<0>, grid, dim3(NTHREADS), lds, stream, A, B, C, bias, M, N, K, lda, ldb, ldc, k_per_split, alpha);
      break;
    case 1:
      hipLaunchKernelGGL(gemm_tn_kernel<1>, grid, dim3(NTHREADS), lds, stream, A, B, C, bias, M, N, K, lda, ldb, ldc, k_per_split, alpha);
      break;
    default:
      hipLaunchKernelGGL(gemm_tn_kernel<2>, grid, dim3(NTHREADS), lds, stream, A, B, C, bias, M, N, K, lda, ldb, ldc, k_per_split, alpha);
      break;
  }
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}
