// Incremental step for MANY hypotheses (n >= 256): the same cell as lstm_step.hip, but
// organised as big-tile GEMMs so that the state rows and weight slices are staged
// once per 128x128 tile instead of once per thin workgroup.
//
// Split precision without a special GEMM: with a = a_hi + a_lo and w = w_hi + w_lo,
//   a.w ~= a_hi.w_hi + a_lo.w_hi + a_hi.w_lo = [a_hi | a_lo | a_hi] . [w_hi | w_hi | w_lo]
// i.e. ONE bf16 GEMM over a 3x longer contraction.  The weight side ([4W][3K], built in
// kl_prepare) is constant; the activation side is produced by the gather kernel below,
// which also resolves the pool-slot indirection of Rater.predict (rating.py:622-629).
#include "kl_common.h"
#include "kl_kernels.h"

namespace {

// out[row] = [hi(s0) hi(s1) | lo(s0) lo(s1) | hi(s0) hi(s1)]   (nb = 3)  or  [hi(s0) hi(s1)]  (nb = 1)
__global__ void split_gather_kernel(const float* __restrict__ s0, long ld0, const int* __restrict__ i0, int w0,
                                    const float* __restrict__ s1, long ld1, const int* __restrict__ i1, int w1, int n,
                                    int nb, bf16_t* __restrict__ out, long ld_out) {
  const int kt = w0 + w1;
  const int chunks = kt >> 3;
  const long total = (long)n * chunks;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % chunks);
    const int row = (int)(e / chunks);
    const int k = c * 8;
    const float* src = k < w0 ? s0 + (long)(i0 ? i0[row] : row) * ld0 + k : s1 + (long)(i1 ? i1[row] : row) * ld1 + (k - w0);
    const float4 x0 = *reinterpret_cast<const float4*>(src);
    const float4 x1 = *reinterpret_cast<const float4*>(src + 4);
    const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    frag16 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) split_bf16(xs[j], hi.s[j], lo.s[j]);
    bf16_t* o = out + (long)row * ld_out + k;
    *reinterpret_cast<uint4*>(o) = hi.u;
    if (nb == 3) {
      *reinterpret_cast<uint4*>(o + kt) = lo.u;
      *reinterpret_cast<uint4*>(o + 2 * kt) = hi.u;
    }
  }
}

// gates on precomputed z: thread = (row, unit)
__global__ void gates_rows_kernel(const float* __restrict__ z, long ldz, int n, int W, const float* __restrict__ T1,
                                  const int* __restrict__ i1, const float* __restrict__ T2, const int* __restrict__ i2,
                                  const float* __restrict__ bias, const float* __restrict__ c_prev, long c_ld,
                                  const int* __restrict__ slot_in, float* __restrict__ c_out, float* __restrict__ h_out,
                                  long out_ld, const int* __restrict__ slot_out) {
  const long total = (long)n * W;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int u = (int)(e % W);
    const int row = (int)(e / W);
    float zz[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const long col = (long)g * W + u;
      float v = z[(long)row * ldz + col];
      if (T1) v += T1[(long)(i1 ? i1[row] : row) * 4 * W + col];
      if (T2) v += T2[(long)(i2 ? i2[row] : row) * 4 * W + col];
      if (bias) v += bias[col];
      zz[g] = v;
    }
    const float gi = sigmoidf_(zz[0]), gf = sigmoidf_(zz[1]), gg = tanhf_(zz[2]), go = sigmoidf_(zz[3]);
    const float cp = c_prev[(long)slot_in[row] * c_ld + u];
    const float c = gf * cp + gi * gg;
    const long o = (long)slot_out[row] * out_ld + u;
    c_out[o] = c;
    h_out[o] = go * tanhf_(c);
  }
}

// recurrent halves of every layer's activation rows in one launch: layer l (blockIdx.y) takes
// h_l of slot_in[row] as (hi | lo | hi) blocks of stride K_l at column offset (l > 0 ? W : 0)
__global__ void gather_recurrent_kernel(const KlGatherRec a) {
  const int l = blockIdx.y;
  const int W = a.W, chunks = W >> 3;
  const long total = (long)a.n * chunks;
  const long kl = l == 0 ? W : 2L * W;
  bf16_t* out = a.out[l] + (l == 0 ? 0 : W);
  const long ld_out = 3 * kl;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % chunks);
    const int row = (int)(e / chunks);
    const float* src = a.pool + (long)a.slot_in[row] * a.slot_ld + (long)2 * l * W + c * 8;
    const float4 x0 = *reinterpret_cast<const float4*>(src);
    const float4 x1 = *reinterpret_cast<const float4*>(src + 4);
    const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    frag16 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) split_bf16(xs[j], hi.s[j], lo.s[j]);
    bf16_t* o = out + (long)row * ld_out + c * 8;
    *reinterpret_cast<uint4*>(o) = hi.u;
    if (a.nb == 3) {
      *reinterpret_cast<uint4*>(o + kl) = lo.u;
      *reinterpret_cast<uint4*>(o + 2 * kl) = hi.u;
    }
  }
}

// dst row (u / 32) * 128 + g * 32 + u % 32  <-  src row g * W + u   (rows of `cols` bf16)
__global__ void permute_gate_rows_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int W, long cols) {
  const long chunks = cols >> 3;
  const long total = 4L * W * chunks;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const long c = e % chunks;
    const int r = (int)(e / chunks);
    const int g = r / W, u = r % W;
    const long d = (long)(u >> 5) * 128 + g * 32 + (u & 31);
    *reinterpret_cast<uint4*>(dst + d * cols + c * 8) = *reinterpret_cast<const uint4*>(src + (long)r * cols + c * 8);
  }
}

inline int grid_for(long total, int block) {
  long g = (total + block - 1) / block;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  return (int)g;
}
inline int ok() { return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH; }

}  // namespace

int kl_launch_split_gather(const float* s0, long ld0, const int* i0, int w0, const float* s1, long ld1, const int* i1,
                           int w1, int n, int nb, bf16_t* out, long ld_out, hipStream_t stream) {
  if ((w0 & 7) || (w1 & 7) || (ld_out & 7) || (nb != 1 && nb != 3)) return KL_ERR_SHAPE;
  const long total = (long)n * ((w0 + w1) >> 3);
  hipLaunchKernelGGL(split_gather_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, s0, ld0, i0, w0, s1 ? s1 : s0,
                     ld1, i1, w1, n, nb, out, ld_out);
  return ok();
}

int kl_launch_gates_rows(const float* z, long ldz, int n, int W, const float* T1, const int* i1, const float* T2,
                         const int* i2, const float* bias, const float* c_prev, long c_ld, const int* slot_in,
                         float* c_out, float* h_out, long out_ld, const int* slot_out, hipStream_t stream) {
  hipLaunchKernelGGL(gates_rows_kernel, dim3(grid_for((long)n * W, 256)), dim3(256), 0, stream, z, ldz, n, W, T1, i1, T2,
                     i2, bias, c_prev, c_ld, slot_in, c_out, h_out, out_ld, slot_out);
  return ok();
}

int kl_launch_gather_recurrent(const KlGatherRec& a, int L, hipStream_t stream) {
  if ((a.W & 7) || L < 1 || L > KL_SCAN_MAXL || (a.nb != 1 && a.nb != 3)) return KL_ERR_SHAPE;
  const long total = (long)a.n * (a.W >> 3);
  hipLaunchKernelGGL(gather_recurrent_kernel, dim3(grid_for(total, 256), L), dim3(256), 0, stream, a);
  return ok();
}

int kl_launch_permute_gate_rows(const bf16_t* src, bf16_t* dst, int W, long cols, hipStream_t stream) {
  if ((W & 31) || (cols & 7)) return KL_ERR_SHAPE;
  hipLaunchKernelGGL(permute_gate_rows_kernel, dim3(grid_for(4L * W * (cols >> 3), 256)), dim3(256), 0, stream, src, dst, W, cols);
  return ok();
}
