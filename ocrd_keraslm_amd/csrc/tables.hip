// Layer-0 input contraction as table look-ups.
//
// The first LSTM layer's input is one-hot structured: x_t = [E[idx] | Ctx_n[ctx_n]],
// so x_t . K0 = EK[idx] + sum_n CtxK_n[ctx_n] with EK = E . K0[:W] ([V,4W]) and
// CtxK_n = Ctx_n . K0[W+10n : W+10n+10] ([200,4W]).  The tables cost V*W*4W MACs
// once per weight update instead of B*T*(W+10C)*4W per window, and the backward
// of the look-up is a segment sum (one-hot^T . dZ) -- see api.hip.
// This file: the context tables (K-dim 10 is too short for MFMA: plain f32 FMA),
// the row gather that materialises P1, and the f32 context-gradient kernels.
#include "kl_common.h"
#include "kl_kernels.h"

namespace {

// C[r][n] = sum_k A[r][k] * Kmat[k][n]   (A [R][D] f32, Kmat [D][N] f32 row-major)
__global__ void small_table_kernel(const float* __restrict__ A, int R, int D, const float* __restrict__ Kmat,
                                   long ldk, int N, float* __restrict__ C, long ldc) {
  const long total = (long)R * N;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int r = (int)(e / N), n = (int)(e % N);
    float a = 0.f;
    for (int k = 0; k < D; ++k) a = fmaf(A[(long)r * D + k], Kmat[(long)k * ldk + n], a);
    C[(long)r * ldc + n] = a;
  }
}

struct Tabs { const float* t[8]; };

// P[t*B+b][:] = EK[idx[b,t]] + sum_n CtxK_n[ctx[b,t,n]] + bias   (float4 per thread)
__global__ void p1_gather_kernel(const float* __restrict__ EK, Tabs ctxk, int n_ctx, const float* __restrict__ bias,
                                 const int* __restrict__ idx, const int* __restrict__ ctx, int B, int T, int N4,
                                 float* __restrict__ P) {
  const long total = (long)B * T * N4;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % N4);
    const long row = e / N4;
    const int b = (int)(row % B), t = (int)(row / B);
    const long src = (long)b * T + t;
    float4 v = reinterpret_cast<const float4*>(EK)[(long)idx[src] * N4 + c];
    const float4 bb = reinterpret_cast<const float4*>(bias)[c];
    v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
    for (int n = 0; n < n_ctx; ++n) {
      const float4 q = reinterpret_cast<const float4*>(ctxk.t[n])[(long)ctx[src * n_ctx + n] * N4 + c];
      v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    reinterpret_cast<float4*>(P)[row * N4 + c] = v;
  }
}

// out[col] += sum_r in[r][col]   (bias gradient; grid.x over 64-col groups, grid.y over row chunks)
__global__ void colsum_bf16_kernel(const bf16_t* __restrict__ in, long ld, int rows, int cols, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + tx;
  const int rchunk = (rows + gridDim.y - 1) / gridDim.y;
  const int rb = blockIdx.y * rchunk, re = min(rows, rb + rchunk);
  float a = 0.f;
  if (col < cols)
    for (int r = rb + ty; r < re; r += 4) a += bf2f(in[(long)r * ld + col]);
  red[ty][tx] = a;
  __syncthreads();
  if (ty == 0 && col < cols) atomicAdd(out + col, red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]);
}

// Hbuf/Cbuf last block -> carried state (slot layout [B][2L][W] f32)
__global__ void rows_to_state_kernel(const void* __restrict__ h_rows, int h_is_f32, const float* __restrict__ c_rows,
                                     int B, int W, int L, int layer, float* __restrict__ states) {
  const long total = (long)B * W;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int b = (int)(e / W), u = (int)(e % W);
    float* s = states + ((long)b * 2 * L + 2 * layer) * W;
    s[u] = h_is_f32 ? reinterpret_cast<const float*>(h_rows)[e] : bf2f(reinterpret_cast<const bf16_t*>(h_rows)[e]);
    s[W + u] = c_rows[e];
  }
}

// context-table gradients from dCtxKT [4W][ldt] f32 (transposed segment sums):
//   gK[d][col]  += sum_r Ctx[r][d] * dCtxKT[col][r]          (rows W+10n.. of K0)
//   gCtx[r][d]  += sum_col dCtxKT[col][r] * K0[W+10n+d][col]
__global__ void ctx_grad_k_kernel(const float* __restrict__ Ctx, int R, int D, const float* __restrict__ dT, long ldt,
                                  int N, float* __restrict__ gK, long ldg) {
  const long total = (long)D * N;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int d = (int)(e / N), col = (int)(e % N);
    float a = 0.f;
    for (int r = 0; r < R; ++r) a = fmaf(Ctx[(long)r * D + d], dT[(long)col * ldt + r], a);
    gK[(long)d * ldg + col] += a;
  }
}
__global__ void ctx_grad_c_kernel(const float* __restrict__ Krows, long ldk, int R, int D, const float* __restrict__ dT,
                                  long ldt, int N, float* __restrict__ gCtx) {
  // one wave per (r,d)
  const int lane = threadIdx.x & 63;
  const long item = blockIdx.x * (long)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (item >= (long)R * D) return;
  const int r = (int)(item / D), d = (int)(item % D);
  float a = 0.f;
  for (int col = lane; col < N; col += 64) a = fmaf(dT[(long)col * ldt + r], Krows[(long)d * ldk + col], a);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
  if (lane == 0) gCtx[(long)r * D + d] += a;
}

// probs rows time-major [T*B][V] -> batch-major [B][T][V]
__global__ void rows_tm_to_bm_kernel(const float* __restrict__ in, long ld_in, float* __restrict__ out, int B, int T,
                                     int V) {
  const long total = (long)B * T * V;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int v = (int)(e % V);
    const long row = e / V;   // batch-major b*T + t
    const int b = (int)(row / T), t = (int)(row % T);
    out[e] = in[((long)t * B + b) * ld_in + v];
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

int kl_launch_small_table(const float* A, int R, int D, const float* Kmat, long ldk, int N, float* C, long ldc,
                          hipStream_t stream) {
  hipLaunchKernelGGL(small_table_kernel, dim3(grid_for((long)R * N, 256)), dim3(256), 0, stream, A, R, D, Kmat, ldk, N,
                     C, ldc);
  return ok();
}

int kl_launch_p1_gather(const float* EK, const float* const* ctxk, int n_ctx, const float* bias, const int* idx,
                        const int* ctx, int B, int T, int N, float* P, hipStream_t stream) {
  if (n_ctx > 8 || (N & 3)) return KL_ERR_SHAPE;
  Tabs tabs;
  for (int i = 0; i < 8; ++i) tabs.t[i] = i < n_ctx ? ctxk[i] : nullptr;
  hipLaunchKernelGGL(p1_gather_kernel, dim3(grid_for((long)B * T * (N / 4), 256)), dim3(256), 0, stream, EK, tabs,
                     n_ctx, bias, idx, ctx, B, T, N / 4, P);
  return ok();
}

int kl_launch_colsum_bf16(const bf16_t* in, long ld, int rows, int cols, float* out, hipStream_t stream) {
  int ry = (rows + 255) / 256;
  if (ry > 64) ry = 64;
  if (ry < 1) ry = 1;
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3((cols + 63) / 64, ry), dim3(256), 0, stream, in, ld, rows, cols, out);
  return ok();
}

int kl_launch_rows_to_state(const void* h_rows, int h_is_f32, const float* c_rows, int B, int W, int L, int layer,
                            float* states, hipStream_t stream) {
  hipLaunchKernelGGL(rows_to_state_kernel, dim3(grid_for((long)B * W, 256)), dim3(256), 0, stream, h_rows, h_is_f32,
                     c_rows, B, W, L, layer, states);
  return ok();
}

int kl_launch_ctx_grads(const float* Ctx, const float* K0rows, long ldk, int R, int D, const float* dT, long ldt, int N,
                        float* gK, long ldg, float* gCtx, hipStream_t stream) {
  hipLaunchKernelGGL(ctx_grad_k_kernel, dim3(grid_for((long)D * N, 256)), dim3(256), 0, stream, Ctx, R, D, dT, ldt, N,
                     gK, ldg);
  const long items = (long)R * D;
  hipLaunchKernelGGL(ctx_grad_c_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), 0, stream, K0rows, ldk, R, D, dT,
                     ldt, N, gCtx);
  return ok();
}

int kl_launch_rows_tm_to_bm(const float* in, long ld_in, float* out, int B, int T, int V, hipStream_t stream) {
  hipLaunchKernelGGL(rows_tm_to_bm_kernel, dim3(grid_for((long)B * T * V, 256)), dim3(256), 0, stream, in, ld_in, out,
                     B, T, V);
  return ok();
}
