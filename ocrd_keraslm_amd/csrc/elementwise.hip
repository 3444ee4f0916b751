// HBM-bound helper kernels around the contractions: embedding gather (F1),
// layout transposes, f32->bf16 weight preparation (hi/lo split), softmax +
// cross-entropy + dlogits (F5/F6/B1), one-hot builders for the embedding
// gradients (B7), embedding regularisers (F7), clip+Adam (O1).
// All accesses are 16-byte vectors where the layout allows it.
#include <string.h>

#include "kl_common.h"
#include "kl_kernels.h"

namespace {

// ---- F1: X[t*B+b] = bf16([E[idx[b,t]] | Ctx_n[ctx[b,t,n]] ... | 0 pad]) ----------
struct CtxTabs { const float* t[8]; };

__global__ void embed_gather_kernel(const float* __restrict__ E, CtxTabs tabs, int n_ctx, int ctx_dim, int W,
                                    const int* __restrict__ idx, const int* __restrict__ ctx, int B, int T,
                                    bf16_t* __restrict__ X, long ldx, int Dp) {
  const int chunks = Dp >> 3;
  const long total = (long)B * T * chunks;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c = (int)(e % chunks);
    const long row = e / chunks;          // time-major row = t*B + b
    const int b = (int)(row % B), t = (int)(row / B);
    const int id = idx[(long)b * T + t];
    frag16 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int d = c * 8 + j;
      float v = 0.f;
      if (d < W) {
        v = E[(long)id * W + d];
      } else if (d < W + n_ctx * ctx_dim) {
        const int n = (d - W) / ctx_dim, dd = (d - W) % ctx_dim;
        const int cid = ctx[((long)b * T + t) * n_ctx + n];
        v = tabs.t[n][(long)cid * ctx_dim + dd];
      }
      o.s[j] = f2bf(v);
    }
    *reinterpret_cast<uint4*>(X + row * ldx + c * 8) = o.u;
  }
}

// ---- bf16 transpose through LDS: out[c][r] = in[r][c] -----------------------------
__global__ void transpose_bf16_kernel(const bf16_t* __restrict__ in, long ld_in, bf16_t* __restrict__ out,
                                      long ld_out, int rows, int cols) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 256 threads: 4 rows per pass
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? in[(long)r * ld_in + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) out[(long)c * ld_out + r] = tile[tx][i];
  }
}

// ---- f32 -> bf16 (hi [+ lo]) with optional transpose -----------------------------
__global__ void f32_to_bf16_kernel(const float* __restrict__ in, long ld_in, int rows, int cols,
                                   bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo, long ld_out,
                                   int transpose) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? in[(long)r * ld_in + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    int orow, ocol;
    float v;
    if (transpose) {
      orow = c0 + i; ocol = r0 + tx; v = tile[tx][i];
      if (orow >= cols || ocol >= rows) continue;
    } else {
      orow = r0 + i; ocol = c0 + tx; v = tile[i][tx];
      if (orow >= rows || ocol >= cols) continue;
    }
    bf16_t hi, lo;
    split_bf16(v, hi, lo);
    out_hi[(long)orow * ld_out + ocol] = hi;
    if (out_lo) out_lo[(long)orow * ld_out + ocol] = lo;
  }
}

// ... a list of such conversions in one launch: blockIdx.z = the job (read through the kernel-argument segment: a by-value
// array indexed at run time would be copied to scratch), blockIdx.x / y = its 64 x 64 tile (workgroups beyond a job's tiles leave)
__global__ void f32_to_bf16_jobs_kernel(const KlConvJobs jobs) {
  __shared__ float tile[64][65];
  const KlConvJob __attribute__((address_space(4)))* jp =
      (const KlConvJob __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + blockIdx.z;
  const float* in = jp->in;
  const long ld_in = jp->ld_in, ld_out = jp->ld_out;
  const int rows = jp->rows, cols = jp->cols, rows_pad = jp->rows_pad, transpose = jp->transpose;
  bf16_t* out_hi = jp->out_hi;
  bf16_t* out_lo = jp->out_lo;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  if (r0 >= rows_pad || c0 >= cols) return;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? in[(long)r * ld_in + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    int orow, ocol;
    float v;
    if (transpose) {
      orow = c0 + i; ocol = r0 + tx; v = tile[tx][i];
      if (orow >= cols || ocol >= rows_pad) continue;
    } else {
      orow = r0 + i; ocol = c0 + tx; v = tile[i][tx];
      if (orow >= rows_pad || ocol >= cols) continue;
    }
    bf16_t hi, lo;
    split_bf16(v, hi, lo);
    out_hi[(long)orow * ld_out + ocol] = hi;
    if (out_lo) out_lo[(long)orow * ld_out + ocol] = lo;
  }
}

// ---- softmax + Keras categorical_crossentropy + accuracy + dlogits ---------------
// one wave per row.  Rows are time-major (row = t*B + b) when time_major != 0,
// targets are [B][T] with -1 = all-zero one-hot row (padded tail).
__global__ void softmax_ce_kernel(float* __restrict__ logits, long ld, int rows, int V, const int* __restrict__ tgt,
                                  int B, int T, float inv_count, bf16_t* __restrict__ dlogits, long ld_dl,
                                  float* __restrict__ rowstat, int time_major, int last_only) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* x = logits + (long)row * ld;
  float mx = -INFINITY;
  int amax = 0;
  for (int v = lane; v < V; v += 64) {
    const float a = x[v];
    if (a > mx) { mx = a; amax = v; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(mx, off);
    const int oi = __shfl_xor(amax, off);
    if (o > mx || (o == mx && oi < amax)) { mx = o; amax = oi; }
  }
  float sum = 0.f;
  for (int v = lane; v < V; v += 64) sum += expf(x[v] - mx);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  const float inv = 1.f / sum;
  int t = -2;
  bool counts = true;        // stateless windows (last_only): only the last position has a target at all
  if (tgt) {
    const int b = time_major ? row % B : row / T;
    const int tt = time_major ? row / B : row % T;
    t = tgt[(long)b * T + tt];
    if (last_only && tt != T - 1) { t = -1; counts = false; }
    if (t < -1) { t = -1; counts = false; }      // (a dummy stream added by the caller's padding: no loss, no gradient, NO accuracy either)
  }
  // probabilities are written back over the logits for the callers that read them (rating); a
  // training window only needs dlogits and the row statistics, so it skips that pass over HBM
  const bool keep_probs = dlogits == nullptr;
  float pt = 0.f;
  for (int v = lane; v < V; v += 64) {
    const float p = expf(x[v] - mx) * inv;
    if (keep_probs) x[v] = p;
    if (v == t) pt = p;
  }
  if (!tgt) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) pt += __shfl_xor(pt, off);
  const bool valid = t >= 0;
  const bool active = valid && pt >= 1e-7f && pt <= 1.f - 1e-7f;
  if (dlogits) {
    bf16_t* d = dlogits + (long)row * ld_dl;
    for (int v = lane; v < ld_dl; v += 64) {   // pad columns [V, ld_dl) are written as zeros
      float g = (active && v < V) ? expf(x[v] - mx) * inv : 0.f;
      if (active && v == t) g -= 1.f;
      d[v] = f2bf(g * inv_count);
    }
  }
  if (lane == 0 && rowstat) {
    float l = 0.f;
    if (valid) {
      const float pc = fminf(fmaxf(pt, 1e-7f), 1.f - 1e-7f);
      l = -logf(pc) * inv_count;
    }
    const int tsafe = valid ? t : 0;
    rowstat[2 * (long)row] = l;
    rowstat[2 * (long)row + 1] = (counts && amax == tsafe) ? inv_count : 0.f;
  }
}

// The same for V <= 256 with V, ld and ld_dl multiples of 4 (the usual character vocabularies): a lane
// keeps its four logits in registers -- ONE pass over the row (16-byte loads), one exponential per
// element, 8-byte stores of the bf16 gradient row -- instead of four strided passes.
__global__ void softmax_ce_v256_kernel(float* __restrict__ logits, long ld, int rows, int V, const int* __restrict__ tgt,
                                       int B, int T, float inv_count, bf16_t* __restrict__ dlogits, long ld_dl,
                                       float* __restrict__ rowstat, int time_major, int last_only) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* x = logits + (long)row * ld;
  const int v0 = lane * 4;
  const bool in = v0 < V;
  float4 a = in ? *reinterpret_cast<const float4*>(x + v0) : float4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  float e[4] = {a.x, a.y, a.z, a.w};
  float mx = e[0];
  int amax = v0;
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (e[k] > mx) { mx = e[k]; amax = v0 + k; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(mx, off);
    const int oi = __shfl_xor(amax, off);
    if (o > mx || (o == mx && oi < amax)) { mx = o; amax = oi; }
  }
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    e[k] = in ? expf(e[k] - mx) : 0.f;
    sum += e[k];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
  const float inv = 1.f / sum;
  int t = -2;
  bool counts = true;
  if (tgt) {
    const int b = time_major ? row % B : row / T;
    const int tt = time_major ? row / B : row % T;
    t = tgt[(long)b * T + tt];
    if (last_only && tt != T - 1) { t = -1; counts = false; }
    if (t < -1) { t = -1; counts = false; }      // (a dummy stream added by the caller's padding: no loss, no gradient, NO accuracy either)
  }
  float pt = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    e[k] *= inv;
    if (v0 + k == t) pt = e[k];
  }
  if (dlogits == nullptr && in) *reinterpret_cast<float4*>(x + v0) = float4{e[0], e[1], e[2], e[3]};
  if (!tgt) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) pt += __shfl_xor(pt, off);
  const bool valid = t >= 0;
  const bool active = valid && pt >= 1e-7f && pt <= 1.f - 1e-7f;
  if (dlogits && v0 < ld_dl) {      // pad columns [V, ld_dl) are written as zeros
    unsigned short g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gk = (active && in) ? e[k] : 0.f;
      if (active && v0 + k == t) gk -= 1.f;
      g[k] = f2bf(gk * inv_count);
    }
    *reinterpret_cast<uint2*>(dlogits + (long)row * ld_dl + v0) = uint2{(unsigned)g[0] | ((unsigned)g[1] << 16), (unsigned)g[2] | ((unsigned)g[3] << 16)};
  }
  if (lane == 0 && rowstat) {
    float l = 0.f;
    if (valid) {
      const float pc = fminf(fmaxf(pt, 1e-7f), 1.f - 1e-7f);
      l = -logf(pc) * inv_count;
    }
    const int tsafe = valid ? t : 0;
    rowstat[2 * (long)row] = l;
    rowstat[2 * (long)row + 1] = (counts && amax == tsafe) ? inv_count : 0.f;
  }
}

// sum of the per-row (loss, hit) pairs -> loss_acc[0], loss_acc[1]; a few blocks, one atomic pair each
__global__ void rowstat_reduce_kernel(const float* __restrict__ rowstat, int rows, float* __restrict__ loss_acc) {
  __shared__ float red[2][256];
  float a = 0.f, b = 0.f;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) {
    const float2 v = reinterpret_cast<const float2*>(rowstat)[r];
    a += v.x;
    b += v.y;
  }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = b;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (threadIdx.x < s) { red[0][threadIdx.x] += red[0][threadIdx.x + s]; red[1][threadIdx.x] += red[1][threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { atomicAdd(loss_acc + 0, red[0][0]); atomicAdd(loss_acc + 1, red[1][0]); }
}

// ---- clip + Adam (Keras 2.3.1 formula; lr_t carries the bias correction) ---------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, size_t n, float lr_t, float b1, float b2, float eps, float clip,
                            float grad_scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float gi = g[i] * grad_scale;      // (1 / world size after a summing all-reduce: the mean over the ranks, no extra pass)
    gi = fminf(fmaxf(gi, -clip), clip);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
  }
}

// ---- one-hot^T builder: out[ids[b,t,col]][t*B+b] = 1 (buffer pre-zeroed) ----------
__global__ void onehot_t_kernel(const int* __restrict__ ids, int B, int T, int n_classes, int col, int n_cols,
                                bf16_t* __restrict__ out, long ld) {
  const long total = (long)B * T;
  for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < total; r += (long)gridDim.x * blockDim.x) {
    const int b = (int)(r % B), t = (int)(r / B);
    const int id = ids[((long)b * T + t) * n_cols + col];
    if (id >= 0 && id < n_classes) out[(long)id * ld + r] = 0x3F80;   // bf16(1.0)
  }
}

// ... the same matrix written DENSE, zeros included: one pass of 16-byte stores over [n_rows][ld] instead of a zero fill
// (400 MB at the bench shape) followed by the scatter.  Thread = eight consecutive columns r (streams b .. b + 7 of
// one step t when B is a multiple of 8) x a tile of 32 class rows; columns from B*T up to ld are zero.
__global__ void onehot_dense_kernel(const int* __restrict__ ids, int B, int T, int n_rows, int col, int n_cols,
                                    bf16_t* __restrict__ out, long ld) {
  const long r0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (r0 >= ld) return;
  const long total = (long)B * T;
  int id[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const long r = r0 + j;
    id[j] = -1;
    if (r < total) {
      const int b = (int)(r % B), t = (int)(r / B);
      id[j] = ids[((long)b * T + t) * n_cols + col];
    }
  }
  const int v0 = blockIdx.y * 32;
  for (int v = v0; v < v0 + 32 && v < n_rows; ++v) {
    uint4 w;
    w.x = (id[0] == v ? 0x3F80u : 0u) | (id[1] == v ? 0x3F800000u : 0u);
    w.y = (id[2] == v ? 0x3F80u : 0u) | (id[3] == v ? 0x3F800000u : 0u);
    w.z = (id[4] == v ? 0x3F80u : 0u) | (id[5] == v ? 0x3F800000u : 0u);
    w.w = (id[6] == v ? 0x3F80u : 0u) | (id[7] == v ? 0x3F800000u : 0u);
    *reinterpret_cast<uint4*>(out + (long)v * ld + r0) = w;
  }
}

// ---- embedding regularisers: gradient (+=) and value ------------------------------
// mode 0 = chars (rating.py:222-246), mode 1 = contexts (rating.py:187-220).
// Two launches per table: table statistics (one block: column means / partial column
// sums / row norms into a scratch vector), then an elementwise multi-block pass that
// applies the gradient and reduces the loss value.
//   scratch = [mean D | s1 D | s2 D | nr R | N1 N2]
__global__ void reg_stats_kernel(const float* __restrict__ X, int R, int D, float* __restrict__ scratch) {
  extern __shared__ float sm[];
  float* col = sm;             // [D] sum over rows 1..R-1
  float* red = sm + D;         // [2][blockDim]
  const int tid = threadIdx.x, nt = blockDim.x;
  float* mean = scratch;
  float* s1 = scratch + D;
  float* s2 = scratch + 2 * D;
  float* nr = scratch + 3 * D;
  for (int d = tid; d < D; d += nt) col[d] = 0.f;
  __syncthreads();
  if (D <= nt) {
    // thread = (row group, column): strided rows summed in registers, one LDS add per thread
    const int ng = nt / D, g = tid / D, d = tid % D;
    if (g < ng) {
      float a = 0.f;
#pragma unroll 8
      for (int r = 1 + g; r < R; r += ng) a += X[(long)r * D + d];
      atomicAdd(&col[d], a);
    }
  } else {
    for (int d = tid; d < D; d += nt) {
      float a = 0.f;
#pragma unroll 8
      for (int r = 1; r < R; ++r) a += X[(long)r * D + d];
      col[d] = a;
    }
  }
  float n1 = 0.f, n2 = 0.f;
  for (int r = tid >> 6; r < R; r += nt >> 6) {
    float a = 0.f;
    for (int d = tid & 63; d < D; d += 64) { const float x = X[(long)r * D + d]; a += x * x; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if ((tid & 63) == 0) {
      nr[r] = a;
      if (r >= 1) { n1 += a; n2 += a * a; }
    }
  }
  __syncthreads();
  for (int d = tid; d < D; d += nt) {
    const float a = col[d];
    mean[d] = a / (float)(R - 1);
    s1[d] = a - X[(long)(R - 1) * D + d];
    s2[d] = a - X[(long)1 * D + d];
  }
  red[tid] = n1;
  red[nt + tid] = n2;
  __syncthreads();
  for (int s = nt >> 1; s > 0; s >>= 1) {
    if (tid < s) { red[tid] += red[tid + s]; red[nt + tid] += red[nt + tid + s]; }
    __syncthreads();
  }
  if (tid == 0) { scratch[3 * D + R] = red[0]; scratch[3 * D + R + 1] = red[nt]; }
}

__global__ void reg_apply_kernel(const float* __restrict__ X, int R, int D, float* __restrict__ gX, int mode,
                                 const float* __restrict__ scratch, float* __restrict__ loss_acc) {
  __shared__ float red[256];
  const float* mean = scratch;
  const float* s1 = scratch + D;
  const float* s2 = scratch + 2 * D;
  const float* nr = scratch + 3 * D;
  const float N1 = scratch[3 * D + R], N2 = scratch[3 * D + R + 1];
  const float c_low = mode == 0 ? 0.01f : 0.02f;
  const float Rp = (float)(R - 1);
  float loss = 0.f;
  const long total = (long)R * D;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int r = (int)(e / D), d = (int)(e % D);
    const float x = X[e];
    const float q = 1.f - nr[r];
    float g = -4.f * c_low * q * x;                      // low-rank term on every row
    if (d == 0) loss += c_low * q * q;
    if (mode == 0) {
      if (r == 0) { const float u = x - mean[d]; g += 2.f * u; loss += u * u; }
    } else {
      if (r >= 2) g += 0.2f * s1[d];                     // smoothness (all-pairs form, rating.py:206)
      if (r == 0) {
        const float md = mean[d];
        g += 4.f * (Rp * x - N1 * md);                   // underspecification at index 0
        loss += 0.2f * s1[d] * s2[d] + 2.f * (Rp * x * x - 2.f * x * md * N1 + md * md * N2);
      }
    }
    gX[e] += g;
  }
  red[threadIdx.x] = loss;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && loss_acc) atomicAdd(loss_acc + 2, red[0]);
}

// ---- carried state (slot layout [B][2L][W] f32) -> bf16 h rows + f32 c rows -------
__global__ void state_to_rows_kernel(const float* __restrict__ states, int B, int W, int L, int layer,
                                     bf16_t* __restrict__ h_bf16, float* __restrict__ h_f32,
                                     float* __restrict__ c_f32) {
  const long total = (long)B * W;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int b = (int)(e / W), u = (int)(e % W);
    const float* s = states + ((long)b * 2 * L + 2 * layer) * W;
    if (h_bf16) h_bf16[e] = f2bf(s[u]);
    if (h_f32) h_f32[e] = s[u];
    if (c_f32) c_f32[e] = s[W + u];
  }
}

__global__ void fill_f32_kernel(float* p, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

inline int grid_for(long total, int block) {
  long g = (total + block - 1) / block;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}
inline int ok() { return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH; }

}  // namespace

int kl_launch_embed_gather(const float* E, const float* const* ctx_tabs, int n_ctx, int ctx_dim, int W,
                           const int* idx, const int* ctx, int B, int T, bf16_t* X, long ldx, int Dp,
                           hipStream_t stream) {
  if (n_ctx > 8 || (Dp & 7) || (ldx & 7)) return KL_ERR_SHAPE;
  CtxTabs tabs;
  for (int i = 0; i < 8; ++i) tabs.t[i] = i < n_ctx ? ctx_tabs[i] : nullptr;
  const long total = (long)B * T * (Dp >> 3);
  hipLaunchKernelGGL(embed_gather_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, E, tabs, n_ctx, ctx_dim,
                     W, idx, ctx, B, T, X, ldx, Dp);
  return ok();
}

int kl_launch_transpose_bf16(const bf16_t* in, long ld_in, bf16_t* out, long ld_out, int rows, int cols,
                             hipStream_t stream) {
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, stream, in, ld_in, out, ld_out, rows, cols);
  return ok();
}

int kl_launch_f32_to_bf16_t(const float* in, long ld_in, int rows, int cols, bf16_t* out_hi, bf16_t* out_lo,
                            long ld_out, int transpose, hipStream_t stream) {
  dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  hipLaunchKernelGGL(f32_to_bf16_kernel, grid, dim3(256), 0, stream, in, ld_in, rows, cols, out_hi, out_lo, ld_out,
                     transpose);
  return ok();
}

int kl_launch_f32_to_bf16_jobs(const KlConvJob* jobs, int n, hipStream_t stream) {
  if (n < 1 || n > KL_CONV_MAX_JOBS) return KL_ERR_SHAPE;
  KlConvJobs all;
  memset(&all, 0, sizeof(all));
  int gx = 1, gy = 1;
  for (int j = 0; j < n; ++j) {
    all.job[j] = jobs[j];
    if (all.job[j].rows_pad < all.job[j].rows) all.job[j].rows_pad = all.job[j].rows;
    gx = max(gx, (jobs[j].cols + 63) / 64);
    gy = max(gy, (all.job[j].rows_pad + 63) / 64);
  }
  hipLaunchKernelGGL(f32_to_bf16_jobs_kernel, dim3(gx, gy, n), dim3(256), 0, stream, all);
  return ok();
}

// ---- table of all sums of a row of A and a row of B, bf16 (8 elements per thread) ----
__global__ void comb_table_kernel(const float* __restrict__ A, const float* __restrict__ B, int R2, int N8, bf16_t* __restrict__ out, long total) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int j = (int)(e % N8);
    const long row = e / N8;
    const int c = (int)(row % R2), v = (int)(row / R2);
    const float4* pa = reinterpret_cast<const float4*>(A + ((long)v * N8 + j) * 8);
    const float4* pb = reinterpret_cast<const float4*>(B + ((long)c * N8 + j) * 8);
    const float4 a0 = pa[0], a1 = pa[1], b0 = pb[0], b1 = pb[1];
    uint4 o;
    o.x = (unsigned)f2bf(a0.x + b0.x) | ((unsigned)f2bf(a0.y + b0.y) << 16);
    o.y = (unsigned)f2bf(a0.z + b0.z) | ((unsigned)f2bf(a0.w + b0.w) << 16);
    o.z = (unsigned)f2bf(a1.x + b1.x) | ((unsigned)f2bf(a1.y + b1.y) << 16);
    o.w = (unsigned)f2bf(a1.z + b1.z) | ((unsigned)f2bf(a1.w + b1.w) << 16);
    reinterpret_cast<uint4*>(out)[e] = o;
  }
}
__global__ void rows_tm_kernel(const int* __restrict__ idx, const int* __restrict__ ctx, int n_ctx, int Bn, int T, int R2, int* __restrict__ out) {
  const long total = (long)Bn * T;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int b = (int)(e % Bn), t = (int)(e / Bn);
    const long src = (long)b * T + t;
    out[e] = idx[src] * R2 + (n_ctx > 0 ? ctx[src * n_ctx] : 0);
  }
}

int kl_launch_comb_table(const float* A, const float* B, int R1, int R2, int N, bf16_t* out, hipStream_t stream) {
  if ((N & 7) || R1 < 1 || R2 < 1) return KL_ERR_SHAPE;
  const long total = (long)R1 * R2 * (N / 8);
  hipLaunchKernelGGL(comb_table_kernel, dim3(grid_for(total, 256)), dim3(256), 0, stream, A, B, R2, N / 8, out, total);
  return ok();
}
int kl_launch_rows_tm(const int* idx, const int* ctx, int n_ctx, int Bn, int T, int R2, int* out, hipStream_t stream) {
  hipLaunchKernelGGL(rows_tm_kernel, dim3(grid_for((long)Bn * T, 256)), dim3(256), 0, stream, idx, ctx, n_ctx, Bn, T, R2, out);
  return ok();
}

int kl_launch_softmax_ce(float* logits, long ld, int rows, int V, const int* tgt, int B, int T, float inv_count,
                         bf16_t* dlogits, long ld_dl, float* loss_acc, float* rowstat, int time_major,
                         hipStream_t stream, int last_only) {
  dim3 grid((rows + 3) / 4);
  const bool stats = tgt != nullptr && loss_acc != nullptr && rowstat != nullptr;
  const bool one_pass = V <= 256 && (V & 3) == 0 && (ld & 3) == 0 && (!dlogits || ((ld_dl & 3) == 0 && ld_dl <= 256));
  if (one_pass)
    hipLaunchKernelGGL(softmax_ce_v256_kernel, grid, dim3(256), 0, stream, logits, ld, rows, V, tgt, B, T, inv_count,
                       dlogits, ld_dl, stats ? rowstat : nullptr, time_major, last_only);
  else
    hipLaunchKernelGGL(softmax_ce_kernel, grid, dim3(256), 0, stream, logits, ld, rows, V, tgt, B, T, inv_count,
                       dlogits, ld_dl, stats ? rowstat : nullptr, time_major, last_only);
  if (stats) {
    int nb = (rows + 4095) / 4096;
    if (nb > 64) nb = 64;
    hipLaunchKernelGGL(rowstat_reduce_kernel, dim3(nb), dim3(256), 0, stream, rowstat, rows, loss_acc);
  }
  return ok();
}

int kl_launch_rowstat_reduce(const float* rowstat, int rows, float* loss_acc, hipStream_t stream) {
  int nb = (rows + 4095) / 4096;
  if (nb > 64) nb = 64;
  hipLaunchKernelGGL(rowstat_reduce_kernel, dim3(nb), dim3(256), 0, stream, rowstat, rows, loss_acc);
  return ok();
}

int kl_launch_adam(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2,
                   float eps, float clip, float grad_scale, hipStream_t stream) {
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for((long)n, 256)), dim3(256), 0, stream, p, g, m, v, n, lr_t, b1, b2,
                     eps, clip, grad_scale);
  return ok();
}

int kl_launch_onehot_t(const int* ids, int B, int T, int n_classes, int col, int n_cols, bf16_t* out, long ld,
                       hipStream_t stream) {
  hipLaunchKernelGGL(onehot_t_kernel, dim3(grid_for((long)B * T, 256)), dim3(256), 0, stream, ids, B, T, n_classes,
                     col, n_cols, out, ld);
  return ok();
}

// dense form: writes all n_rows x ld entries (ld a multiple of 8, out 16-byte aligned); KL_ERR_SHAPE otherwise
int kl_launch_onehot_dense(const int* ids, int B, int T, int n_rows, int col, int n_cols, bf16_t* out, long ld,
                           hipStream_t stream) {
  if ((ld & 7) || ((size_t)out & 15) || ld < (long)B * T) return KL_ERR_SHAPE;
  dim3 grid((unsigned)((ld / 8 + 255) / 256), (unsigned)((n_rows + 31) / 32));
  hipLaunchKernelGGL(onehot_dense_kernel, grid, dim3(256), 0, stream, ids, B, T, n_rows, col, n_cols, out, ld);
  return ok();
}

int kl_launch_regulariser_grads(const float* E, int V, int W, const float* const* ctx_tabs, int n_ctx, int ctx_vocab,
                                int ctx_dim, float* gE, float* const* gCtx, float* loss_acc, float* scratch,
                                hipStream_t stream) {
  // scratch: 3*max(W, ctx_dim) + max(V, ctx_vocab) + 2 floats (kl_reg_scratch_floats)
  const int nt = 1024;
  if (V >= 2) {
    const size_t lds = (size_t)(W + 2 * nt) * sizeof(float);
    if (lds > 64 * 1024) return KL_ERR_SHAPE;
    hipLaunchKernelGGL(reg_stats_kernel, dim3(1), dim3(nt), lds, stream, E, V, W, scratch);
    hipLaunchKernelGGL(reg_apply_kernel, dim3(grid_for((long)V * W, 256)), dim3(256), 0, stream, E, V, W, gE, 0, scratch,
                       loss_acc);
  }
  for (int n = 0; n < n_ctx; ++n) {
    const size_t lds = (size_t)(ctx_dim + 2 * nt) * sizeof(float);
    hipLaunchKernelGGL(reg_stats_kernel, dim3(1), dim3(nt), lds, stream, ctx_tabs[n], ctx_vocab, ctx_dim, scratch);
    hipLaunchKernelGGL(reg_apply_kernel, dim3(grid_for((long)ctx_vocab * ctx_dim, 256)), dim3(256), 0, stream, ctx_tabs[n],
                       ctx_vocab, ctx_dim, gCtx[n], 1, scratch, loss_acc);
  }
  return ok();
}

int kl_launch_state_to_rows(const float* states, int B, int W, int L, int layer, bf16_t* h_bf16, float* h_f32,
                            float* c_f32, hipStream_t stream) {
  hipLaunchKernelGGL(state_to_rows_kernel, dim3(grid_for((long)B * W, 256)), dim3(256), 0, stream, states, B, W, L,
                     layer, h_bf16, h_f32, c_f32);
  return ok();
}

int kl_launch_fill_f32(float* p, size_t n, float v, hipStream_t stream) {
  hipLaunchKernelGGL(fill_f32_kernel, dim3(grid_for((long)n, 256)), dim3(256), 0, stream, p, n, v);
  return ok();
}

// ---- zero fills as kernels ---------------------------------------------------------
// hipMemsetAsync nodes inside a replayed hipGraph were observed to lose their effect /
// ordering after the first host synchronisation on ROCm 7.2 (gradients accumulated
// across windows, status words kept stale values); plain kernels replay correctly.
namespace {
__global__ void zero_kernel(uint4* p16, size_t n16, unsigned* tail, size_t ntail) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    p16[i] = uint4{0, 0, 0, 0};
  if (blockIdx.x == 0 && threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
// write-through zeroing for words other workgroups poll (hand-off counters)
__global__ void zero_coherent_kernel(unsigned* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __hip_atomic_store(p + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace

int kl_zero_async(void* p, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return 0;
  if ((reinterpret_cast<uintptr_t>(p) & 3) || (bytes & 3)) return KL_ERR_ARG;
  uintptr_t a = reinterpret_cast<uintptr_t>(p);
  size_t head = ((16 - (a & 15)) & 15);
  if (head > bytes) head = bytes;
  // unaligned head words + 16-byte body + tail words
  unsigned* hp = reinterpret_cast<unsigned*>(p);
  size_t nhead = head / 4;
  uint4* body = reinterpret_cast<uint4*>(a + head);
  size_t n16 = (bytes - head) / 16;
  unsigned* tp = reinterpret_cast<unsigned*>(a + head + n16 * 16);
  size_t ntail = (bytes - head - n16 * 16) / 4;
  if (nhead) hipLaunchKernelGGL(zero_kernel, dim3(1), dim3(64), 0, stream, (uint4*)nullptr, (size_t)0, hp, nhead);
  long g = (long)((n16 + 255) / 256);
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)g), dim3(256), 0, stream, body, n16, tp, ntail);
  return ok();
}

namespace {
__global__ void fill16_kernel(uint4* p, size_t n16, unsigned v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    p[i] = uint4{v, v, v, v};
}
}  // namespace

// fill a 16-byte aligned region (a multiple of 16 bytes) with one 32-bit pattern:
// the sentinel pre-fill of the scans' exchange buffers
int kl_fill_u32_async(void* p, size_t bytes, unsigned value, hipStream_t stream) {
  if (bytes == 0) return 0;
  if ((reinterpret_cast<uintptr_t>(p) & 15) || (bytes & 15)) return KL_ERR_ARG;
  const size_t n16 = bytes / 16;
  long g = (long)((n16 + 255) / 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(fill16_kernel, dim3((unsigned)g), dim3(256), 0, stream, reinterpret_cast<uint4*>(p), n16, value);
  return ok();
}

int kl_zero_coherent_async(unsigned* p, size_t n_words, hipStream_t stream) {
  if (n_words == 0) return 0;
  long g = (long)((n_words + 255) / 256);
  if (g > 1024) g = 1024;
  hipLaunchKernelGGL(zero_coherent_kernel, dim3((unsigned)g), dim3(256), 0, stream, p, n_words);
  return ok();
}

namespace {
__global__ void fill_bf16_kernel(bf16_t* p, size_t n, bf16_t v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
}  // namespace

int kl_launch_fill_bf16(bf16_t* p, size_t n, unsigned short bits, hipStream_t stream) {
  long g = (long)((n + 255) / 256);
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(fill_bf16_kernel, dim3((unsigned)g), dim3(256), 0, stream, p, n, (bf16_t)bits);
  return ok();
}
