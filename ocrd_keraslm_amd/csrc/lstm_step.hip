// Thin fused LSTM cell steps for gfx950: the recurrence is a chain of T tiny
// dependent contractions, so each time step is ONE launch whose workgroups are
// deliberately small (32 rows x 4 hidden units forward, 16 rows x 16 units
// backward) -- the per-step cost is the per-CU L2->register byte rate, and thin
// workgroups spread the weight slice + state rows over all 256 CUs.  Several
// independent steps (the layer wavefront: layer l at time d-l) share one launch
// through blockIdx.z.  The chain of launches is captured in a hipGraph by the
// caller (api.hip).
//
// Restates: Keras LSTM cell at rating.py:126-145 (gate order i,f,c,o), the
// incremental step of Rater.predict rating.py:578-639 (rows = hypotheses, state
// rows addressed through slot indices), and TF autodiff of the same cell.
#include "kl_common.h"
#include "kl_kernels.h"

namespace {

constexpr int MAX_FUSED = 4;

struct FwdPack { KlFwdStep s[MAX_FUSED]; };
struct BwdPack { KlBwdStep s[MAX_FUSED]; };

// Load lane's A fragment (8 consecutive k) of row `row` at element offset k.
// f32 sources are split on the fly into bf16 hi (+ lo residual).
__device__ __forceinline__ void load_a(const KlOperand& op, long row, int k, bool want_lo, bf16x8& hi, bf16x8& lo) {
  if (op.a_is_f32) {
    const float* p = reinterpret_cast<const float*>(op.A) + row * op.lda + k;
    const float4 x0 = *reinterpret_cast<const float4*>(p);
    const float4 x1 = *reinterpret_cast<const float4*>(p + 4);
    const float xs[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    frag16 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      h.s[j] = f2bf(xs[j]);
      l.s[j] = want_lo ? f2bf(xs[j] - bf2f(h.s[j])) : (bf16_t)0;
    }
    hi = h.v;
    lo = l.v;
  } else {
    frag16 h;
    h.u = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(op.A) + row * op.lda + k);
    hi = h.v;
    frag16 l;
    l.u = uint4{0, 0, 0, 0};
    lo = l.v;
  }
}

// acc[mt] += A[rows of M-tile mt][k-slice of this wave] . WT[wt_row][same k]^T
template <int NMT>
__device__ __forceinline__ void accumulate(const KlOperand& op, int split, const long (&arow)[NMT], long wt_row,
                                           int wave, int lane, f32x4 (&acc)[NMT]) {
  const int nks = op.K >> 5;
  const int kq = (lane >> 4) * 8;
  const bool use_lo = (split == 3) && (op.WT_lo != nullptr);
  for (int ks = wave; ks < nks; ks += 4) {
    const int k = ks * 32 + kq;
    frag16 bh, bl;
    bh.u = *reinterpret_cast<const uint4*>(op.WT_hi + wt_row * op.ldw + k);
    if (use_lo) bl.u = *reinterpret_cast<const uint4*>(op.WT_lo + wt_row * op.ldw + k);
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
      bf16x8 ah, al;
      load_a(op, arow[mt], k, use_lo && op.a_is_f32, ah, al);
      acc[mt] = mfma16(ah, bh.v, acc[mt]);
      if (use_lo) {
        acc[mt] = mfma16(ah, bl.v, acc[mt]);
        if (op.a_is_f32) acc[mt] = mfma16(al, bh.v, acc[mt]);
      }
    }
  }
}

// ---------------------------------------------------------------- forward
// block = 256 threads; blockIdx.x = group of 4 hidden units; blockIdx.y = block
// of 32 rows; blockIdx.z = fused step.  The 16 MFMA columns are gate*4 + unit.
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(const FwdPack pack) {
  const KlFwdStep& S = pack.s[blockIdx.z];
  const int r0 = blockIdx.y * 32;
  if (r0 >= S.n_rows) return;
  const int W = S.W;
  const int u0 = blockIdx.x * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float part[4][2][16][17];

  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const int n = lane & 15;
  const long wt_row = (long)(n >> 2) * W + u0 + (n & 3);
  for (int p = 0; p < S.n_ops; ++p) {
    const KlOperand& op = S.op[p];
    long arow[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      int r = min(r0 + mt * 16 + (lane & 15), S.n_rows - 1);
      arow[mt] = op.row_index ? (long)op.row_index[r] : (long)r;
    }
    accumulate<2>(op, S.split, arow, wt_row, wave, lane, acc);
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][mt][(lane >> 4) * 4 + r][n] = acc[mt][r];
  __syncthreads();
  if (tid >= 128) return;
  const int lr = tid >> 2, j = tid & 3;
  const int row = r0 + lr;
  if (row >= S.n_rows) return;
  const int u = u0 + j;
  float z[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += part[w][lr >> 4][lr & 15][g * 4 + j];
    const long col = (long)g * W + u;
    if (S.T1) v += S.T1[(long)(S.i1 ? S.i1[row] : row) * S.t1_ld + col];
    if (S.T2) v += S.T2[(long)(S.i2 ? S.i2[row] : row) * S.t2_ld + col];
    if (S.bias) v += S.bias[col];
    z[g] = v;
  }
  const float gi = sigmoidf_(z[0]), gf = sigmoidf_(z[1]), gg = tanhf_(z[2]), go = sigmoidf_(z[3]);
  const long crow = S.c_prev_index ? (long)S.c_prev_index[row] : (long)row;
  const float cp = S.c_prev ? S.c_prev[crow * S.c_prev_ld + u] : 0.f;
  const float c = gf * cp + gi * gg;
  const float h = go * tanhf_(c);
  const long orow = S.out_index ? (long)S.out_index[row] : (long)row;
  if (S.c_out) S.c_out[orow * S.c_out_ld + u] = c;
  if (S.h_out_f32) S.h_out_f32[orow * S.h_out_f32_ld + u] = h;
  if (S.h_out_bf16) S.h_out_bf16[(long)row * S.h_out_bf16_ld + u] = f2bf(h);
  if (S.hd_out_bf16) {
    const float mk = S.hmask ? S.hmask[(long)row * S.hmask_ld + u] : 1.f;
    S.hd_out_bf16[(long)row * S.hd_out_ld + u] = f2bf(h * mk);
  }
  if (S.gates_out) {
    bf16_t* gp = S.gates_out + (long)row * S.gates_ld + u;
    gp[0] = f2bf(gi);
    gp[W] = f2bf(gf);
    gp[2 * W] = f2bf(gg);
    gp[3 * W] = f2bf(go);
  }
}

// ---------------------------------------------------------------- backward
// block = 256 threads; blockIdx.x = group of 16 hidden units; blockIdx.y = block
// of 16 rows; blockIdx.z = fused step.
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(const BwdPack pack) {
  const KlBwdStep& S = pack.s[blockIdx.z];
  const int r0 = blockIdx.y * 16;
  if (r0 >= S.n_rows) return;
  const int W = S.W;
  const int u0 = blockIdx.x * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float part[4][16][17];

  f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
  const int n = lane & 15;
  for (int p = 0; p < S.n_ops; ++p) {
    const KlOperand& op = S.op[p];
    long arow[1];
    int r = min(r0 + (lane & 15), S.n_rows - 1);
    arow[0] = op.row_index ? (long)op.row_index[r] : (long)r;
    accumulate<1>(op, 1, arow, (long)(u0 + n), wave, lane, acc);
    if (p == 0 && S.op0_mask) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int mr = min(r0 + (lane >> 4) * 4 + rr, S.n_rows - 1);
        acc[0][rr] *= S.op0_mask[(long)mr * S.op0_mask_ld + u0 + n];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) part[wave][(lane >> 4) * 4 + r][n] = acc[0][r];
  __syncthreads();
  const int lr = tid >> 4, j = tid & 15;
  const int row = r0 + lr;
  if (row >= S.n_rows) return;
  const int u = u0 + j;
  float dh = part[0][lr][j] + part[1][lr][j] + part[2][lr][j] + part[3][lr][j];
  if (S.dh_in) {
    float d = S.dh_in[(long)row * S.dh_in_ld + u];
    if (S.dh_mask) d *= S.dh_mask[(long)row * S.dh_mask_ld + u];
    dh += d;
  }
  const bf16_t* gp = S.gates + (long)row * S.gates_ld + u;
  const float gi = bf2f(gp[0]), gf = bf2f(gp[W]), gg = bf2f(gp[2 * W]), go = bf2f(gp[3 * W]);
  const float c = S.c[(long)row * S.c_ld + u];
  const float cp = S.c_prev ? S.c_prev[(long)row * S.c_prev_ld + u] : 0.f;
  const float tc = tanhf_(c);
  float dc = dh * go * (1.f - tc * tc);
  if (S.dc_in) dc += S.dc_in[(long)row * S.dc_in_ld + u];
  const float d_o = dh * tc;
  const float d_i = dc * gg, d_g = dc * gi, d_f = dc * cp;
  bf16_t* zp = S.dz_out + (long)row * S.dz_ld + u;
  zp[0] = f2bf(d_i * gi * (1.f - gi));
  zp[W] = f2bf(d_f * gf * (1.f - gf));
  zp[2 * W] = f2bf(d_g * (1.f - gg * gg));
  zp[3 * W] = f2bf(d_o * go * (1.f - go));
  if (S.dc_out) S.dc_out[(long)row * S.dc_out_ld + u] = dc * gf;
}

// ---------------------------------------------------------------- thin GEMM
// C[M,N] = A[M,K] . WT[N,K]^T (+bias[N]) with optional split precision.
// block = 256 threads (4 waves split K); tile 32 rows x 16 cols.
__global__ __launch_bounds__(256) void thin_gemm_kernel(const KlOperand op, int M, int N, float* C, long ldc,
                                                        const float* bias, int split) {
  const int r0 = blockIdx.y * 32;
  const int n0 = blockIdx.x * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float part[4][2][16][17];
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const int n = lane & 15;
  const long wt_row = min(n0 + n, N - 1);
  long arow[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int r = min(r0 + mt * 16 + (lane & 15), M - 1);
    arow[mt] = op.row_index ? (long)op.row_index[r] : (long)r;
  }
  accumulate<2>(op, split, arow, wt_row, wave, lane, acc);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][mt][(lane >> 4) * 4 + r][n] = acc[mt][r];
  __syncthreads();
  for (int e = tid; e < 32 * 16; e += 256) {
    const int lr = e >> 4, j = e & 15;
    const int row = r0 + lr, col = n0 + j;
    if (row >= M || col >= N) continue;
    float v = part[0][lr >> 4][lr & 15][j] + part[1][lr >> 4][lr & 15][j] + part[2][lr >> 4][lr & 15][j] +
              part[3][lr >> 4][lr & 15][j];
    if (bias) v += bias[col];
    C[(long)row * ldc + col] = v;
  }
}

bool operand_ok(const KlOperand& op) {
  if (op.K <= 0 || (op.K & 31) || op.ldw < op.K || (op.ldw & 7)) return false;
  if (op.a_is_f32 ? (op.lda & 3) : (op.lda & 7)) return false;
  return op.A != nullptr && op.WT_hi != nullptr;
}

}  // namespace

int kl_launch_fwd_steps(const KlFwdStep* steps, int n_steps, hipStream_t stream) {
  if (n_steps < 1 || n_steps > MAX_FUSED) return KL_ERR_ARG;
  FwdPack pack;
  int max_rows = 0, W = steps[0].W;
  for (int i = 0; i < n_steps; ++i) {
    const KlFwdStep& s = steps[i];
    if (s.W != W || (W & 3) || s.n_rows < 1 || s.n_ops < 0 || s.n_ops > 2) return KL_ERR_SHAPE;
    for (int p = 0; p < s.n_ops; ++p)
      if (!operand_ok(s.op[p])) return KL_ERR_SHAPE;
    pack.s[i] = s;
    if (s.n_rows > max_rows) max_rows = s.n_rows;
  }
  dim3 grid(W / 4, (max_rows + 31) / 32, n_steps);
  hipLaunchKernelGGL(lstm_fwd_step_kernel, grid, dim3(256), 0, stream, pack);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

int kl_launch_bwd_steps(const KlBwdStep* steps, int n_steps, hipStream_t stream) {
  if (n_steps < 1 || n_steps > MAX_FUSED) return KL_ERR_ARG;
  BwdPack pack;
  int max_rows = 0, W = steps[0].W;
  for (int i = 0; i < n_steps; ++i) {
    const KlBwdStep& s = steps[i];
    if (s.W != W || (W & 15) || s.n_rows < 1 || s.n_ops < 0 || s.n_ops > 2) return KL_ERR_SHAPE;
    for (int p = 0; p < s.n_ops; ++p)
      if (!operand_ok(s.op[p]) || s.op[p].a_is_f32) return KL_ERR_SHAPE;
    pack.s[i] = s;
    if (s.n_rows > max_rows) max_rows = s.n_rows;
  }
  dim3 grid(W / 16, (max_rows + 15) / 16, n_steps);
  hipLaunchKernelGGL(lstm_bwd_step_kernel, grid, dim3(256), 0, stream, pack);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

int kl_launch_thin_gemm(const KlOperand* op, int M, int N, float* C, long ldc, const float* bias, int split,
                        hipStream_t stream) {
  if (M < 1 || N < 1 || !operand_ok(*op)) return KL_ERR_SHAPE;
  dim3 grid((N + 15) / 16, (M + 31) / 32, 1);
  hipLaunchKernelGGL(thin_gemm_kernel, grid, dim3(256), 0, stream, *op, M, N, C, ldc, bias, split);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}
