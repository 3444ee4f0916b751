// Thin fused LSTM cell steps for gfx950.
//
// The recurrence is a chain of T tiny dependent contractions: per step the
// arithmetic is microseconds of MFMA work spread over the chip, so the step time
// is set by memory LATENCY, not by flops or bytes.  Design rules that follow:
//   * one launch per time step (a kernel boundary is the cheapest chip-wide
//     barrier on this part), several independent steps -- the layer wavefront,
//     layer l at time d-l -- packed into one launch through blockIdx.z; the
//     caller captures the chain in a hipGraph;
//   * thin workgroups (forward: 32 rows x 4 hidden units x 4 gates = one 16-wide
//     MFMA column tile; backward: 16 rows x 16 units) so that the weight slice
//     and the state rows of a step are spread over all 256 CUs;
//   * K is split over the waves of a workgroup and EVERY global load of a wave
//     (weight fragments, state fragments, and the epilogue's gate inputs) is
//     issued before the first MFMA, so a step exposes about one memory latency.
//     To keep the compiler from serialising them, the loads are unconditional:
//     the launcher replaces absent inputs by pointers into a zero page (stride 0)
//     and out-of-range k-steps load k-step 0 and are zeroed by a select;
//   * partial tiles meet in LDS, the gate math runs on the reduced tile.
//
// Restates: Keras LSTM cell at rating.py:126-145 (gate order i,f,c,o), the
// incremental step of Rater.predict rating.py:578-639 (rows = hypotheses, state
// rows addressed through slot indices), and TF autodiff of the same cell.
#include <string.h>

#include "kl_common.h"
#include "kl_kernels.h"

namespace {

constexpr int MAX_FUSED = 4;
constexpr int KS = 4;               // k-steps (of 32) a wave loads per phase and operand
constexpr int ZERO_WORDS = 1 << 15; // zero page: 128 KiB, covers 4W floats for W <= 8192

__device__ float kl_zero_page[ZERO_WORDS];

// In-kernel stamps: diagnostic build only (-DKL_STAMP, tools/probe_step.py); the
// shipped library contains no stamp code.
#ifdef KL_STAMP
__device__ unsigned long long kl_stamps[16];
#define STAMP(i)                                                                              \
  do {                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    if (blockIdx.x == 1 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) kl_stamps[i] = clock64(); \
    __builtin_amdgcn_sched_barrier(0);                                                        \
  } while (0)
#define STAMP_WAIT() __builtin_amdgcn_s_waitcnt(0)
#else
#define STAMP(i)
#define STAMP_WAIT()
#endif

struct FwdPack { KlFwdStep s[MAX_FUSED]; };
struct BwdPack { KlBwdStep s[MAX_FUSED]; };

template <bool AF32>
struct AFrag {           // raw registers of one lane's A fragment (8 consecutive k)
  uint4 r[AF32 ? 2 : 1];
};

template <bool AF32>
__device__ __forceinline__ void load_araw(const void* A, long lda, long row, int k, AFrag<AF32>& f) {
  if (AF32) {
    const float* p = reinterpret_cast<const float*>(A) + row * lda + k;
    f.r[0] = *reinterpret_cast<const uint4*>(p);
    f.r[AF32 ? 1 : 0] = *reinterpret_cast<const uint4*>(p + 4);
  } else {
    const bf16_t* p = reinterpret_cast<const bf16_t*>(A) + row * lda + k;
    f.r[0] = *reinterpret_cast<const uint4*>(p);
  }
}

template <bool AF32, bool LO>
__device__ __forceinline__ void split_a(const AFrag<AF32>& f, bf16x8& hi, bf16x8& lo) {
  frag16 h, l;
  if (AF32) {
    const uint32_t w[8] = {f.r[0].x, f.r[0].y, f.r[0].z, f.r[0].w,
                           f.r[AF32 ? 1 : 0].x, f.r[AF32 ? 1 : 0].y, f.r[AF32 ? 1 : 0].z, f.r[AF32 ? 1 : 0].w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = __builtin_bit_cast(float, w[j]);
      h.s[j] = f2bf(x);
      l.s[j] = LO ? f2bf(x - bf2f(h.s[j])) : (bf16_t)0;
    }
  } else {
    h.u = f.r[0];
    l.u = uint4{0, 0, 0, 0};
  }
  hi = h.v;
  lo = l.v;
}

__device__ __forceinline__ uint4 sel0(bool keep, uint4 v) {
  return uint4{keep ? v.x : 0u, keep ? v.y : 0u, keep ? v.z : 0u, keep ? v.w : 0u};
}

// One phase: this wave's KS k-steps of both operands.  load() issues every global
// load unconditionally; mma() consumes them.
template <int NMT, int NWAVES, bool AF32, bool LO>
struct Phase {
  uint4 bh[2][KS], bl[2][KS];
  AFrag<AF32> a[2][KS][NMT];

  __device__ __forceinline__ void load(const KlOperand (&ops)[2], const int (&nks)[2], const long (&arow)[2][NMT],
                                       long wt_row, int base, int wave, int kq) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const bf16_t* wh = ops[p].WT_hi + wt_row * ops[p].ldw;
      const bf16_t* wl = (LO ? ops[p].WT_lo : ops[p].WT_hi) + wt_row * ops[p].ldw;
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        const int ks = base + wave + NWAVES * j;
        const int k = (ks < nks[p] ? ks : 0) * 32 + kq;   // out-of-range steps re-read step 0
        bh[p][j] = *reinterpret_cast<const uint4*>(wh + k);
        if (LO) bl[p][j] = *reinterpret_cast<const uint4*>(wl + k);
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) load_araw<AF32>(ops[p].A, ops[p].lda, arow[p][mt], k, a[p][j][mt]);
      }
    }
  }

  __device__ __forceinline__ void mma(const int (&nks)[2], int base, int wave, f32x4 (&acc0)[NMT], f32x4 (&acc1)[NMT]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        const bool valid = (base + wave + NWAVES * j) < nks[p];
        frag16 fbh, fbl;
        fbh.u = sel0(valid, bh[p][j]);       // a zero B fragment contributes nothing
        if (LO) fbl.u = sel0(valid, bl[p][j]);
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
          bf16x8 ah, al;
          split_a<AF32, LO>(a[p][j][mt], ah, al);
          f32x4& dst = (p == 0) ? acc0[mt] : acc1[mt];
          dst = mfma16(ah, fbh.v, dst);
          if (LO) {
            dst = mfma16(ah, fbl.v, dst);
            if (AF32) dst = mfma16(al, fbh.v, dst);
          }
        }
      }
    }
  }
};

// ---------------------------------------------------------------- forward
// block = 256 threads (4 waves split K); blockIdx.x = group of 4 hidden units;
// blockIdx.y = block of 32 rows; blockIdx.z = fused step.  The 16 MFMA columns
// are gate*4 + unit.
template <bool AF32, bool LO>
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(const FwdPack pack) {
  STAMP(0);
  const KlFwdStep S = pack.s[blockIdx.z];
  const int r0 = blockIdx.y * 32;
  if (r0 >= S.n_rows) return;
  STAMP(1);
  const int W = S.W;
  const int u0 = blockIdx.x * 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float part[4][2][16][17];

  // ---- addresses (row gathers first, they feed the other loads)
  const int lr = (tid >> 2) & 31, ej = tid & 3;
  const int erow = min(r0 + lr, S.n_rows - 1), eu = u0 + ej;
  const bool e_on = tid < 128 && (r0 + lr) < S.n_rows;
  int r1 = erow, r2 = erow, rc = erow, ro = erow;
  if (S.i1) r1 = S.i1[erow];
  if (S.i2) r2 = S.i2[erow];
  if (S.c_prev_index) rc = S.c_prev_index[erow];
  if (S.out_index) ro = S.out_index[erow];
  const int n = lane & 15;
  const long wt_row = (long)(n >> 2) * W + u0 + (n & 3);
  long arow[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int r = min(r0 + mt * 16 + (lane & 15), S.n_rows - 1);
      arow[p][mt] = S.op[p].row_index ? (long)S.op[p].row_index[r] : (long)r;
    }
  }
  const int kq = (lane >> 4) * 8;
  const int nks[2] = {S.n_ops > 0 ? S.op[0].K >> 5 : 0, S.n_ops > 1 ? S.op[1].K >> 5 : 0};
  const int maxk = nks[0] > nks[1] ? nks[0] : nks[1];

  // ---- all loads of the first phase + the epilogue operands, then the MFMAs
  STAMP(2);
  Phase<2, 4, AF32, LO> ph;
  ph.load(S.op, nks, arow, wt_row, 0, wave, kq);
  float t1v[4], t2v[4], bv[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const long col = (long)g * W + eu;
    t1v[g] = S.T1[(long)r1 * S.t1_ld + col];
    t2v[g] = S.T2[(long)r2 * S.t2_ld + col];
    bv[g] = S.bias[col];
  }
  const float cp = S.c_prev[(long)rc * S.c_prev_ld + eu];
  float mk = 1.f;
  if (S.hmask) mk = S.hmask[(long)erow * S.hmask_ld + eu];

  STAMP(3);
  STAMP_WAIT();
  STAMP(4);
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  ph.mma(nks, 0, wave, acc, acc);
  for (int base = 4 * KS; base < maxk; base += 4 * KS) {
    ph.load(S.op, nks, arow, wt_row, base, wave, kq);
    ph.mma(nks, base, wave, acc, acc);
  }
  STAMP(5);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][mt][(lane >> 4) * 4 + r][n] = acc[mt][r];
  __syncthreads();
  STAMP(6);
  if (!e_on) return;
  float z[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float v = t1v[g] + t2v[g] + bv[g];
#pragma unroll
    for (int w = 0; w < 4; ++w) v += part[w][lr >> 4][lr & 15][g * 4 + ej];
    z[g] = v;
  }
  const float gi = sigmoidf_(z[0]), gf = sigmoidf_(z[1]), gg = tanhf_(z[2]), go = sigmoidf_(z[3]);
  const float c = gf * cp + gi * gg;
  const float h = go * tanhf_(c);
  STAMP(7);
  if (S.c_out) S.c_out[(long)ro * S.c_out_ld + eu] = c;
  if (S.h_out_f32) S.h_out_f32[(long)ro * S.h_out_f32_ld + eu] = h;
  if (S.h_out_bf16) S.h_out_bf16[(long)erow * S.h_out_bf16_ld + eu] = f2bf(h);
  if (S.hd_out_bf16) S.hd_out_bf16[(long)erow * S.hd_out_ld + eu] = f2bf(h * mk);
  if (S.gates_out) {
    bf16_t* gp = S.gates_out + (long)erow * S.gates_ld + eu;
    gp[0] = f2bf(gi);
    gp[W] = f2bf(gf);
    gp[2 * W] = f2bf(gg);
    gp[3 * W] = f2bf(go);
  }
  STAMP(8);
}

// ---------------------------------------------------------------- backward
// block = 1024 threads (16 waves split K = 4W); blockIdx.x = group of 16 hidden
// units; blockIdx.y = block of 16 rows; blockIdx.z = fused step.
__global__ __launch_bounds__(1024) void lstm_bwd_step_kernel(const BwdPack pack) {
  const KlBwdStep S = pack.s[blockIdx.z];
  const int r0 = blockIdx.y * 16;
  if (r0 >= S.n_rows) return;
  const int W = S.W;
  const int u0 = blockIdx.x * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float part[16][16][17];

  const int lr = (tid >> 4) & 15, ej = tid & 15;
  const int erow = min(r0 + lr, S.n_rows - 1), eu = u0 + ej;
  const bool e_on = tid < 256 && (r0 + lr) < S.n_rows;
  const int n = lane & 15;
  long arow[2][1];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int r = min(r0 + (lane & 15), S.n_rows - 1);
    arow[p][0] = S.op[p].row_index ? (long)S.op[p].row_index[r] : (long)r;
  }
  const int kq = (lane >> 4) * 8;
  const int nks[2] = {S.n_ops > 0 ? S.op[0].K >> 5 : 0, S.n_ops > 1 ? S.op[1].K >> 5 : 0};
  const int maxk = nks[0] > nks[1] ? nks[0] : nks[1];

  Phase<1, 16, false, false> ph;
  ph.load(S.op, nks, arow, (long)(u0 + n), 0, wave, kq);
  // epilogue operands (canonical pointers: absent inputs read the zero page)
  const bf16_t* gp = S.gates + (long)erow * S.gates_ld + eu;
  const bf16_t g0 = gp[0], g1 = gp[W], g2 = gp[2 * W], g3 = gp[3 * W];
  const float c = S.c[(long)erow * S.c_ld + eu];
  const float cp = S.c_prev[(long)erow * S.c_prev_ld + eu];
  const float dcin = S.dc_in[(long)erow * S.dc_in_ld + eu];
  float dhin = S.dh_in[(long)erow * S.dh_in_ld + eu];
  float dmask = 1.f;
  if (S.dh_mask) dmask = S.dh_mask[(long)erow * S.dh_mask_ld + eu];
  float omask[4] = {1.f, 1.f, 1.f, 1.f};
  if (S.op0_mask) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int mr = min(r0 + (lane >> 4) * 4 + rr, S.n_rows - 1);
      omask[rr] = S.op0_mask[(long)mr * S.op0_mask_ld + u0 + n];
    }
  }

  f32x4 acc0[1] = {f32x4{0.f, 0.f, 0.f, 0.f}}, acc1[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
  ph.mma(nks, 0, wave, acc0, acc1);
  for (int base = 16 * KS; base < maxk; base += 16 * KS) {
    ph.load(S.op, nks, arow, (long)(u0 + n), base, wave, kq);
    ph.mma(nks, base, wave, acc0, acc1);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) part[wave][(lane >> 4) * 4 + r][n] = acc0[0][r] * omask[r] + acc1[0][r];
  __syncthreads();
  if (!e_on) return;
  float dh = dhin * dmask;
#pragma unroll
  for (int w = 0; w < 16; ++w) dh += part[w][lr][ej];
  const float gi = bf2f(g0), gf = bf2f(g1), gg = bf2f(g2), go = bf2f(g3);
  const float tc = tanhf_(c);
  const float dc = dh * go * (1.f - tc * tc) + dcin;
  const float d_o = dh * tc;
  const float d_i = dc * gg, d_g = dc * gi, d_f = dc * cp;
  bf16_t* zp = S.dz_out + (long)erow * S.dz_ld + eu;
  zp[0] = f2bf(d_i * gi * (1.f - gi));
  zp[W] = f2bf(d_f * gf * (1.f - gf));
  zp[2 * W] = f2bf(d_g * (1.f - gg * gg));
  zp[3 * W] = f2bf(d_o * go * (1.f - go));
  if (S.dc_out) S.dc_out[(long)erow * S.dc_out_ld + eu] = dc * gf;
}

// ---------------------------------------------------------------- thin GEMM
// C[M,N] = A[M,K] . WT[N,K]^T (+bias[N]); block = 256 threads (4 waves split K);
// tile 32 rows x 16 cols.
struct ThinArgs { KlOperand op[2]; };

template <bool AF32, bool LO>
__global__ __launch_bounds__(256) void thin_gemm_kernel(const ThinArgs args, int M, int N, float* C, long ldc,
                                                        const float* bias) {
  const int r0 = blockIdx.y * 32;
  const int n0 = blockIdx.x * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ float part[4][2][16][17];
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
  const int n = lane & 15;
  const long wt_row = min(n0 + n, N - 1);
  long arow[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int r = min(r0 + mt * 16 + (lane & 15), M - 1);
    arow[0][mt] = args.op[0].row_index ? (long)args.op[0].row_index[r] : (long)r;
    arow[1][mt] = arow[0][mt];
  }
  const int kq = (lane >> 4) * 8;
  const int nks[2] = {args.op[0].K >> 5, 0};
  Phase<2, 4, AF32, LO> ph;
  for (int base = 0; base < nks[0]; base += 4 * KS) {
    ph.load(args.op, nks, arow, wt_row, base, wave, kq);
    ph.mma(nks, base, wave, acc, acc);
  }
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
  if (op.K <= 0 || (op.K & 31) || (op.ldw != 0 && op.ldw < op.K) || (op.ldw & 7)) return false;
  if (op.a_is_f32 ? (op.lda & 3) : (op.lda & 7)) return false;
  return op.A != nullptr && op.WT_hi != nullptr;
}

// operand-type signature of a step: all operands must share it
int step_kind(const KlOperand* ops, int n_ops, int split, int* af32, int* lo) {
  if (n_ops == 0) { *af32 = 0; *lo = 0; return 0; }
  *af32 = ops[0].a_is_f32;
  *lo = (split == 3);
  for (int p = 0; p < n_ops; ++p) {
    if (!operand_ok(ops[p])) return KL_ERR_SHAPE;
    if (ops[p].a_is_f32 != *af32) return KL_ERR_SHAPE;
    if (*lo && ops[p].WT_lo == nullptr) return KL_ERR_ARG;
  }
  return 0;
}

float* zero_page() {
  static float* p = nullptr;
  if (!p) {
    void* q = nullptr;
    if (hipGetSymbolAddress(&q, HIP_SYMBOL(kl_zero_page)) != hipSuccess) return nullptr;
    p = reinterpret_cast<float*>(q);
  }
  return p;
}

// operands a step does not use still get loaded (k-step 0, zeroed by select):
// point them at a valid operand or at the zero page
void canon_ops(KlOperand* op, int n_ops, const float* zp) {
  if (n_ops == 0) {
    memset(&op[0], 0, sizeof(KlOperand));
    op[0].A = zp; op[0].lda = 0; op[0].WT_hi = reinterpret_cast<const bf16_t*>(zp); op[0].WT_lo = op[0].WT_hi;
    op[0].ldw = 0; op[0].K = 32; op[0].a_is_f32 = 0;
  }
  if (n_ops < 2) op[1] = op[0];
}

}  // namespace

int kl_zero_page_ready() { return zero_page() != nullptr ? 0 : KL_ERR_LAUNCH; }

#ifdef KL_STAMP
extern "C" int kl_test_read_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(kl_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : KL_ERR_LAUNCH;
}
#endif

int kl_launch_fwd_steps(const KlFwdStep* steps, int n_steps, hipStream_t stream) {
  if (n_steps < 1 || n_steps > MAX_FUSED) return KL_ERR_ARG;
  const float* zp = zero_page();
  if (!zp) return KL_ERR_LAUNCH;
  FwdPack pack;
  memset(&pack, 0, sizeof(pack));
  int max_rows = 0, W = steps[0].W;
  int af32 = -1, lo = -1;
  for (int i = 0; i < n_steps; ++i) {
    KlFwdStep s = steps[i];
    if (s.W != W || (W & 3) || 4L * W > ZERO_WORDS || s.n_rows < 1 || s.n_ops < 1 || s.n_ops > 2) return KL_ERR_SHAPE;
    int a, l;
    const int e = step_kind(s.op, s.n_ops, s.split, &a, &l);
    if (e) return e;
    if (af32 < 0) { af32 = a; lo = l; }
    else if (af32 != a || lo != l) return KL_ERR_SHAPE;
    canon_ops(s.op, s.n_ops, zp);
    if (!s.T1) { s.T1 = zp; s.t1_ld = 0; s.i1 = nullptr; }
    if (!s.T2) { s.T2 = zp; s.t2_ld = 0; s.i2 = nullptr; }
    if (!s.bias) s.bias = zp;
    if (!s.c_prev) { s.c_prev = zp; s.c_prev_ld = 0; s.c_prev_index = nullptr; }
    pack.s[i] = s;
    if (s.n_rows > max_rows) max_rows = s.n_rows;
  }
  dim3 grid(W / 4, (max_rows + 31) / 32, n_steps);
  if (af32 && lo) hipLaunchKernelGGL((lstm_fwd_step_kernel<true, true>), grid, dim3(256), 0, stream, pack);
  else if (af32) hipLaunchKernelGGL((lstm_fwd_step_kernel<true, false>), grid, dim3(256), 0, stream, pack);
  else if (lo) hipLaunchKernelGGL((lstm_fwd_step_kernel<false, true>), grid, dim3(256), 0, stream, pack);
  else hipLaunchKernelGGL((lstm_fwd_step_kernel<false, false>), grid, dim3(256), 0, stream, pack);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

int kl_launch_bwd_steps(const KlBwdStep* steps, int n_steps, hipStream_t stream) {
  if (n_steps < 1 || n_steps > MAX_FUSED) return KL_ERR_ARG;
  const float* zp = zero_page();
  if (!zp) return KL_ERR_LAUNCH;
  BwdPack pack;
  memset(&pack, 0, sizeof(pack));
  int max_rows = 0, W = steps[0].W;
  for (int i = 0; i < n_steps; ++i) {
    KlBwdStep s = steps[i];
    if (s.W != W || (W & 15) || 4L * W > ZERO_WORDS || s.n_rows < 1 || s.n_ops < 0 || s.n_ops > 2) return KL_ERR_SHAPE;
    for (int p = 0; p < s.n_ops; ++p)
      if (!operand_ok(s.op[p]) || s.op[p].a_is_f32) return KL_ERR_SHAPE;
    if (!s.gates || !s.c || !s.dz_out) return KL_ERR_ARG;
    canon_ops(s.op, s.n_ops, zp);
    if (!s.c_prev) { s.c_prev = zp; s.c_prev_ld = 0; }
    if (!s.dc_in) { s.dc_in = zp; s.dc_in_ld = 0; }
    if (!s.dh_in) { s.dh_in = zp; s.dh_in_ld = 0; }
    pack.s[i] = s;
    if (s.n_rows > max_rows) max_rows = s.n_rows;
  }
  dim3 grid(W / 16, (max_rows + 15) / 16, n_steps);
  hipLaunchKernelGGL(lstm_bwd_step_kernel, grid, dim3(1024), 0, stream, pack);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

int kl_launch_thin_gemm(const KlOperand* op, int M, int N, float* C, long ldc, const float* bias, int split,
                        hipStream_t stream) {
  int af32, lo;
  if (M < 1 || N < 1) return KL_ERR_SHAPE;
  const int e = step_kind(op, 1, split, &af32, &lo);
  if (e) return e;
  ThinArgs args;
  args.op[0] = *op;
  args.op[1] = *op;
  dim3 grid((N + 15) / 16, (M + 31) / 32, 1);
  if (af32 && lo) hipLaunchKernelGGL((thin_gemm_kernel<true, true>), grid, dim3(256), 0, stream, args, M, N, C, ldc, bias);
  else if (af32) hipLaunchKernelGGL((thin_gemm_kernel<true, false>), grid, dim3(256), 0, stream, args, M, N, C, ldc, bias);
  else if (lo) hipLaunchKernelGGL((thin_gemm_kernel<false, true>), grid, dim3(256), 0, stream, args, M, N, C, ldc, bias);
  else hipLaunchKernelGGL((thin_gemm_kernel<false, false>), grid, dim3(256), 0, stream, args, M, N, C, ldc, bias);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}
