// Internal launcher interface between the C-ABI (api.hip) and the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "keraslm_hip.h"   // error codes + public ABI (include/)

typedef unsigned short bf16_t;

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the function ON A DEVICE: a launcher that needs more than 64 KiB
// of dynamic LDS keeps, per kernel instantiation, what each device has been granted so far (one `static KlLdsGrant` per
// instantiation) and sets the attribute only when a launch needs more -- the call costs microseconds, the incremental
// step is a handful of them.  Safe against two threads stepping at once (the slow path is serialised).
#include <atomic>
#include <mutex>
#define KL_MAX_DEVICES 64
struct KlLdsGrant {
  std::atomic<size_t> granted[KL_MAX_DEVICES];
  std::mutex slow;
};
static inline int kl_grant_lds(KlLdsGrant& g, const void* fn, size_t lds) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return KL_ERR_LAUNCH;
  if (dev < 0 || dev >= KL_MAX_DEVICES)      // (no room to remember: set it every time)
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 0 : KL_ERR_LAUNCH;
  if (lds <= g.granted[dev].load(std::memory_order_acquire)) return 0;
  std::lock_guard<std::mutex> lock(g.slow);
  if (lds <= g.granted[dev].load(std::memory_order_relaxed)) return 0;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return KL_ERR_LAUNCH;
  g.granted[dev].store(lds, std::memory_order_release);
  return 0;
}

// ---- gemm.hip -----------------------------------------------------------
// recurrent halves of all layers' activation rows for the fused incremental step (step_big.hip)
struct KlGatherRec {
  const float* pool; long slot_ld; const int* slot_in;
  int n, W, nb;
  bf16_t* out[4];                      // per layer [n][3 * K_l], K_0 = W, K_l = 2W
};
int kl_launch_gather_recurrent(const KlGatherRec& a, int L, hipStream_t stream);
int kl_launch_permute_gate_rows(const bf16_t* src, bf16_t* dst, int W, long cols, hipStream_t stream);

// LSTM cell epilogue of the fused incremental-step GEMM (gemm.hip OUT == 3)
struct KlGateEpi {
  const float* T1; const int* i1;      // table rows added to z: T1[i1 ? i1[row] : row][4W] (null: none)
  const float* T2; const int* i2;
  const float* bias;                   // [4W] (null: none)
  const float* c_prev; long c_ld; const int* slot_in;   // c_prev[slot_in[row] * c_ld + u]
  float* c_out; float* h_out; long out_ld; const int* slot_out;
  bf16_t* xn; long ldn; int kn; int nbn;   // next layer's activation rows: h' as (hi | lo | hi) at block stride kn (null: none)
  int W;
};
int kl_launch_gemm_gates(const bf16_t* A3, const bf16_t* WTperm, int n, int W, int Kc, long lda, const KlGateEpi* epi,
                         hipStream_t stream);

int kl_launch_gemm_tn(const bf16_t* A, const bf16_t* B, void* C, const float* bias, int M, int N, int K,
                      long lda, long ldb, long ldc, int out_mode, int splits, float alpha, hipStream_t stream);
// the same contraction with a K-major A operand [K][M] (split-K, f32 atomics; optionally C^T): gemm.hip
bool kl_gemm_an_applicable(int M, int N, int K, long lda_km);
int kl_launch_gemm_an(const bf16_t* A_km, const bf16_t* B, float* C, int M, int N, int K, long lda_km, long ldb, long ldc,
                      int c_transposed, hipStream_t stream, int b_km = 0);
// ... and a second product over the same A in the same launch: C2 (+)= A^T-view . B2^T (N a multiple of 128; see KlGemmSecond)
int kl_launch_gemm_an2(const bf16_t* A_km, const bf16_t* B, float* C, int M, int N, int K, long lda_km, long ldb, long ldc,
                       int c_transposed, const bf16_t* B2, float* C2, int N2, long ldb2, long ldc2, int c_transposed2,
                       hipStream_t stream, int b_km);

// ---- lstm_step.hip ------------------------------------------------------
// One (activation, weight) operand pair of a thin fused step: rows of A are
// contracted with rows of WT (both K-contiguous).
struct KlOperand {
  const void* A;         // [rows][K] f32 (a_is_f32) or bf16
  long lda;              // elements
  const int* row_index;  // optional gather of A rows (state-pool slots); null = identity
  const bf16_t* WT_hi;   // [n_out_rows][K] bf16
  const bf16_t* WT_lo;   // residual (split mode) or null
  long ldw;              // row stride of WT (elements, >= K)
  int K;
  int a_is_f32;
};

// forward cell step (F2+F3 fused; S1 when rows are hypotheses)
struct KlFwdStep {
  KlOperand op[2];
  int n_ops;
  int n_rows, W;
  int split;                 // 1 = bf16, 3 = split-bf16 (hi*hi + lo*hi + hi*lo)
  const float* T1; const int* i1; long t1_ld;   // z init: T1[i1[r]] (i1 null = identity rows)
  const float* T2; const int* i2; long t2_ld;   // + T2[i2[r]]
  const float* bias;                            // + bias[4W]
  const float* c_prev; long c_prev_ld; const int* c_prev_index;
  const int* out_index;                         // scatter of output rows (slots); null = identity
  float* c_out; long c_out_ld;
  float* h_out_f32; long h_out_f32_ld;
  bf16_t* h_out_bf16; long h_out_bf16_ld;
  bf16_t* gates_out; long gates_ld;             // [rows][4W] post-activation i,f,g,o (training)
  const float* hmask; long hmask_ld;            // optional dropout keep-mask (scaled) for the copy below
  bf16_t* hd_out_bf16; long hd_out_ld;          // h * hmask, what the layer above / softmax consumes
};
int kl_launch_fwd_steps(const KlFwdStep* steps, int n_steps, hipStream_t stream);

// backward cell step (B3 fused: dh = dh_in + sum_p A_p . WT_p^T ; gate derivatives)
struct KlBwdStep {
  KlOperand op[2];
  int n_ops;
  int n_rows, W;
  const float* op0_mask; long op0_mask_ld;      // optional dropout mask on the op[0] contribution
  const float* dh_in; long dh_in_ld;            // gradient from above (softmax side), optional
  const float* dh_mask; long dh_mask_ld;        // optional dropout mask multiplying dh_in
  const bf16_t* gates; long gates_ld;           // [rows][4W] i,f,g,o of this step
  const float* c; long c_ld;                    // c_t
  const float* c_prev; long c_prev_ld;          // c_{t-1}
  const float* dc_in; long dc_in_ld;            // dc carried from t+1 (already times f_{t+1}), optional
  float* dc_out; long dc_out_ld;                // dc_t * f_t
  bf16_t* dz_out; long dz_ld;                   // [rows][4W]
};
int kl_launch_bwd_steps(const KlBwdStep* steps, int n_steps, hipStream_t stream);
int kl_zero_page_ready();   // resolves the zero page's device address (call outside stream capture)

// ---- lstm_scan.hip ------------------------------------------------------
// persistent scans (whole window, all layers, one launch; weights in registers)
#define KL_SCAN_MAXL 4
struct KlScanFwd {
  int B, T, W, L;
  int n_rb, n_rg;                      // filled by the launcher: row blocks of 16, row groups
  const bf16_t* UT[KL_SCAN_MAXL];      // [4W][W] recurrent kernels, transposed
  const bf16_t* KT[KL_SCAN_MAXL];      // [4W][W] input kernels, transposed (l >= 1)
  const float* bias[KL_SCAN_MAXL];     // [4W] (l >= 1; layer 0's bias is folded into P1)
  const float* P1;                     // [T*B][4W] layer-0 input contraction + bias
  bf16_t* H[KL_SCAN_MAXL];             // [(T+1)B][W], block 0 = carried-in state
  float* C[KL_SCAN_MAXL];              // [(T+1)B][W], block 0 = carried-in state
  bf16_t* G[KL_SCAN_MAXL];             // [T*B][4W] gate activations (null: not kept)
  bf16_t* Hd[KL_SCAN_MAXL];            // [T*B][W] dropout-masked outputs (null: none)
  const float* mask[KL_SCAN_MAXL];     // [B][W] keep-masks (null: none)
  unsigned* counters;                  // [L][n_rb][T], zeroed by the launcher
  unsigned* status;                    // 0 = ok, 1 = a bounded spin timed out
  int sentinel;                        // width-1024 scan only: 1 = hand-off by data sentinels (blocks 1..T of H pre-filled with 0xFFFF halfwords)
  unsigned* xcc_slots; unsigned gen;   // width-1024 scan only: as KlScanFwdWide
};
int kl_launch_scan_fwd(KlScanFwd args, hipStream_t stream);
// width 1024, one layer per launch: eight-wave workgroups of 32 units, the tile through LDS (lstm_scan_w32.hip); KL_ERR_SHAPE = not applicable
bool kl_scan_w32_applicable(int B, int T, int W);
int kl_launch_scan_fwd_w32(KlScanFwd args, hipStream_t stream);

// split-precision (bf16 hi + lo) inference scan, all layers fused
struct KlScanFwdSplit {
  int B, T, W, L;
  int n_rb, n_rg;                      // filled by the launcher
  const bf16_t* UT_hi[KL_SCAN_MAXL]; const bf16_t* UT_lo[KL_SCAN_MAXL];   // [4W][W]
  const bf16_t* KT_hi[KL_SCAN_MAXL]; const bf16_t* KT_lo[KL_SCAN_MAXL];   // [4W][W] (l >= 1)
  const float* bias[KL_SCAN_MAXL];     // [4W] (l >= 1)
  const float* P1;                     // [T*B][4W] layer-0 input contraction + bias
  bf16_t* Xhi[KL_SCAN_MAXL]; bf16_t* Xlo[KL_SCAN_MAXL];   // [(T+1)B][W] exchanged state planes, block 0 = carried in
  float* Hf[KL_SCAN_MAXL];             // [(T+1)B][W] f32 outputs (blocks 1..T written)
  float* C[KL_SCAN_MAXL];              // [(T+1)B][W]: block 0 read, block T written
  unsigned* counters;                  // [L][n_rb][T]
  unsigned* status;
  int sentinel;                        // 1: hand-off by data -- blocks 1..T of Xhi / Xlo pre-filled with 0xFFFF by the caller, no counters
  int units8, l0;                      // width 1024: eight units per workgroup, layers l0 .. l0 + L - 1 (L <= 2) as a wavefront (the arrays are indexed by absolute layer)
};
int kl_launch_scan_fwd_split(KlScanFwdSplit args, hipStream_t stream);

// one layer per launch, 64-unit workgroups (B >= 512 streams)
struct KlScanFwdWide {
  int B, T, W;
  int n_rb, n_rg;                      // filled by the launcher
  const bf16_t* UT;                    // [4W][W]
  const float* P;                      // [T*B][4W] input contraction + bias (layers >= 1), or null = tables:
  const float* EK; const float* CtxK[8]; int n_ctx;   // layer 0: EK[idx] + sum_n CtxK_n[ctx_n] + bias
  const int* idx; const int* ctx;      // [B][T], [B][T][n_ctx]
  const float* bias;                   // [4W]
  bf16_t* H; float* C; bf16_t* G;      // as KlScanFwd, this layer
  bf16_t* Hd; const float* mask;       // dropout-masked copy (null: none)
  bf16_t* HT; long ldt;                // transposed outputs [W][ldt], column (t+1)*B + row (null: none)
  bf16_t* HdT; long ldt_d;             // transposed masked outputs [W][ldt_d], column t*B + row (null: none)
  unsigned* counters;                  // [n_rb][T] (counter hand-off)
  unsigned* status;
  int sentinel;                        // 1: hand-off by data -- H blocks 1..T pre-filled with 0xFFFF halfwords, no counters
  unsigned* xcc_slots; unsigned gen;   // [256] + launch token: XCD-local hand-off if the workgroups of a row group share an XCD (null: off)
  // second generation (lstm_scan2.hip): P, EK (+ bias), CtxK[0] and G are gate-interleaved ([row][unit][4 gates])
  const int* ids_tm;                   // [T+1][B][2] byte offsets of the EK / CtxK[0] rows, time-major (kl_launch_ids_tm)
  int V, ctx_vocab;                    // rows of EK / CtxK[0]
  int pf_mode;                         // where the next phase's tile is requested: 0 = top of a phase, 1 = behind the MFMA phase
  bf16_t* Cb;                          // second generation: cell states for the backward scan as bf16 [(T+1)B][W] (blocks 1..T written;
                                       // C then only receives block T, the carried-out state); null: every block goes to C in f32
  int p_bf16;                          // second generation: P is bf16 [T*B][W][4 gates] (8 bytes per cell) instead of f32
  const bf16_t* KT; const bf16_t* X;   // width-128 scan only: input kernel [4W][W] + input rows [T*B][W] (with `bias`): the input side inside the scan
};
// width 128: a workgroup = a 16-row block of streams with ALL hidden units of one layer, no hand-off between workgroups
// (lstm_scan_w128.hip); forward takes f32 P rows only; backward as the wide one-layer kernels (a.L == 1, db summed)
bool kl_scan_w128_applicable(int B, int T, int W);
int kl_launch_scan_fwd_w128(KlScanFwdWide args, hipStream_t stream);
int kl_scan_w128_tables_max_ctx();      // layer 0 straight from the look-up tables (args.P == null) with up to this many context variables
// ... all layers in one launch (layers x row blocks <= CUs, else KL_ERR_SHAPE), a layer polling the rows the one below publishes: the caller
// pre-fills those rows (layers[l].X, l > 0) with 0xFFFF halfwords
bool kl_scan_w128_multi_fits(int B, int L);      // 2 <= L <= 4 and L x 16-row blocks <= CUs
int kl_launch_scan_fwd_w128_multi(const KlScanFwdWide* layers, int L, hipStream_t stream);
bool kl_scan_fwd_wide_applicable(int B, int T, int W);
int kl_launch_scan_fwd_wide(KlScanFwdWide args, hipStream_t stream);
// second generation: every workgroup serves NP = 2..max_np phases of `rows` (16 or 32) rows per step; 0 = not applicable
// (forward scans: max_np 4; backward scan, always 16-row blocks: 6)
int kl_scan_wide2_phases(int B, int T, int W, int rows, int max_np);
int kl_launch_scan_fwd_wide2(KlScanFwdWide args, int rows, hipStream_t stream);
// ... 32-row phases on eight waves without workgroup barriers (lstm_scan_fwd8.hip): bf16 P rows, rolling sentinels; pf_mode 0..3
// (see the kernel); KL_ERR_SHAPE = not applicable
int kl_launch_scan_fwd8(KlScanFwdWide args, hipStream_t stream, bool lockstep = false);      // lockstep: two workgroup barriers per phase instead of the LDS counters
int kl_launch_permute_gate_cols_f32(const float* in, const float* bias, float* out, long rows, int W, hipStream_t stream);
int kl_launch_permute_gate_rows_bf16(const bf16_t* in, bf16_t* out, int W, int K, hipStream_t stream);
// layer 0's gate inputs as gate-interleaved bf16 P rows (time-major) from the permuted tables: several context variables
int kl_launch_p_gather_il(const float* EKp, const float* const* CtxKp, int n_ctx, const int* idx, const int* ctx, int B, int T, int W,
                          int V, int ctx_vocab, bf16_t* out, hipStream_t stream);
int kl_launch_ids_tm(const int* idx, const int* ctx, int n_ctx, int B, int T, int W, int V, int ctx_vocab, int* out,
                     hipStream_t stream);

struct KlScanBwd {
  int B, T, W, L;
  int n_rb, n_rg;
  const bf16_t* Un[KL_SCAN_MAXL];      // [W][4W] recurrent kernels, natural layout
  const bf16_t* Kn[KL_SCAN_MAXL];      // [W][4W] input kernels (rows of layer l's K; used by layer l-1)
  const bf16_t* G[KL_SCAN_MAXL];       // [T*B][4W]
  const float* C[KL_SCAN_MAXL];        // [(T+1)B][W]
  bf16_t* dZ[KL_SCAN_MAXL];            // [T*B][4W]
  const float* dH;                     // [T*B][W] gradient from the softmax (top layer)
  const float* mask[KL_SCAN_MAXL];     // dropout mask on the OUTPUT of layer l (null: none)
  unsigned* counters;                  // [L][n_rb][T]
  unsigned* status;
  bf16_t* dZT; long ldt;               // wide one-layer kernel only: also write dZ transposed [4W][ldt] (null: no)
  float* db;                           // wide one-layer kernel only: += column sums of dZ (bias gradient; null: no)
  float* db_l[KL_SCAN_MAXL];           // width-128 multi-layer launch only: the same per layer
  int sentinel;                        // wide one-layer kernel only: 1 = hand-off by data sentinels (dZ pre-filled with 0xFFFF halfwords)
  unsigned* xcc_slots; unsigned gen;   // as KlScanFwdWide (sentinel hand-off only)
  int pf_mode;                         // second generation: as KlScanFwdWide
  const bf16_t* dHb;                   // second generation: the gradient from above as bf16 [T*B][W] (dH unused)
  const bf16_t* Cb;                    // second generation: cell states as bf16 [(T+1)B][W], blocks 1..T (null: C, f32)
  unsigned* flags; const unsigned* epoch;   // second generation: hand-off by flags [n_rb][64] instead of sentinels (null: sentinels); *epoch: this launch's
};
int kl_launch_scan_epoch(unsigned* flags, int n_flags, unsigned* epoch, unsigned step, hipStream_t stream);   // epoch += step in front of a flag-mode scan
int kl_launch_scan_bwd(KlScanBwd args, hipStream_t stream);
bool kl_scan_bwd_wide_applicable(int B, int T, int W);
int kl_scan_wide_blocks_per_wg(int B, int W);
int kl_launch_scan_bwd_wide(KlScanBwd args, hipStream_t stream);   // one layer per launch, 64-unit workgroups
int kl_launch_scan_bwd_wide2(KlScanBwd args, hipStream_t stream);
// eight waves, two cells per thread, the tile through registers two blocks ahead (flags only; same arguments); from
// kl_scan_bwd_regtile_min_np() blocks per workgroup and step
int kl_launch_scan_bwd_regtile(KlScanBwd args, hipStream_t stream);
int kl_launch_scan_bwd_w32(KlScanBwd args, hipStream_t stream);     // width 1024 (a.sentinel must be 1, a.L 1)
int kl_launch_scan_bwd_w128(KlScanBwd args, hipStream_t stream);    // width 128 (a.L 1, a.dH f32)
// ... all a.L layers in one launch (kl_scan_w128_multi_fits), a layer polling the dZ rows the one above publishes: the caller
// pre-fills dZ[1 .. L - 1] with 0xFFFF halfwords; bias gradients to db_l
int kl_launch_scan_bwd_w128_multi(KlScanBwd args, hipStream_t stream);
int kl_scan_bwd_regtile_min_np();
// output projection + softmax + CE + dlogits of a training window in one pass (V = 256, width 512); KL_ERR_SHAPE = not applicable
int kl_launch_logits_ce_ws(const bf16_t* X, const bf16_t* E, const int* tgt, bf16_t* dlogits, float* rowstat, int B, int T, int W,
                           int V, long ld_dl, float inv_count, int last_only, hipStream_t stream);
// dH = dlogits . E (bf16) of a training window, weight-stationary (V padded to 256, width 512); KL_ERR_SHAPE = not applicable
int kl_launch_dh_ws(const bf16_t* dlogits, const bf16_t* ET, bf16_t* dH, long M, int W, int Vp, hipStream_t stream);
// ... at width 128 (V <= Vp <= 256): also dH = dlogits . E as f32 rows, in the same pass (lstm_scan_w128.hip)
int kl_launch_logits_ce_w128(const bf16_t* X, const bf16_t* E, const bf16_t* ET, const int* tgt, bf16_t* dlogits, float* dH, float* rowstat, int B,
                             int T, int W, int V, int Vp, float inv_count, int last_only, hipStream_t stream);
int kl_launch_rowstat_reduce(const float* rowstat, int rows, float* loss_acc, hipStream_t stream);
// weight-stationary P = X . KTp^T + bp for width 512 (lstm_scan2.hip: proj_ws_kernel); KL_ERR_SHAPE = not applicable
int kl_launch_proj_ws(const bf16_t* X, const bf16_t* KTp, const float* bp, bf16_t* P, long M, int W, unsigned* status,
                      hipStream_t stream);  // second generation (lstm_scan2.hip): a.G gate-interleaved, rolling sentinels

// thin split-precision contraction C[M,N] = A[M,K] . WT[N,K]^T (+bias) for
// small M (tables, inference logits)
int kl_launch_thin_gemm(const KlOperand* op, int M, int N, float* C, long ldc, const float* bias, int split,
                        hipStream_t stream);

// ---- elementwise.hip ----------------------------------------------------
int kl_launch_embed_gather(const float* E, const float* const* ctx_tabs, int n_ctx, int ctx_dim, int W,
                           const int* idx, const int* ctx, int B, int T, bf16_t* X, long ldx, int Dp,
                           hipStream_t stream);
int kl_launch_transpose_bf16(const bf16_t* in, long ld_in, bf16_t* out, long ld_out, int rows, int cols,
                             hipStream_t stream);
// Several conversions in ONE launch (the operands a training step re-derives after every Adam update: ten small matrices
// per two-layer model, each a launch of a few microseconds before).  rows_pad >= rows: source rows up to rows_pad count as
// zeros and are written (the embedding's rows beyond the vocabulary).
#define KL_CONV_MAX_JOBS 24
struct KlConvJob {
  const float* in; long ld_in; int rows, cols, rows_pad, transpose;
  bf16_t* out_hi; bf16_t* out_lo; long ld_out;
};
struct KlConvJobs { KlConvJob job[KL_CONV_MAX_JOBS]; };
int kl_launch_f32_to_bf16_jobs(const KlConvJob* jobs, int n, hipStream_t stream);
// out[(v * R2 + c)][j] = bf16(A[v][j] + B[c][j]) for all pairs of rows (N a multiple of 8): the table of every gate-input row of layer 0
int kl_launch_comb_table(const float* A, const float* B, int R1, int R2, int N, bf16_t* out, hipStream_t stream);
// rows_tm[t * Bn + b] = idx[b][t] * R2 + ctx[b][t][0]: the rows of that table a window asks for, time-major
int kl_launch_rows_tm(const int* idx, const int* ctx, int n_ctx, int Bn, int T, int R2, int* out, hipStream_t stream);
int kl_launch_f32_to_bf16_t(const float* in, long ld_in, int rows, int cols, bf16_t* out_hi, bf16_t* out_lo,
                            long ld_out, int transpose, hipStream_t stream);
int kl_launch_softmax_ce(float* logits, long ld, int rows, int V, const int* tgt, int B, int T, float inv_count,
                         bf16_t* dlogits, long ld_dl, float* loss_acc, float* rowstat, int time_major,
                         hipStream_t stream, int last_only = 0);
int kl_launch_adam(float* p, const float* g, float* m, float* v, size_t n, float lr_t, float b1, float b2,
                   float eps, float clip, float grad_scale, hipStream_t stream);
int kl_launch_onehot_t(const int* ids, int B, int T, int n_classes, int col, int n_cols, bf16_t* out, long ld,
                       hipStream_t stream);
int kl_launch_onehot_dense(const int* ids, int B, int T, int n_rows, int col, int n_cols, bf16_t* out, long ld,
                           hipStream_t stream);      // zeros included: no fill in front
int kl_launch_regulariser_grads(const float* E, int V, int W, const float* const* ctx_tabs, int n_ctx, int ctx_vocab,
                                int ctx_dim, float* gE, float* const* gCtx, float* loss_acc, float* scratch,
                                hipStream_t stream);
int kl_launch_state_to_rows(const float* states, int B, int W, int L, int layer, bf16_t* h_bf16, float* h_f32,
                            float* c_f32, hipStream_t stream);
int kl_launch_fill_f32(float* p, size_t n, float v, hipStream_t stream);
int kl_launch_fill_bf16(bf16_t* p, size_t n, unsigned short bits, hipStream_t stream);
int kl_zero_async(void* p, size_t bytes, hipStream_t stream);               // kernel-based memset(0)
int kl_fill_u32_async(void* p, size_t bytes, unsigned value, hipStream_t stream);
int kl_zero_coherent_async(unsigned* p, size_t n_words, hipStream_t stream);  // write-through zero of polled words

// ---- tables.hip ---------------------------------------------------------
int kl_launch_small_table(const float* A, int R, int D, const float* Kmat, long ldk, int N, float* C, long ldc,
                          hipStream_t stream);
int kl_launch_p1_gather(const float* EK, const float* const* ctxk, int n_ctx, const float* bias, const int* idx,
                        const int* ctx, int B, int T, int N, float* P, hipStream_t stream);
int kl_launch_colsum_bf16(const bf16_t* in, long ld, int rows, int cols, float* out, hipStream_t stream);
int kl_launch_rows_to_state(const void* h_rows, int h_is_f32, const float* c_rows, int B, int W, int L, int layer,
                            float* states, hipStream_t stream);
int kl_launch_ctx_grads(const float* Ctx, const float* K0rows, long ldk, int R, int D, const float* dT, long ldt, int N,
                        float* gK, long ldg, float* gCtx, hipStream_t stream);
int kl_launch_rows_tm_to_bm(const float* in, long ld_in, float* out, int B, int T, int V, hipStream_t stream);

// ---- step_small.hip -----------------------------------------------------
// one LSTM cell step of a layer for n hypotheses with pool slots (KL_SMALL_STEP_N <= n < KL_BIG_STEP_N), see inc_cell_kernel
#define KL_SMALL_STEP_N 16
struct KlIncCellArgs {
  int n, W, split;                     // split: 1 = bf16, 3 = split-bf16
  float* pool; long slot_ld;           // [slots][2L][W] f32
  const int* slot_in; const int* slot_out;
  int h_off, c_off, x_off;             // float offsets inside a slot: this layer's h and c, the layer below's (new) h (-1: layer 0)
  const bf16_t* UF; const bf16_t* KF;  // U^T and K^T of the layer FRAGMENT-MAJOR (kl_launch_frag_major of the [4W][W] hi / lo arrays;
                                       // hi and lo planes interleaved per block when split == 3); KF null for layer 0
  const float* T1; const int* i1; const float* T2; const int* i2; const float* bias;    // z init: T1[i1[r]] + T2[i2[r]] + bias (null: none)
};
// kl_step_batch_host: the per-hypothesis indices of a step as part of the KERNEL ARGUMENTS (up to KL_HOST_STEP_MAX rows; 3 KiB of the
// 4 KiB a dispatch may carry): pool slots as they are, characters and the one context variable as halfwords
#define KL_HOST_STEP_MAX 256
struct KlHostIdx {
  int slot_in[KL_HOST_STEP_MAX];
  int slot_out[KL_HOST_STEP_MAX];
  unsigned short idx[KL_HOST_STEP_MAX];
  unsigned short ctx[KL_HOST_STEP_MAX];
};
struct KlHostTargets { unsigned short t[KL_HOST_STEP_MAX]; };
// what the last launch of a host-driven step delivers into (device-visible) host memory, see step_finish_kernel
struct KlStepFinish {
  int n, V, W;
  const float* logits; long ld;      // device [n][V]: logits (softmax != 0) or probabilities
  int softmax, by_target, head_k;
  const int* target;                 // device, or null: KlHostTargets in the kernel arguments (n <= KL_HOST_STEP_MAX)
  const float* pool; long slot_ld; const int* slot_out;      // device; heads = the first head_k * W floats of the new slots
  float* probs_host; float* heads_host; unsigned* done_host; unsigned ticket;
  unsigned* counter;                 // device, zero between steps
};
int kl_launch_step_finish(const KlStepFinish& a, const KlHostTargets* tx, hipStream_t stream);
// hx != null: the indices come from *hx (slot_in / slot_out / the VALUES behind i1 / i2 of `a` are ignored; i1 / i2 non-null
// still mean "table rows by index"); slots_copy (device, n ints) receives slot_out for the launches that follow
int kl_launch_inc_cell(const KlIncCellArgs& a, hipStream_t stream, const KlHostIdx* hx = nullptr, int* slots_copy = nullptr);      // KL_ERR_SHAPE: not applicable

// ---- step_tile.hip: the same for n >= KL_BIG_STEP_N, TR x 128 tiles with the operands read once (variant: timing builds, 0)
int kl_launch_inc_tile(const KlIncCellArgs& a, int variant, hipStream_t stream, int rows = -1);      // KL_ERR_SHAPE: not applicable; rows: 64 / 128 per tile, -1 = by size
// output layer in one launch: probs[n][V] = softmax(h_top . E^T), h_top rows through slot_out (V <= 256, W % 128 == 0)
int kl_launch_out_softmax(const float* pool, long slot_ld, const int* slot_out, int h_off, const bf16_t* EF, int split,
                          int n, int W, int V, float* probs, long ldp, hipStream_t stream);      // KL_ERR_SHAPE: not applicable
// [rows][K] bf16 hi (+ lo) -> fragment-major [rows / 16][K / 32][planes][64][8] (the operands of the two launchers above)
int kl_launch_frag_major(const bf16_t* hi, const bf16_t* lo, int rows, int K, long ld, bf16_t* out, hipStream_t stream);

// ---- step_big.hip -------------------------------------------------------
#define KL_BIG_STEP_N 256   // from this many hypotheses on, kl_step_batch uses big-tile GEMMs
int kl_launch_split_gather(const float* s0, long ld0, const int* i0, int w0, const float* s1, long ld1, const int* i1,
                           int w1, int n, int nb, bf16_t* out, long ld_out, hipStream_t stream);
int kl_launch_gates_rows(const float* z, long ldz, int n, int W, const float* T1, const int* i1, const float* T2,
                         const int* i2, const float* bias, const float* c_prev, long c_ld, const int* slot_in,
                         float* c_out, float* h_out, long out_ld, const int* slot_out, hipStream_t stream);
