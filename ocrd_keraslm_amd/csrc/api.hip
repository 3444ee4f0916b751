// C-ABI orchestration of the Rater hot path on gfx950 (see include/keraslm_hip.h).
//
// The library owns no device memory: parameters, gradients, optimizer moments,
// state rows and workspaces all belong to the caller.  This file lays out the
// workspaces, derives the bf16 operand copies, and issues the launch sequences:
//
//   forward   P1 = table gather (layer-0 input contraction as look-ups) ->
//             layer wavefront of thin fused cell steps (one launch per diagonal:
//             layer l runs time d-l) -> tied-embedding logits -> softmax/CE
//   backward  dH = dlogits.E -> reverse layer wavefront of fused backward cell
//             steps -> weight gradients as big K=B*T GEMMs over transposed
//             activations -> layer-0/embedding gradients through one-hot^T
//             segment sums -> regularisers
//   step      L launches + logits + softmax for n hypotheses with pool slots
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <functional>
#include <utility>

#include <vector>

#include "kl_common.h"
#include "kl_kernels.h"

namespace {

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int round_up_i(int x, int a) { return (x + a - 1) / a * a; }

// bump allocator; with base == nullptr it only measures
struct Carver {
  unsigned char* base;
  size_t off = 0;
  explicit Carver(void* b) : base(reinterpret_cast<unsigned char*>(b)) {}
  template <typename T>
  T* take(size_t count) {
    off = align_up(off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

constexpr int KL_SCAN_FLAGS = 256 * 64;
struct Derived {
  std::vector<bf16_t*> UT_hi, UT_lo, KT_hi, KT_lo, Un, Kn;   // per layer (KT/Kn of layer 0 = rows [0,W) of K0)
  bf16_t *E_hi = nullptr, *E_lo = nullptr, *ET = nullptr;
  std::vector<bf16_t*> WTcat;    // [4W][3*K_l] = [hi | hi | lo] blocks, K_0 = W (U), K_l = 2W (K then U): big-n step path
  std::vector<bf16_t*> UF, KF;   // per layer: U^T / K^T fragment-major (hi and lo planes per 16 x 32 block), step_tile.hip / step_small.hip (W % 32 == 0)
  bf16_t* EF = nullptr;          // the embedding likewise
  std::vector<bf16_t*> WTperm;   // WTcat with rows in (unit block of 32, gate, unit) order: fused cell epilogue (W % 32 == 0)
  bf16_t* Ecat = nullptr;        // [Vp][3W]
  float* EK = nullptr;
  std::vector<float*> CtxK;
  // gate-interleaved copies for the second-generation wide scans (lstm_scan2.hip): column / row u*4+g <- g*W+u
  std::vector<bf16_t*> KTp;      // [4W][W], l >= 1
  std::vector<float*> bp;        // [4W], l >= 1 (layer 0's bias is folded into EKp)
  float* EKp = nullptr;          // [V][4W] = EK + b_0
  bf16_t* comb = nullptr;        // width 512, one context variable: [V * ctx_vocab][W][4] bf16 = EKp[v] + CtxKp_0[c], every gate-input row layer 0 can
                                 // ask for (200 MiB at V = 256) -- the eight-wave forward scan gathers its rows from it (lstm_scan_fwd8.hip, table mode)
  std::vector<float*> CtxKp;     // [ctx_vocab][W][4] per context variable (the scan's table mode reads that of variable 0)
  unsigned* scan_flags = nullptr;   // lstm_scan_bwd_wide2_kernel's flags + epoch (zeroed once per bind: the numbers only grow)
};

struct WindowWs {
  float* P1;
  std::vector<void*> H;        // [(T+1)B][W]  f32 (inference) or bf16 (training)
  std::vector<float*> C;       // [(T+1)B][W]
  float* logits;               // [BT][V]
  float* rowstat;              // [BT][2] per-row (loss, hit)
  int *s_idx, *s_ctx, *s_tgt;  // staged inputs (fixed addresses for graph replay)
  int* ids_tm = nullptr;       // training: [T+1][B][2] table-row byte offsets, time-major (second-generation wide scans)
  int scan2_rows = 0;          // second-generation wide scans planned for this window: rows per forward phase (0: first generation)
  bool scan2_bwd = false;      // ... and the backward scan reads gate-interleaved G
  float *s_masks, *s_probs;
  unsigned *scan_cnt, *scan_status;   // persistent-scan hand-off counters [L][ceil(B/16)][T], status words [2]
  float* reg_scratch;                 // statistics of the embedding regularisers
  // training only
  std::vector<bf16_t*> G, dZ, Hd;
  std::vector<bf16_t*> Cb;            // second-generation wide scans: cell states for the backward scan as bf16 [(T+1)B][W]
  std::vector<bf16_t*> Xhi, Xlo;      // inference: state planes exchanged by the split-precision scan [(T+1)B][W]
  std::vector<bf16_t*> HTf, HdT;      // transposed outputs written by the wide forward scans: [W][(T+1)B], [W][BT]
  bool ht_ready = false;              // ... valid for this window
  bool km_plan = false;               // the weight-gradient GEMMs will read dZ and the activations K-major (row-major as written): no transposed copies
  bf16_t *dZT, *HT, *dlogits, *dlogitsT, *OHT, *dEKT_bf, *dEK_bf;
  std::vector<bf16_t*> OHC;
  float *dH, *dEKT;
  std::vector<float*> dc0, dc1, dCtxKT;
};

}  // namespace

struct kl_handle {
  kl_config cfg;
  int Vp;                       // voc_size rounded up to 32
  size_t n_params;
  std::vector<size_t> off_K, off_U, off_b, off_Ctx;
  size_t off_E;
  float* params = nullptr;
  void* derived_ws = nullptr;
  size_t derived_bytes = 0;
  Derived d;
  int precision = 0;            // 0 = not prepared
  // hipGraph cache of whole-window launch sequences (keyed by every baked-in pointer)
  struct GraphKey {
    int kind, B, T, flags, precision;
    const void *states, *loss_acc, *ws, *grads;
    int layout = 0;               // workspace layout the body was captured with (kl_forward_window picks it from ws_bytes)
    bool operator==(const GraphKey& o) const {
      return kind == o.kind && B == o.B && T == o.T && flags == o.flags && precision == o.precision &&
             states == o.states && loss_acc == o.loss_acc && ws == o.ws && grads == o.grads && layout == o.layout;
    }
  };
  std::vector<std::pair<GraphKey, hipGraphExec_t>> graphs;
  bool graphs_enabled = true;
  bool scan_enabled = true;     // persistent scans (KL_SCAN=0 forces the launch-per-step path)
  bool seq_bwd = true;          // layer-sequential backward scans for many row blocks (KL_SEQ_BWD=0: always fused)
  bool wide_bwd = true;         // ... with 64-unit workgroups (KL_WIDE_BWD=0: thin workgroups)
  void* host_step_ready = nullptr;      // kl_step_batch_host: the workspace whose ticket counter has been zeroed
  std::vector<int32_t> host_pack;       // ... its packed index block for the copy to the device (n > 256, fall-backs)
  bool host_kernarg = true;             // ... indices in the kernel arguments (KL_HOST_KERNARG=0: always the copy)
  bool inc_ready = false;       // the incremental step's fragment-major operands match the current weights (prepare_incremental)
  bool big_ready = false;       // ... and those of the gather + GEMM path (prepare_big_step)
  int last_only = 0;            // stateless windows: one target per row, at the last position (kl_set_window_mode)
  int loss_rows = 0;            // rows the training means are taken over when the batch carries dummy streams (kl_set_loss_rows; 0: B)
  bool sentinel = true;         // wide scans hand off by data sentinels instead of counters (KL_SENTINEL=0: counters)
  bool gemm_an = true;          // weight gradients read the backward scan's dZ K-major, no transposed copy (KL_GEMM_AN=0: dZ^T)
  bool xcd_local = false;       // KL_XCD_LOCAL=1: sentinel hand-off inside one XCD through its L2 (plain stores) where the placement allows
  bool sentinel_bwd = true;
  bool sentinel_roll = true;    // KL_SENTINEL_ROLL=0: pre-fill all of dZ instead of re-arming two steps ahead inside the scan
  bool sentinel_bwd_all = false; // KL_SENTINEL_BWD=2: also with one row block per workgroup
  bool xcd_local_bwd = false;    // KL_XCD_LOCAL_BWD=1     // the same for the wide backward scan (KL_SENTINEL_BWD=0, or KL_SENTINEL=0: counters)
  bool w32 = true;              // width 1024: the eight-wave scans of lstm_scan_w32.hip (KL_W32=0: the thin scans)
  bool w32_local = false;       // KL_W32_LOCAL=1: ... handing over through the XCD's own L2 where the placement allows (measured slower: 112 vs 104 ms per cfg5 step)
  int w32_min_rb = 8;           // ... from this many row blocks of 16 streams (KL_W32_MIN_RB)
  bool split8 = true;           // width-1024 rating windows with up to 32 streams: two layers per launch, eight units per workgroup (KL_SPLIT8=0: layer by layer)
  bool split_sentinel = true;   // rating windows: the split-precision scan hands over by data sentinels (KL_SPLIT_SENTINEL=0: counters)
  bool inc_small = true;        // incremental step: step_small.hip's kernels (KL_INC_SMALL=0: the launch-per-layer thin kernels + thin GEMM + softmax)
  bool fused_step = true;       // incremental step, n >= 256: cell fused into the GEMM epilogue (KL_FUSED_STEP=0: separate kernels)
  bool inc_tile = true;         // incremental step, n >= 256: step_tile.hip's one launch per layer (KL_INC_TILE=0: gather + [hi|lo|hi] GEMM)
  int tile_var = 0;             // KL_TILE_VAR: timing variants of inc_tile_kernel (never in production)
  int tile_rows = -1;           // KL_TILE_ROWS=64|128: rows per tile of inc_tile_kernel (default: by size)
  bool out_fused = true;        // incremental step: logits + softmax in one launch (KL_OUT_FUSED=0: thin GEMM + softmax kernel)
  int out_fused_min = 512;      // ... from this many hypotheses on (KL_OUT_FUSED_MIN; width 512: 128 rows 25.7 us per step against 24.2, 256 rows 35.8 / 35.1, 1024 rows 41.3 / 43.5)
  int inc_small_min = KL_SMALL_STEP_N;      // step_small.hip's kernel from this many hypotheses on (KL_INC_SMALL_MIN)
  bool scan2 = true;            // second-generation wide scans where their grid plan applies (KL_SCAN2=0: first generation)
  int scan2_rows = 0;           // KL_SCAN2_ROWS = 16 / 32: rows per forward phase (0: chosen by shape)
  int scan2_pf = -1;            // KL_SCAN2_PF: where the forward scan requests its next tile (0: top of a phase, 1: behind the MFMA phase, 2: two phases ahead; -1: by shape)
  bool scan2_bf16 = true;       // KL_SCAN2_BF16=0: f32 instead of bf16 for what the scans exchange with later kernels (P, dH, c for backward)
  bool logits_ws = true;        // KL_LOGITS_WS = 0: GEMM + softmax kernel for the training window's output layer instead of the fused kernel
  bool proj_ws = true;          // KL_PROJ_WS = 0: the ring GEMM for the gate inputs P of the second-generation scans too
  bool w128 = true;             // KL_W128 = 0: the thin fused scans also at width 128 (lstm_scan_w128.hip: a workgroup per 16-row block, all units)
  bool w128_tables = true;      // KL_W128_TABLES = 0: layer 0's gate inputs gathered into f32 rows first instead of inside the scan
  bool w128_multi = true;       // KL_W128_MULTI = 0: one launch per layer also where all layers' workgroups fit the CUs at once
  bool w128_fuse = true;        // KL_W128_FUSE = 0: ... with the input side of the layers above the first from products over all steps
  int w128_min = 1;             // ... from this many streams on (KL_W128_MIN; measured against the thin fused scans, ms per step at 1 / 16 / 64 / 128 / 256 streams:
                                // 1.54 / 1.68 / 1.69 / 1.74 / 1.87 against 1.57 / 1.80 / 1.79 / 1.96 / 2.29)
  bool fwd8 = true;             // KL_FWD8 = 0: the 16-wave forward scan also for the layers above the first (default: the eight-wave scan of
                                // lstm_scan_fwd8.hip there -- 3.06 against 3.36 ms per launch at 3072 streams)
  bool fwd8_all = false;        // KL_FWD8 = 2: ... also for layer 0 (its gate inputs gathered into P rows first)
  bool fwd8_tab = true;         // KL_FWD8_TAB = 0: layer 0 (one context variable) stays on the 16-wave scan's table mode (default: the eight-wave scan,
                                // its gate-input rows gathered from the table of all (character, context value) sums -- 200 MiB more of derived operands;
                                // 23.03 / 23.20 against 23.21 / 23.28 ms per step with 200 context values in the batch, where its gathers leave the L2)
  bool fwd8_ls = true;          // KL_FWD8_LS = 0: the counter form of the eight-wave forward scan instead of the two-barrier form
  bool fwd8_local = true;       // KL_FWD8_LOCAL = 0: write-through publishes in the eight-wave forward scan even where its partners share an XCD
  int fwd8_pf = -1;             // KL_FWD8_PF = 0..3: where it requests its tiles (default by phases per step)
  bool rt_local = true;         // KL_RT_LOCAL = 0: write-through publishes in the register-tile backward scan even where its partners share an XCD
  bool regtile = true;          // KL_REGTILE = 0: the backward scan's tiles by LDS-DMA at every size (else through registers from five blocks per step)
  bool scan2_flags = true;      // KL_SCAN2_FLAGS = 0: the backward scan hands over by data sentinels at every size (else by flags from three blocks per step)
  bool flags_zeroed = false;
  bool fuse_wg = true;          // KL_FUSE_WG = 0: one launch per weight-gradient product (else products over the same dZ share a pass)
  int scan2_pfb = -1;           // KL_SCAN2_PFB: the same for the backward scan (-1: by shape)
  int wide_fwd_min = 96;        // layer-sequential wide forward scans from this many 64-unit workgroups (KL_WIDE_FWD_MIN; 0 = never)
  double trace_flops[2] = {0.0, 0.0};   // algorithmic FLOPs of ONE timed launch
  // optional per-launch timing of the cell-step kernels with HIP events (bench.py's
  // roofline leg).  While tracing, windows run eagerly (events are not captured).
  bool trace_on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> trace_ev[2];   // [0] forward steps, [1] backward steps
  size_t trace_used[2] = {0, 0};
  bool trace_open[2] = {false, false};
  bool trace_persistent[2] = {false, false};   // the timed launches were whole-window persistent scans
  const char* trace_name[2] = {"lstm_fwd_step_kernel", "lstm_bwd_step_kernel"};   // kernel the timed launches ran
  void trace_begin(int kind, hipStream_t s) {
    if (!trace_on) return;
    if (trace_used[kind] == trace_ev[kind].size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { trace_on = false; return; }
      trace_ev[kind].emplace_back(a, b);
    }
    (void)hipEventRecord(trace_ev[kind][trace_used[kind]].first, s);
    trace_open[kind] = true;
  }
  void trace_end(int kind, hipStream_t s) {
    if (!trace_on || !trace_open[kind]) return;
    trace_open[kind] = false;
    (void)hipEventRecord(trace_ev[kind][trace_used[kind]].second, s);
    ++trace_used[kind];
  }
  void drop_graphs() {
    for (auto& g : graphs) (void)hipGraphExecDestroy(g.second);
    graphs.clear();
  }
};

namespace {

void layout(kl_handle* h) {
  const kl_config& c = h->cfg;
  const size_t W = c.width;
  size_t off = 0;
  h->off_E = off;
  off += (size_t)c.voc_size * W;
  h->off_Ctx.clear();
  for (int n = 0; n < c.n_ctx; ++n) {
    h->off_Ctx.push_back(off);
    off += (size_t)c.ctx_vocab * c.ctx_dim;
  }
  h->off_K.clear();
  h->off_U.clear();
  h->off_b.clear();
  for (int l = 0; l < c.depth; ++l) {
    const size_t D = l == 0 ? W + (size_t)c.ctx_dim * c.n_ctx : W;
    h->off_K.push_back(off);
    off += D * 4 * W;
    h->off_U.push_back(off);
    off += W * 4 * W;
    h->off_b.push_back(off);
    off += 4 * W;
  }
  h->n_params = off;
  h->Vp = round_up_i(c.voc_size, 32);
}

bool config_ok(const kl_config* c) {
  if (!c) return false;
  if (c->depth < 1 || c->depth > 16) return false;
  if (c->width < 32 || (c->width & 31)) return false;
  if (c->voc_size < 1) return false;
  if (c->n_ctx < 0 || c->n_ctx > 8) return false;
  if (c->ctx_vocab < 2 || c->ctx_dim < 1) return false;
  return true;
}

size_t carve_derived(const kl_handle* h, void* base, Derived* d) {
  const kl_config& c = h->cfg;
  const size_t W = c.width, V = c.voc_size, Vp = h->Vp;
  Carver cv(base);
  Derived tmp;
  Derived& o = d ? *d : tmp;
  o.UT_hi.assign(c.depth, nullptr); o.UT_lo.assign(c.depth, nullptr);
  o.KT_hi.assign(c.depth, nullptr); o.KT_lo.assign(c.depth, nullptr);
  o.Un.assign(c.depth, nullptr); o.Kn.assign(c.depth, nullptr);
  for (int l = 0; l < c.depth; ++l) {
    o.UT_hi[l] = cv.take<bf16_t>(4 * W * W);
    o.UT_lo[l] = cv.take<bf16_t>(4 * W * W);
    o.KT_hi[l] = cv.take<bf16_t>(4 * W * W);
    o.KT_lo[l] = cv.take<bf16_t>(4 * W * W);
    o.Un[l] = cv.take<bf16_t>(4 * W * W);
    o.Kn[l] = cv.take<bf16_t>(4 * W * W);
  }
  o.E_hi = cv.take<bf16_t>(Vp * W);
  o.E_lo = cv.take<bf16_t>(Vp * W);
  o.ET = cv.take<bf16_t>(W * Vp);
  o.EK = cv.take<float>(V * 4 * W);
  o.WTcat.assign(c.depth, nullptr);
  for (int l = 0; l < c.depth; ++l) o.WTcat[l] = cv.take<bf16_t>(4 * W * 3 * (l == 0 ? W : 2 * W));
  o.WTperm.assign(c.depth, nullptr);
  if ((W & 31) == 0)
    for (int l = 0; l < c.depth; ++l) o.WTperm[l] = cv.take<bf16_t>(4 * W * 3 * (l == 0 ? W : 2 * W));
  o.Ecat = cv.take<bf16_t>(Vp * 3 * W);
  o.UF.assign(c.depth, nullptr);
  o.KF.assign(c.depth, nullptr);
  if ((W & 31) == 0) {
    for (int l = 0; l < c.depth; ++l) {
      o.UF[l] = cv.take<bf16_t>(4 * W * W * 2);
      if (l > 0) o.KF[l] = cv.take<bf16_t>(4 * W * W * 2);
    }
    o.EF = cv.take<bf16_t>(Vp * W * 2);
  }
  o.CtxK.assign(c.n_ctx, nullptr);
  for (int n = 0; n < c.n_ctx; ++n) o.CtxK[n] = cv.take<float>((size_t)c.ctx_vocab * 4 * W);
  o.KTp.assign(c.depth, nullptr);
  o.bp.assign(c.depth, nullptr);
  for (int l = 1; l < c.depth; ++l) {
    o.KTp[l] = cv.take<bf16_t>(4 * W * W);
    o.bp[l] = cv.take<float>(4 * W);
  }
  o.EKp = cv.take<float>(V * 4 * W);
  o.CtxKp.assign(c.n_ctx, nullptr);
  for (int n = 0; n < c.n_ctx; ++n) o.CtxKp[n] = cv.take<float>((size_t)c.ctx_vocab * 4 * W);
  o.scan_flags = cv.take<unsigned>(KL_SCAN_FLAGS + 64);      // hand-off flags of the backward scan [256 row blocks][64] + the epoch word
  o.comb = nullptr;
  if (h->fwd8_tab && W == 512 && c.n_ctx == 1 && V * (size_t)c.ctx_vocab * 4 * W * sizeof(bf16_t) <= ((size_t)512 << 20))
    o.comb = cv.take<bf16_t>(V * (size_t)c.ctx_vocab * 4 * W);
  return align_up(cv.off, 256);
}

size_t carve_window(const kl_handle* h, void* base, int B, int T, int training, WindowWs* out) {
  const kl_config& c = h->cfg;
  const size_t W = c.width, V = c.voc_size, Vp = h->Vp, L = c.depth;
  const size_t BT = (size_t)B * T, BTp = round_up_i((int)BT, 8);
  Carver cv(base);
  WindowWs tmp;
  WindowWs& o = out ? *out : tmp;
  o.P1 = cv.take<float>(BT * 4 * W);
  o.H.assign(L, nullptr);
  o.C.assign(L, nullptr);
  for (size_t l = 0; l < L; ++l) {
    o.H[l] = training ? (void*)cv.take<bf16_t>((BT + B) * W) : (void*)cv.take<float>((BT + B) * W);
    o.C[l] = cv.take<float>((BT + B) * W);
  }
  o.Xhi.assign(L, nullptr); o.Xlo.assign(L, nullptr);
  if (!training) {
    for (size_t l = 0; l < L; ++l) {
      o.Xhi[l] = cv.take<bf16_t>((BT + B) * W);
      o.Xlo[l] = cv.take<bf16_t>((BT + B) * W);
    }
  }
  o.logits = cv.take<float>(BT * V);
  o.rowstat = cv.take<float>(BT * 2);
  o.s_idx = cv.take<int>(BT);
  o.s_ctx = cv.take<int>(BT * (size_t)(c.n_ctx > 0 ? c.n_ctx : 1));
  o.s_tgt = cv.take<int>(BT);
  o.s_masks = cv.take<float>(training ? L * (size_t)B * W : 1);
  o.s_probs = cv.take<float>(BT * V);      // (training layout too: validation windows run on it, kl_forward_window)
  o.scan_cnt = cv.take<unsigned>(L * ((size_t)(B + 15) / 16) * T);
  o.scan_status = cv.take<unsigned>(4 + 256);   // status words [4] + XCC posts of the wide scans' workgroups [256]
  o.reg_scratch = cv.take<float>(3 * (W > (size_t)c.ctx_dim ? W : (size_t)c.ctx_dim) + (V > (size_t)c.ctx_vocab ? V : (size_t)c.ctx_vocab) + 8);
  if (training) {
    o.G.assign(L, nullptr); o.dZ.assign(L, nullptr); o.Hd.assign(L, nullptr); o.Cb.assign(L, nullptr);
    o.dc0.assign(L, nullptr); o.dc1.assign(L, nullptr);
    for (size_t l = 0; l < L; ++l) {
      o.G[l] = cv.take<bf16_t>(BT * 4 * W);
      o.dZ[l] = cv.take<bf16_t>(BT * 4 * W);
      o.Hd[l] = l > 0 ? cv.take<bf16_t>(BT * W) : nullptr;
      o.Cb[l] = cv.take<bf16_t>((BT + B) * W);
      o.dc0[l] = cv.take<float>((size_t)B * W);
      o.dc1[l] = cv.take<float>((size_t)B * W);
    }
    o.dZT = cv.take<bf16_t>(4 * W * BTp);
    o.HT = cv.take<bf16_t>(W * BTp);
    o.ids_tm = cv.take<int>((BT + B) * 2);
    o.HTf.assign(L, nullptr); o.HdT.assign(L, nullptr);
    if ((B & 7) == 0) {
      for (size_t l = 0; l < L; ++l) {
        o.HTf[l] = cv.take<bf16_t>(W * (BT + B));
        o.HdT[l] = l > 0 ? cv.take<bf16_t>(W * BT) : nullptr;
      }
    }
    o.dlogits = cv.take<bf16_t>(BT * Vp);
    o.dlogitsT = cv.take<bf16_t>(Vp * BTp);
    o.OHT = cv.take<bf16_t>(Vp * BTp);
    o.OHC.assign(c.n_ctx, nullptr);
    o.dCtxKT.assign(c.n_ctx, nullptr);
    for (int n = 0; n < c.n_ctx; ++n) {
      o.OHC[n] = cv.take<bf16_t>((size_t)c.ctx_vocab * BTp);
      o.dCtxKT[n] = cv.take<float>(4 * W * (size_t)c.ctx_vocab);
    }
    o.dH = cv.take<float>(BT * W);
    o.dEKT = cv.take<float>(4 * W * Vp);
    o.dEKT_bf = cv.take<bf16_t>(4 * W * Vp);
    o.dEK_bf = cv.take<bf16_t>(Vp * 4 * W);
  }
  return align_up(cv.off, 256);
}

#define KL_TRY(expr)            \
  do {                          \
    int _e = (expr);            \
    if (_e != 0) return _e;     \
  } while (0)

int hip_ok(hipError_t e) { return e == hipSuccess ? 0 : KL_ERR_LAUNCH; }

int prepare_impl(kl_handle* h, int precision, hipStream_t s) {
  const kl_config& c = h->cfg;
  const int W = c.width, V = c.voc_size, Vp = h->Vp;
  const float* P = h->params;
  Derived& d = h->d;
  const bool split = precision == KL_PREC_SPLIT;
  if (!h->flags_zeroed) {      // once per bind, and not inside a capture (a replay must not turn the epoch back)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
      KL_TRY(kl_zero_async(d.scan_flags, (size_t)(KL_SCAN_FLAGS + 64) * sizeof(unsigned), s));
      h->flags_zeroed = true;
    }
  }
  // every bf16 operand derived from the parameters, in as few launches as the job list takes (a training step comes here after
  // every Adam update): per layer U^T, K^T (hi [+ lo]) and U, K as they are; the embedding E (rows beyond the vocabulary: zeros)
  // and E^T
  const float* E = P + h->off_E;
  {
    KlConvJob jobs[KL_CONV_MAX_JOBS];
    int nj = 0;
    auto add = [&](const float* in, long ld_in, int rows, int cols, int rows_pad, bf16_t* hi, bf16_t* lo, long ld_out, int tr) -> int {
      jobs[nj++] = KlConvJob{in, ld_in, rows, cols, rows_pad, tr, hi, lo, ld_out};
      if (nj < KL_CONV_MAX_JOBS) return 0;
      nj = 0;
      return kl_launch_f32_to_bf16_jobs(jobs, KL_CONV_MAX_JOBS, s);
    };
    for (int l = 0; l < c.depth; ++l) {
      const float* K = P + h->off_K[l];   // layer 0: rows [0,W) are the char-embedding part
      const float* U = P + h->off_U[l];
      KL_TRY(add(U, 4 * W, W, 4 * W, W, d.UT_hi[l], split ? d.UT_lo[l] : nullptr, W, 1));
      KL_TRY(add(K, 4 * W, W, 4 * W, W, d.KT_hi[l], split ? d.KT_lo[l] : nullptr, W, 1));
      KL_TRY(add(U, 4 * W, W, 4 * W, W, d.Un[l], nullptr, 4 * W, 0));
      KL_TRY(add(K, 4 * W, W, 4 * W, W, d.Kn[l], nullptr, 4 * W, 0));
    }
    KL_TRY(add(E, W, V, W, Vp, d.E_hi, split ? d.E_lo : nullptr, W, 0));
    KL_TRY(add(E, W, V, W, Vp, d.ET, nullptr, Vp, 1));
    if (nj > 0) KL_TRY(kl_launch_f32_to_bf16_jobs(jobs, nj, s));
  }
  // layer-0 look-up tables: EK = E . K0[:W] ; CtxK_n = Ctx_n . K0[W+10n ..]
  KlOperand op;
  memset(&op, 0, sizeof(op));
  op.A = E; op.lda = W; op.a_is_f32 = 1;
  op.WT_hi = d.KT_hi[0]; op.WT_lo = split ? d.KT_lo[0] : nullptr; op.ldw = W; op.K = W;
  KL_TRY(kl_launch_thin_gemm(&op, V, 4 * W, d.EK, 4 * W, nullptr, precision, s));
  for (int n = 0; n < c.n_ctx; ++n) {
    const float* Kc = P + h->off_K[0] + (size_t)(W + n * c.ctx_dim) * 4 * W;
    KL_TRY(kl_launch_small_table(P + h->off_Ctx[n], c.ctx_vocab, c.ctx_dim, Kc, 4 * W, 4 * W, d.CtxK[n], 4 * W, s));
  }
  if (precision == KL_PREC_BF16 && h->scan2 && W == 512) {
    // gate-interleaved copies for the second-generation wide scans
    for (int l = 1; l < c.depth; ++l) {
      KL_TRY(kl_launch_permute_gate_rows_bf16(d.KT_hi[l], d.KTp[l], W, W, s));
      KL_TRY(kl_launch_permute_gate_cols_f32(P + h->off_b[l], nullptr, d.bp[l], 1, W, s));
    }
    KL_TRY(kl_launch_permute_gate_cols_f32(d.EK, P + h->off_b[0], d.EKp, V, W, s));
    for (int n = 0; n < c.n_ctx; ++n) KL_TRY(kl_launch_permute_gate_cols_f32(d.CtxK[n], nullptr, d.CtxKp[n], c.ctx_vocab, W, s));
  }
  h->precision = precision;
  h->inc_ready = false;      // the incremental step's own operands are rebuilt on their first use (prepare_incremental,
  h->big_ready = false;      // prepare_big_step)
  return 0;
}

// Operands only the incremental step reads.  Built lazily: a training step re-derives the window operands after every
// Adam update and never touches these.
// (1) fragment-major copies of the [4W][W] / [Vp][W] hi / lo arrays kl_prepare keeps (step_small.hip, step_tile.hip)
int prepare_incremental(kl_handle* h, hipStream_t s) {
  const kl_config& c = h->cfg;
  const int W = c.width, Vp = h->Vp;
  Derived& d = h->d;
  const bool split = h->precision == KL_PREC_SPLIT;
  if (d.EF) {
    for (int l = 0; l < c.depth; ++l) {
      KL_TRY(kl_launch_frag_major(d.UT_hi[l], split ? d.UT_lo[l] : nullptr, 4 * W, W, W, d.UF[l], s));
      if (l > 0) KL_TRY(kl_launch_frag_major(d.KT_hi[l], split ? d.KT_lo[l] : nullptr, 4 * W, W, W, d.KF[l], s));
    }
    KL_TRY(kl_launch_frag_major(d.E_hi, split ? d.E_lo : nullptr, Vp, W, W, d.EF, s));
  }
  h->inc_ready = true;
  return 0;
}

// (2) the gather + GEMM path (widths beyond 1024 that are not multiples of 256; vocabularies from 1024 characters on):
// concatenated [hi | hi | lo] weights, their gate-permuted copies and the concatenated embedding
int prepare_big_step(kl_handle* h, hipStream_t s) {
  const kl_config& c = h->cfg;
  const int W = c.width, V = c.voc_size, Vp = h->Vp;
  const float* P = h->params;
  Derived& d = h->d;
  const bool split = h->precision == KL_PREC_SPLIT;
  const float* E = P + h->off_E;
  for (int l = 0; l < c.depth; ++l) {
    const int Kl = l == 0 ? W : 2 * W;
    const long ld = 3L * Kl;
    const float* K = P + h->off_K[l];
    const float* U = P + h->off_U[l];
    bf16_t* base = d.WTcat[l];
    const int uoff = l == 0 ? 0 : W;     // U part follows the K part inside each block
    if (l > 0) {
      KL_TRY(kl_launch_f32_to_bf16_t(K, 4 * W, W, 4 * W, base, split ? base + 2 * Kl : nullptr, ld, 1, s));
      if (split) KL_TRY(kl_launch_f32_to_bf16_t(K, 4 * W, W, 4 * W, base + Kl, nullptr, ld, 1, s));
    }
    KL_TRY(kl_launch_f32_to_bf16_t(U, 4 * W, W, 4 * W, base + uoff, split ? base + 2 * Kl + uoff : nullptr, ld, 1, s));
    if (split) KL_TRY(kl_launch_f32_to_bf16_t(U, 4 * W, W, 4 * W, base + Kl + uoff, nullptr, ld, 1, s));
    if (d.WTperm[l]) KL_TRY(kl_launch_permute_gate_rows(base, d.WTperm[l], W, ld, s));
  }
  KL_TRY(kl_zero_async(d.Ecat, (size_t)Vp * 3 * W * sizeof(bf16_t), s));
  KL_TRY(kl_launch_f32_to_bf16_t(E, W, V, W, d.Ecat, split ? d.Ecat + 2 * W : nullptr, 3 * W, 0, s));
  if (split) KL_TRY(kl_launch_f32_to_bf16_t(E, W, V, W, d.Ecat + W, nullptr, 3 * W, 0, s));
  h->big_ready = true;
  return 0;
}

// Second-generation wide scans (lstm_scan2.hip) for this window?  Returns the rows per forward phase (16 / 32) or 0.
// need_bwd: the window is a training window, i.e. the backward scan must be able to read the gate-interleaved G.
int plan_scan2(const kl_handle* h, int B, int T, bool km_plan, bool need_bwd) {
  const kl_config& c = h->cfg;
  const int W = c.width;
  if (!h->scan2 || !h->scan_enabled || !h->sentinel || !km_plan || T < 3) return 0;
  if (c.n_ctx > 1 && !h->scan2_bf16) return 0;      // (several context variables: layer 0's gate inputs as bf16 P rows, kl_launch_p_gather_il)
  if (W != 512) return 0;      // (one tile row = one 1 KiB DMA piece)
  if (h->wide_fwd_min <= 0 || !kl_scan_fwd_wide_applicable(B, T, W) || ((B + 15) / 16) * (W / 64) < h->wide_fwd_min) return 0;
  if (need_bwd && (!h->wide_bwd || !h->seq_bwd || !h->sentinel_bwd || !h->sentinel_roll || !kl_scan_wide2_phases(B, T, W, 16, 6))) return 0;
  const int p16 = kl_scan_wide2_phases(B, T, W, 16, 5), p32 = kl_scan_wide2_phases(B, T, W, 32, 4);
  if (h->scan2_rows == 16) return p16 ? 16 : 0;
  if (h->scan2_rows == 32) return p32 ? 32 : 0;
  // (by shape, measured at B = 1024 .. 3072: 32-row phases as soon as a workgroup has two of them per step)
  if (p32 >= 2) return 32;
  if (p16) return 16;
  return p32 ? 32 : 0;
}

// forward over one window; fills ws (activations) and states
int forward_impl(kl_handle* h, int B, int T, const int* idx, const int* ctx, float* states, const float* masks,
                 int training, WindowWs& w, hipStream_t s) {
  const kl_config& c = h->cfg;
  const int W = c.width, L = c.depth;
  const size_t BW = (size_t)B * W;
  const float* P = h->params;
  Derived& d = h->d;
  const int split = training ? 1 : h->precision;
  std::vector<const float*> ctxk(c.n_ctx);
  for (int n = 0; n < c.n_ctx; ++n) ctxk[n] = d.CtxK[n];
  for (int l = 0; l < L; ++l)
    KL_TRY(kl_launch_state_to_rows(states, B, W, L, l, training ? (bf16_t*)w.H[l] : nullptr,
                                   training ? nullptr : (float*)w.H[l], w.C[l], s));
  auto hrow = [&](int l, int block) -> const void* {
    return training ? (const void*)((bf16_t*)w.H[l] + (size_t)block * BW) : (const void*)((float*)w.H[l] + (size_t)block * BW);
  };
  bool scanned = false;
  // Many streams (B >= 512 at cfg2): one layer per launch with 64-unit workgroups.  The
  // input contraction of layers >= 1 comes from one big GEMM over all steps, layer 0
  // reads the look-up tables directly, and the scans also leave H^T for the weight gradients.
  const int n_rb = (B + 15) / 16;
  w.ht_ready = false;
  if (training && h->scan_enabled && h->wide_fwd_min > 0 && kl_scan_fwd_wide_applicable(B, T, W) &&
      n_rb * (W / 64) >= h->wide_fwd_min) {
    for (int l = 0; l < L; ++l) {
      if (!w.km_plan) KL_TRY(kl_launch_transpose_bf16((const bf16_t*)w.H[l], W, w.HTf[l], (long)(T + 1) * B, B, W, s));
      const bool masked = masks != nullptr && l > 0;
      KlScanFwdWide a;
      memset(&a, 0, sizeof(a));
      a.B = B; a.T = T; a.W = W;
      a.UT = d.UT_hi[l];
      const bool v2 = w.scan2_rows != 0;
      // the eight-wave scan (lstm_scan_fwd8.hip) takes bf16 P rows only: layer 0's gate inputs are gathered in front of it
      // (layer 0 with one context variable stays on the 16-wave scan's table mode: gathering its P rows first costs more --
      //  1.25 ms at 3072 streams -- than the eight-wave scan saves; KL_FWD8=2 takes it for layer 0 as well)
      // (layer 0 with one context variable: the eight-wave scan gathers its rows from the table of all sums -- d.comb)
      const bool f8_tab = v2 && l == 0 && h->fwd8 && h->fwd8_tab && !h->fwd8_all && d.comb != nullptr && c.n_ctx == 1 && w.scan2_rows == 32 &&
                          h->scan2_bf16 && h->sentinel_roll && T >= 3 && h->fwd8_ls;
      const bool f8 = v2 && h->fwd8 && w.scan2_rows == 32 && h->scan2_bf16 && h->sentinel_roll && T >= 3 &&
                      (l > 0 || c.n_ctx > 1 || h->fwd8_all || f8_tab);
      if (l > 0) {
        const bool masked_in = masks != nullptr && (l - 1) > 0;
        const bf16_t* X = masked_in ? w.Hd[l - 1] : (const bf16_t*)w.H[l - 1] + BW;
        // (second generation: P in bf16, gate-interleaved -- half the bytes written here and read by the scan)
        // (weight-stationary kernel where it applies -- width 512, bf16 P -- else the ring GEMM)
        int pe = KL_ERR_SHAPE;
        if (v2 && h->scan2_bf16 && h->proj_ws)
          pe = kl_launch_proj_ws(X, d.KTp[l], d.bp[l], reinterpret_cast<bf16_t*>(w.P1), (long)B * T, W, w.scan_status, s);
        if (pe == KL_ERR_SHAPE)
          pe = kl_launch_gemm_tn(X, v2 ? d.KTp[l] : d.KT_hi[l], w.P1, v2 ? d.bp[l] : P + h->off_b[l], B * T, 4 * W, W, W, W, 4 * W,
                                 v2 && h->scan2_bf16 ? 1 : 0, 1, 1.f, s);
        KL_TRY(pe);
        a.P = w.P1;
        a.p_bf16 = v2 && h->scan2_bf16 ? 1 : 0;
      } else if (f8_tab) {
        // every gate-input row layer 0 can ask for: 200 MiB, ~0.05 ms -- built by every window that gathers from it (no "already
        // built" flag: a captured window is replayed after later updates of the operands, and must then build it again)
        KL_TRY(kl_launch_comb_table(d.EKp, d.CtxKp[0], c.voc_size, c.ctx_vocab, 4 * W, d.comb, s));
        KL_TRY(kl_launch_rows_tm(idx, ctx, c.n_ctx, B, T, c.ctx_vocab, w.ids_tm, s));
        a.P = reinterpret_cast<const float*>(d.comb);
        a.p_bf16 = 1;
        a.ids_tm = w.ids_tm;      // (here: row numbers of d.comb, [T][B])
        a.V = c.voc_size; a.ctx_vocab = c.ctx_vocab;
      } else if (v2 && (c.n_ctx > 1 || f8)) {
        // several context variables: the scan's table mode adds ONE context row to the character row, so the gate inputs
        // of layer 0 are gathered into P rows first (as the layers above get them from proj_ws_kernel)
        KL_TRY(kl_launch_p_gather_il(d.EKp, d.CtxKp.data(), c.n_ctx, idx, ctx, B, T, W, c.voc_size, c.ctx_vocab,
                                     reinterpret_cast<bf16_t*>(w.P1), s));
        a.P = w.P1;
        a.p_bf16 = 1;
      } else if (v2) {
        KL_TRY(kl_launch_ids_tm(idx, ctx, c.n_ctx, B, T, W, c.voc_size, c.ctx_vocab, w.ids_tm, s));
        a.EK = d.EKp;
        a.CtxK[0] = c.n_ctx > 0 ? d.CtxKp[0] : nullptr;
        a.n_ctx = c.n_ctx;
        a.ids_tm = w.ids_tm;
        a.V = c.voc_size; a.ctx_vocab = c.ctx_vocab;
      } else {
        a.EK = d.EK;
        for (int n = 0; n < c.n_ctx; ++n) a.CtxK[n] = d.CtxK[n];
        a.n_ctx = c.n_ctx;
        a.idx = idx; a.ctx = ctx;
        a.bias = P + h->off_b[0];
      }
      a.H = (bf16_t*)w.H[l]; a.C = w.C[l]; a.G = w.G[l];
      a.Cb = (v2 && w.scan2_bwd && h->scan2_bf16) ? w.Cb[l] : nullptr;      // (the backward scans read the cell states as bf16, blocks 0..T)
      a.Hd = masked ? w.Hd[l] : nullptr;
      a.mask = masked ? masks + (size_t)l * BW : nullptr;
      a.HT = w.km_plan ? nullptr : w.HTf[l]; a.ldt = (long)(T + 1) * B;       // (K-major GEMMs: no transposed outputs)
      a.HdT = (masked && !w.km_plan) ? w.HdT[l] : nullptr; a.ldt_d = (long)B * T;
      a.counters = w.scan_cnt;
      a.status = w.scan_status;
      a.sentinel = h->sentinel ? 1 : 0;
      a.xcc_slots = (a.sentinel && h->xcd_local) ? w.scan_status + 4 : nullptr;
      a.gen = (unsigned)(1 + l);                   // (the posts are zeroed once per window: a token per launch)
      if (a.sentinel && v2 && h->sentinel_roll && T >= 3) {
        // rolling sentinels: the scan arms block t + 3 while it publishes block t + 1; only the first two start armed
        a.sentinel = 2;
        KL_TRY(kl_fill_u32_async((bf16_t*)w.H[l] + BW, (size_t)2 * BW * sizeof(bf16_t), 0xFFFFFFFFu, s));
      } else if (a.sentinel)   // hand-off by data: the blocks the scan is going to publish start out as sentinels
        KL_TRY(kl_fill_u32_async((bf16_t*)w.H[l] + BW, (size_t)T * BW * sizeof(bf16_t), 0xFFFFFFFFu, s));
      else
        KL_TRY(kl_zero_coherent_async(w.scan_cnt, (size_t)n_rb * T, s));
      // (two phases ahead needs the rows to have been published a phase before the request: three or more phases per workgroup)
      a.pf_mode = h->scan2_pf >= 0 ? h->scan2_pf : (v2 && kl_scan_wide2_phases(B, T, W, w.scan2_rows, w.scan2_rows == 16 ? 5 : 4) >= 3 ? 2 : 1);
      h->trace_begin(0, s);      // (every layer's scan launch is timed while tracing)
      bool took8 = false;
      if (f8 && a.sentinel == 2) {
        KlScanFwdWide a8 = a;
        a8.xcc_slots = h->fwd8_local ? w.scan_status + 4 : nullptr;
        a8.pf_mode = h->fwd8_pf >= 0 ? h->fwd8_pf : (h->fwd8_ls ? 0 : 1);      // (measured at 3072 streams: two-barrier form 3.06 / 3.32 / 3.30 ms per launch with pf 0 / 1 / 3)
        const int e8 = kl_launch_scan_fwd8(a8, s, h->fwd8_ls);
        if (e8 != KL_ERR_SHAPE) KL_TRY(e8);
        took8 = e8 == 0;
      }
      if (!took8 && f8_tab) {      // (the eight-wave scan did not take the shape: the 16-wave scan's own table mode)
        KL_TRY(kl_launch_ids_tm(idx, ctx, c.n_ctx, B, T, W, c.voc_size, c.ctx_vocab, w.ids_tm, s));
        a.P = nullptr; a.p_bf16 = 0;
        a.EK = d.EKp;
        a.CtxK[0] = d.CtxKp[0];
        a.n_ctx = c.n_ctx;
        a.ids_tm = w.ids_tm;
      }
      if (took8) {}
      else if (v2) KL_TRY(kl_launch_scan_fwd_wide2(a, w.scan2_rows, s));
      else KL_TRY(kl_launch_scan_fwd_wide(a, s));
      {
        h->trace_persistent[0] = true;
        h->trace_name[0] = took8 ? "lstm_scan_fwd8_kernel" : v2 ? "lstm_scan_fwd_wide2_kernel" : "lstm_scan_fwd_wide_kernel";
        h->trace_flops[0] = (double)B * T * (2.0 * W * 4.0 * W);   // one layer's recurrent contraction
        h->trace_end(0, s);
      }
    }
    scanned = true;
    w.ht_ready = !w.km_plan;
  }
  // Width 128 (the reference's own model sizes): a workgroup = a 16-row block of streams with ALL hidden units of a layer -- no
  // hand-off of state between workgroups (lstm_scan_w128.hip).  Layer 0 takes its gate inputs straight from the look-up tables
  // (up to two context variables; else the rows gathered here), the layers above it contract their inputs inside the scan.
  const bool w128_path = !scanned && training && h->scan_enabled && h->w128 && B >= h->w128_min && kl_scan_w128_applicable(B, T, W);
  const bool w128_tabs = w128_path && h->w128_tables && c.n_ctx <= kl_scan_w128_tables_max_ctx();
  if (!scanned && !w128_tabs)
    KL_TRY(kl_launch_p1_gather(d.EK, ctxk.data(), c.n_ctx, P + h->off_b[0], idx, ctx, B, T, 4 * W, w.P1, s));
  if (w128_path) {
    KlScanFwdWide layer_args[KL_SCAN_MAXL];
    const bool try_multi = h->w128_multi && h->w128_fuse && L >= 2 && L <= KL_SCAN_MAXL;
    for (int l = 0; l < L; ++l) {
      const bool masked = masks != nullptr && l > 0;
      KlScanFwdWide& a = layer_args[l < KL_SCAN_MAXL ? l : 0];
      memset(&a, 0, sizeof(a));
      a.B = B; a.T = T; a.W = W;
      a.UT = d.UT_hi[l];
      a.P = w.P1;
      if (l == 0 && w128_tabs) {
        a.P = nullptr; a.EK = d.EK; a.n_ctx = c.n_ctx; a.idx = idx; a.ctx = ctx; a.bias = P + h->off_b[0];
        for (int n = 0; n < c.n_ctx; ++n) a.CtxK[n] = ctxk[n];
      }
      if (l > 0) {
        const bool masked_in = masks != nullptr && (l - 1) > 0;
        const bf16_t* X = masked_in ? w.Hd[l - 1] : (const bf16_t*)w.H[l - 1] + BW;
        if (h->w128_fuse) {      // the input contraction inside the scan: no P rows, no product over all steps
          a.P = nullptr; a.KT = d.KT_hi[l]; a.X = X; a.bias = P + h->off_b[l];
        } else {
          KL_TRY(kl_launch_gemm_tn(X, d.KT_hi[l], w.P1, P + h->off_b[l], B * T, 4 * W, W, W, W, 4 * W, 0, 1, 1.f, s));
        }
      }
      a.H = (bf16_t*)w.H[l]; a.C = w.C[l]; a.G = w.G[l];
      a.Hd = masked ? w.Hd[l] : nullptr;
      a.mask = masked ? masks + (size_t)l * BW : nullptr;
      a.status = w.scan_status;
      if (try_multi) continue;
      h->trace_begin(0, s);
      KL_TRY(kl_launch_scan_fwd_w128(a, s));
      h->trace_persistent[0] = true;
      h->trace_name[0] = "lstm_scan_fwd_w128_kernel";
      h->trace_flops[0] = (double)B * T * (2.0 * W * 4.0 * W);
      h->trace_end(0, s);
    }
    if (try_multi) {
      // All layers in one launch where that fits the CUs (layers x 16-row blocks <= 256: up to 2048 streams at depth 2): a layer
      // follows the one below it a step behind, polling the rows that one publishes -- which therefore start out as sentinels.
      bool multi = kl_scan_w128_multi_fits(B, L);
      if (multi) {
        const bool roll = h->sentinel_roll && T >= 3;      // (rolling sentinels: only the first two steps start out armed)
        for (int l = 1; l < L; ++l) {
          layer_args[l - 1].sentinel = roll ? 2 : 1;
          KL_TRY(kl_fill_u32_async(const_cast<bf16_t*>(layer_args[l].X), (size_t)(roll ? 2 : T) * BW * sizeof(bf16_t), 0xFFFFFFFFu, s));
        }
        h->trace_begin(0, s);
        const int e = kl_launch_scan_fwd_w128_multi(layer_args, L, s);
        if (e == KL_ERR_SHAPE) multi = false;
        else KL_TRY(e);
      }
      if (multi) {
        h->trace_persistent[0] = true;
        h->trace_name[0] = "lstm_scan_fwd_w128_multi_kernel";
        h->trace_flops[0] = (double)L * B * T * (2.0 * W * 4.0 * W) + (double)(L - 1) * B * T * (2.0 * W * 4.0 * W);
        h->trace_end(0, s);
      } else {
        for (int l = 0; l < L; ++l) {
          h->trace_begin(0, s);
          KL_TRY(kl_launch_scan_fwd_w128(layer_args[l], s));
          h->trace_persistent[0] = true;
          h->trace_name[0] = "lstm_scan_fwd_w128_kernel";
          h->trace_flops[0] = (double)B * T * (2.0 * W * 4.0 * W);
          h->trace_end(0, s);
        }
      }
    }
    scanned = true;
  }
  // Width 1024 (the cfg5 topology): the recurrent weights of 16 units alone fill half of a 256-thread
  // workgroup's registers, so neither the fused scans (U and K resident) nor the 64-unit wide scans apply.
  // One layer per launch with the thin workgroups instead, the input side from one GEMM over all steps as
  // in the wide path (layer 0: the table gather above).
  // The same layer-by-layer launches serve models deeper than the fused scans' four layers (the reference takes up to
  // ten, scripts/run.py:35).
  if (!scanned && training && h->scan_enabled && (W == 1024 || L > KL_SCAN_MAXL)) {
    bool all = true;
    for (int l = 0; l < L && all; ++l) {
      if (l > 0) {
        const bool masked_in = masks != nullptr && (l - 1) > 0;
        const bf16_t* X = masked_in ? w.Hd[l - 1] : (const bf16_t*)w.H[l - 1] + BW;
        KL_TRY(kl_launch_gemm_tn(X, d.KT_hi[l], w.P1, P + h->off_b[l], B * T, 4 * W, W, W, W, 4 * W, 0, 1, 1.f, s));
      }
      const bool masked = masks != nullptr && l > 0;
      KlScanFwd a;
      memset(&a, 0, sizeof(a));
      a.B = B; a.T = T; a.W = W; a.L = 1;
      a.UT[0] = d.UT_hi[l];
      a.H[0] = (bf16_t*)w.H[l];
      a.C[0] = w.C[l];
      a.G[0] = w.G[l];
      a.Hd[0] = masked ? w.Hd[l] : nullptr;
      a.mask[0] = masked ? masks + (size_t)l * BW : nullptr;
      a.P1 = w.P1;
      a.counters = w.scan_cnt;
      a.status = w.scan_status;
      // width 1024 from eight row blocks: eight-wave workgroups of 32 units, the state tile through LDS, hand-off by
      // data sentinels inside one XCD (lstm_scan_w32.hip); else the thin workgroups with their counters
      const bool w32 = h->w32 && (B + 15) / 16 >= h->w32_min_rb && kl_scan_w32_applicable(B, T, W);
      if (w32) {
        a.sentinel = 1;
        a.xcc_slots = h->w32_local ? w.scan_status + 4 : nullptr;
        a.gen = (unsigned)(1 + l);                   // (the posts are zeroed once per window: a token per launch)
        KL_TRY(kl_fill_u32_async((bf16_t*)w.H[l] + BW, (size_t)T * BW * sizeof(bf16_t), 0xFFFFFFFFu, s));
      } else {
        KL_TRY(kl_zero_coherent_async(w.scan_cnt, (size_t)((B + 15) / 16) * T, s));
      }
      h->trace_begin(0, s);      // (every layer's scan launch is timed while tracing)
      const int e = w32 ? kl_launch_scan_fwd_w32(a, s) : kl_launch_scan_fwd(a, s);
      if (e == KL_ERR_SHAPE && l == 0) {
        all = false;          // (too many row blocks: the launch-per-step path below takes the window, P1 is in place)
      } else if (e != 0) {
        return e;
      } else {
        h->trace_persistent[0] = true;
        h->trace_name[0] = w32 ? "lstm_scan_fwd_w32_kernel" : "lstm_scan_fwd_kernel";
        h->trace_flops[0] = (double)B * T * (2.0 * W * 4.0 * W);
        h->trace_end(0, s);
      }
    }
    if (all) scanned = true;
    else KL_TRY(kl_launch_p1_gather(d.EK, ctxk.data(), c.n_ctx, P + h->off_b[0], idx, ctx, B, T, 4 * W, w.P1, s));
  }
  // rating windows in split precision at width 1024 (cfg5): the (hi, lo) planes of the recurrent weights of 16 units are a
  // workgroup's whole register budget, so the layers run one after the other -- per layer ONE split-precision product
  // over all steps for the input side (layer 0: the table gather above), then the persistent scan with U alone
  // (was: a launch per step and layer wavefront, 18 ms per 1 x 512 window)
  // ... and with few streams (two row blocks at most): eight units per workgroup, so that U and K fit and TWO layers
  // run as a wavefront per launch (256 workgroups, one per CU): half the chain
  if (!scanned && !training && split == KL_PREC_SPLIT && h->scan_enabled && h->split_sentinel && h->split8 && W == 1024 &&
      (B + 15) / 16 <= 2 && (long)(T + 1) * B * W * 2 <= 0x7fffffffL) {
    bool all = true;
    for (int l0 = 0; l0 < L && all; l0 += 2) {
      const int nl = L - l0 < 2 ? L - l0 : 2;
      KlScanFwdSplit a;
      memset(&a, 0, sizeof(a));
      a.B = B; a.T = T; a.W = W; a.L = nl; a.l0 = l0; a.units8 = 1;
      if (l0 + nl > KL_SCAN_MAXL) { all = false; break; }
      for (int l = (l0 > 0 ? l0 - 1 : 0); l < l0 + nl; ++l) {
        a.UT_hi[l] = d.UT_hi[l]; a.UT_lo[l] = d.UT_lo[l];
        a.KT_hi[l] = l > 0 ? d.KT_hi[l] : nullptr; a.KT_lo[l] = l > 0 ? d.KT_lo[l] : nullptr;
        a.bias[l] = l > 0 ? P + h->off_b[l] : nullptr;
        a.Xhi[l] = w.Xhi[l]; a.Xlo[l] = w.Xlo[l];
        a.Hf[l] = (float*)w.H[l];
        a.C[l] = w.C[l];
      }
      a.P1 = w.P1;
      a.counters = w.scan_cnt;
      a.status = w.scan_status;
      a.sentinel = 1;
      for (int l = l0; l < l0 + nl; ++l) {
        KL_TRY(kl_launch_f32_to_bf16_t((const float*)w.H[l], W, B, W, w.Xhi[l], w.Xlo[l], W, 0, s));
        KL_TRY(kl_fill_u32_async(w.Xhi[l] + BW, (size_t)T * BW * sizeof(bf16_t), 0xFFFFFFFFu, s));
        KL_TRY(kl_fill_u32_async(w.Xlo[l] + BW, (size_t)T * BW * sizeof(bf16_t), 0xFFFFFFFFu, s));
      }
      const int e = kl_launch_scan_fwd_split(a, s);
      if (e == KL_ERR_SHAPE && l0 == 0) all = false;      // (fewer than 256 CUs: layer by layer below)
      else if (e != 0) return e;
    }
    if (all) scanned = true;
  }
  if (!scanned && !training && split == KL_PREC_SPLIT && h->scan_enabled && h->split_sentinel && W == 1024 &&
      (B + 15) / 16 <= 16 && (long)(T + 1) * B * W * 2 <= 0x7fffffffL) {
    for (int l = 0; l < L; ++l) {
      if (l > 0 && B * T >= 2048) {
        // many rows: the three products of the split (hi.hi + bias, lo.hi, hi.lo) on the big-tile GEMM, over the (hi, lo)
        // planes the scan below left behind (the thin GEMM re-reads the weights per 32 rows: 2.9 ms per layer at 16 x 512 rows)
        const bf16_t* xh = w.Xhi[l - 1] + BW;
        const bf16_t* xl = w.Xlo[l - 1] + BW;
        KL_TRY(kl_launch_gemm_tn(xh, d.KT_hi[l], w.P1, P + h->off_b[l], B * T, 4 * W, W, W, W, 4 * W, 0, 1, 1.f, s));
        KL_TRY(kl_launch_gemm_tn(xl, d.KT_hi[l], w.P1, nullptr, B * T, 4 * W, W, W, W, 4 * W, 2, 1, 1.f, s));
        KL_TRY(kl_launch_gemm_tn(xh, d.KT_lo[l], w.P1, nullptr, B * T, 4 * W, W, W, W, 4 * W, 2, 1, 1.f, s));
      } else if (l > 0) {
        KlOperand op;
        memset(&op, 0, sizeof(op));
        op.A = (float*)w.H[l - 1] + BW; op.lda = W; op.a_is_f32 = 1;
        op.WT_hi = d.KT_hi[l]; op.WT_lo = d.KT_lo[l]; op.ldw = W; op.K = W;
        KL_TRY(kl_launch_thin_gemm(&op, B * T, 4 * W, w.P1, 4 * W, P + h->off_b[l], KL_PREC_SPLIT, s));
      }
      KlScanFwdSplit a;
      memset(&a, 0, sizeof(a));
      a.B = B; a.T = T; a.W = W; a.L = 1;
      a.UT_hi[0] = d.UT_hi[l]; a.UT_lo[0] = d.UT_lo[l];
      a.Xhi[0] = w.Xhi[l]; a.Xlo[0] = w.Xlo[l];
      a.Hf[0] = (float*)w.H[l];
      a.C[0] = w.C[l];
      a.P1 = w.P1;
      a.counters = w.scan_cnt;
      a.status = w.scan_status;
      a.sentinel = 1;
      KL_TRY(kl_launch_f32_to_bf16_t((const float*)w.H[l], W, B, W, w.Xhi[l], w.Xlo[l], W, 0, s));
      KL_TRY(kl_fill_u32_async(w.Xhi[l] + BW, (size_t)T * BW * sizeof(bf16_t), 0xFFFFFFFFu, s));
      KL_TRY(kl_fill_u32_async(w.Xlo[l] + BW, (size_t)T * BW * sizeof(bf16_t), 0xFFFFFFFFu, s));
      KL_TRY(kl_launch_scan_fwd_split(a, s));
    }
    scanned = true;
  }
  // rating windows in split precision: one persistent launch for all layers and steps
  if (!scanned && !training && split == KL_PREC_SPLIT && h->scan_enabled && L <= KL_SCAN_MAXL) {
    KlScanFwdSplit a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.T = T; a.W = W; a.L = L;
    for (int l = 0; l < L; ++l) {
      a.UT_hi[l] = d.UT_hi[l]; a.UT_lo[l] = d.UT_lo[l];
      a.KT_hi[l] = l > 0 ? d.KT_hi[l] : nullptr; a.KT_lo[l] = l > 0 ? d.KT_lo[l] : nullptr;
      a.bias[l] = l > 0 ? P + h->off_b[l] : nullptr;
      a.Xhi[l] = w.Xhi[l]; a.Xlo[l] = w.Xlo[l];
      a.Hf[l] = (float*)w.H[l];
      a.C[l] = w.C[l];
    }
    a.P1 = w.P1;
    a.counters = w.scan_cnt;
    a.status = w.scan_status;
    // the carried-in state as (hi, lo) planes; hand-off counters zeroed write-through
    for (int l = 0; l < L; ++l)
      KL_TRY(kl_launch_f32_to_bf16_t((const float*)w.H[l], W, B, W, w.Xhi[l], w.Xlo[l], W, 0, s));
    a.sentinel = h->split_sentinel ? 1 : 0;
    if (a.sentinel) {      // hand-off by data: the blocks the scan is going to publish start out as sentinels
      for (int l = 0; l < L; ++l) {
        KL_TRY(kl_fill_u32_async(w.Xhi[l] + BW, (size_t)T * BW * sizeof(bf16_t), 0xFFFFFFFFu, s));
        KL_TRY(kl_fill_u32_async(w.Xlo[l] + BW, (size_t)T * BW * sizeof(bf16_t), 0xFFFFFFFFu, s));
      }
    } else {
      KL_TRY(kl_zero_coherent_async(w.scan_cnt, (size_t)L * ((B + 15) / 16) * T, s));
    }
    const int e = kl_launch_scan_fwd_split(a, s);
    if (e == 0) scanned = true;
    else if (e != KL_ERR_SHAPE) return e;
  }
  // persistent scan (one launch for all layers and steps) where the shape allows it
  if (!scanned && training && h->scan_enabled && L <= KL_SCAN_MAXL) {
    KlScanFwd a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.T = T; a.W = W; a.L = L;
    for (int l = 0; l < L; ++l) {
      a.UT[l] = d.UT_hi[l];
      a.KT[l] = l > 0 ? d.KT_hi[l] : nullptr;
      a.bias[l] = l > 0 ? P + h->off_b[l] : nullptr;
      a.H[l] = (bf16_t*)w.H[l];
      a.C[l] = w.C[l];
      a.G[l] = w.G[l];
      const bool masked = masks != nullptr && l > 0;
      a.Hd[l] = masked ? w.Hd[l] : nullptr;
      a.mask[l] = masked ? masks + (size_t)l * BW : nullptr;
    }
    a.P1 = w.P1;
    a.counters = w.scan_cnt;
    a.status = w.scan_status;
    KL_TRY(kl_zero_coherent_async(w.scan_cnt, (size_t)L * ((B + 15) / 16) * T, s));
    h->trace_begin(0, s);
    const int e = kl_launch_scan_fwd(a, s);
    if (e == 0) {
      scanned = true;
      h->trace_persistent[0] = true;
      h->trace_name[0] = "lstm_scan_fwd_kernel";
      h->trace_flops[0] = (double)B * T * (2.0 * (2.0 * L - 1.0) * W * 4.0 * W);   // U_l for all l, K_l for l >= 1
      h->trace_end(0, s);
    } else if (e != KL_ERR_SHAPE) {
      return e;
    }
  }
  for (int dgl = 0; !scanned && dgl < T + L - 1; ++dgl) {
    KlFwdStep steps[16];     // one per layer (config_ok: depth <= 16); launched in packs of 4 below
    int ns = 0;
    for (int l = 0; l < L; ++l) {
      const int t = dgl - l;
      if (t < 0 || t >= T) continue;
      KlFwdStep& S = steps[ns++];
      memset(&S, 0, sizeof(S));
      S.n_rows = B; S.W = W; S.split = split;
      int p = 0;
      if (l > 0) {
        KlOperand& o = S.op[p++];
        const bool masked_in = training && masks != nullptr && (l - 1) > 0;
        if (masked_in) {
          o.A = w.Hd[l - 1] + (size_t)t * BW; o.a_is_f32 = 0;
        } else {
          o.A = hrow(l - 1, t + 1); o.a_is_f32 = training ? 0 : 1;
        }
        o.lda = W; o.WT_hi = d.KT_hi[l]; o.WT_lo = split == 3 ? d.KT_lo[l] : nullptr; o.ldw = W; o.K = W;
        S.bias = P + h->off_b[l];
      } else {
        S.T1 = w.P1 + (size_t)t * B * 4 * W; S.t1_ld = 4 * W;
      }
      {
        KlOperand& o = S.op[p++];
        o.A = hrow(l, t); o.a_is_f32 = training ? 0 : 1; o.lda = W;
        o.WT_hi = d.UT_hi[l]; o.WT_lo = split == 3 ? d.UT_lo[l] : nullptr; o.ldw = W; o.K = W;
      }
      S.n_ops = p;
      S.c_prev = w.C[l] + (size_t)t * BW; S.c_prev_ld = W;
      S.c_out = w.C[l] + (size_t)(t + 1) * BW; S.c_out_ld = W;
      if (training) {
        S.h_out_bf16 = (bf16_t*)w.H[l] + (size_t)(t + 1) * BW; S.h_out_bf16_ld = W;
        S.gates_out = w.G[l] + (size_t)t * B * 4 * W; S.gates_ld = 4 * W;
        if (l > 0 && masks != nullptr) {
          S.hmask = masks + (size_t)l * BW; S.hmask_ld = W;
          S.hd_out_bf16 = w.Hd[l] + (size_t)t * BW; S.hd_out_ld = W;
        }
      } else {
        S.h_out_f32 = (float*)w.H[l] + (size_t)(t + 1) * BW; S.h_out_f32_ld = W;
      }
    }
    // a launch packs at most 4 independent steps
    for (int i = 0; i < ns; i += 4) {
      // trace: one event pair brackets 8 consecutive steady-state launches (dgl = 8k .. 8k+7)
      const bool steady = (ns == L) && dgl >= L && dgl + 8 < T;
      if (steady && dgl % 8 == 0) h->trace_begin(0, s);
      KL_TRY(kl_launch_fwd_steps(steps + i, ns - i < 4 ? ns - i : 4, s));
      if (steady && dgl % 8 == 7) h->trace_end(0, s);
    }
  }
  for (int l = 0; l < L; ++l)
    KL_TRY(kl_launch_rows_to_state(hrow(l, T), training ? 0 : 1, w.C[l] + (size_t)T * BW, B, W, L, l, states, s));
  return 0;
}

__global__ void scan_status_kernel(const unsigned* status, float* loss_acc) {
  if (threadIdx.x == 0 && (status[0] | status[1]) != 0) loss_acc[3] += 1.f;
}

__global__ void state_dist2_kernel(const float* __restrict__ pool, long slot_stride, int W, int k,
                                   const int* __restrict__ a, const int* __restrict__ b, int n,
                                   float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= n) return;
  const float* pa = pool + (long)a[i] * slot_stride + (long)k * W;
  const float* pb = pool + (long)b[i] * slot_stride + (long)k * W;
  float acc = 0.f;
  for (int u = lane; u < W; u += 64) { const float q = pa[u] - pb[u]; acc += q * q; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) out[i] = acc;
}

}  // namespace

extern "C" {

int kl_abi_version(void) { return KL_ABI_VERSION; }

const char* kl_error_string(int code) {
  switch (code) {
    case KL_OK: return "ok";
    case KL_ERR_SHAPE: return "unsupported or inconsistent dimensions";
    case KL_ERR_LAUNCH: return "HIP launch/runtime error";
    case KL_ERR_STATE: return "call order violated (bind/prepare first)";
    case KL_ERR_WORKSPACE: return "workspace too small";
    case KL_ERR_ARG: return "null or invalid argument";
    default: return "unknown error";
  }
}

size_t kl_param_count(const kl_config* cfg) {
  if (!config_ok(cfg)) return 0;
  kl_handle tmp;
  tmp.cfg = *cfg;
  layout(&tmp);
  return tmp.n_params;
}

int kl_param_layout(const kl_config* cfg, int index, char* name, size_t name_cap, size_t* offset, size_t* rows,
                    size_t* cols) {
  if (!config_ok(cfg) || index < 0) return KL_ERR_ARG;
  kl_handle t;
  t.cfg = *cfg;
  layout(&t);
  const size_t W = cfg->width;
  char buf[32];
  size_t off, r, c;
  const int n_emb = 1 + cfg->n_ctx;
  if (index == 0) {
    snprintf(buf, sizeof buf, "E"); off = t.off_E; r = cfg->voc_size; c = W;
  } else if (index < n_emb) {
    snprintf(buf, sizeof buf, "Ctx%d", index - 1); off = t.off_Ctx[index - 1]; r = cfg->ctx_vocab; c = cfg->ctx_dim;
  } else {
    const int j = index - n_emb, l = j / 3, k = j % 3;
    if (l >= cfg->depth) return KL_ERR_ARG;
    const size_t D = l == 0 ? W + (size_t)cfg->ctx_dim * cfg->n_ctx : W;
    if (k == 0) { snprintf(buf, sizeof buf, "K%d", l); off = t.off_K[l]; r = D; c = 4 * W; }
    else if (k == 1) { snprintf(buf, sizeof buf, "U%d", l); off = t.off_U[l]; r = W; c = 4 * W; }
    else { snprintf(buf, sizeof buf, "b%d", l); off = t.off_b[l]; r = 1; c = 4 * W; }
  }
  if (name && name_cap) { strncpy(name, buf, name_cap - 1); name[name_cap - 1] = 0; }
  if (offset) *offset = off;
  if (rows) *rows = r;
  if (cols) *cols = c;
  return 0;
}

kl_handle* kl_create(const kl_config* cfg) {
  if (!config_ok(cfg)) return nullptr;
  kl_handle* h = new kl_handle();
  h->cfg = *cfg;
  layout(h);
  // (read HERE, not with the other switches in kl_bind: it decides how large the derived workspace is -- kl_derived_bytes)
  const char* env8tb = getenv("KL_FWD8_TAB");
  if (env8tb) h->fwd8_tab = atoi(env8tb) != 0;
  return h;
}

void kl_destroy(kl_handle* h) {
  if (h) {
    h->drop_graphs();
    for (auto& list : h->trace_ev)
      for (auto& ev : list) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
      }
  }
  delete h;
}

size_t kl_derived_bytes(const kl_handle* h) { return h ? carve_derived(h, nullptr, nullptr) : 0; }

int kl_bind(kl_handle* h, float* params, void* derived, size_t derived_bytes) {
  if (!h || !params || !derived) return KL_ERR_ARG;
  if (derived_bytes < carve_derived(h, nullptr, nullptr)) return KL_ERR_WORKSPACE;
  h->params = params;
  h->derived_ws = derived;
  h->derived_bytes = derived_bytes;
  carve_derived(h, derived, &h->d);
  h->precision = 0;
  h->drop_graphs();
  const char* env = getenv("KL_GRAPH");
  h->graphs_enabled = !(env && env[0] == '0');
  const char* env2 = getenv("KL_SCAN");
  h->scan_enabled = !(env2 && env2[0] == '0');
  const char* env3 = getenv("KL_SEQ_BWD");
  h->seq_bwd = !(env3 && env3[0] == '0');
  const char* env4 = getenv("KL_WIDE_BWD");
  h->wide_bwd = !(env4 && env4[0] == '0');
  const char* env7 = getenv("KL_SENTINEL");
  h->sentinel = !(env7 && env7[0] == '0');
  const char* env7e = getenv("KL_GEMM_AN");
  h->gemm_an = !(env7e && env7e[0] == '0');
  const char* env7c = getenv("KL_XCD_LOCAL");
  h->xcd_local = env7c && env7c[0] == '1';
  const char* env7b = getenv("KL_SENTINEL_BWD");
  h->sentinel_bwd = h->sentinel && !(env7b && env7b[0] == '0');
  h->sentinel_bwd_all = env7b && env7b[0] == '2';
  const char* env7f = getenv("KL_SENTINEL_ROLL");
  h->sentinel_roll = !(env7f && env7f[0] == '0');
  const char* env7d = getenv("KL_XCD_LOCAL_BWD");
  h->xcd_local_bwd = h->xcd_local && env7d && env7d[0] == '1';
  const char* env6f = getenv("KL_W32");
  h->w32 = !(env6f && env6f[0] == '0');
  const char* env6g = getenv("KL_W32_LOCAL");
  h->w32_local = env6g && env6g[0] == '1';
  const char* env6h = getenv("KL_W32_MIN_RB");
  h->w32_min_rb = env6h ? atoi(env6h) : 8;
  const char* env6j = getenv("KL_SPLIT8");
  h->split8 = !(env6j && env6j[0] == '0');
  const char* env6c = getenv("KL_SPLIT_SENTINEL");
  h->split_sentinel = !(env6c && env6c[0] == '0');
  const char* env6b = getenv("KL_INC_SMALL");
  h->inc_small = !(env6b && env6b[0] == '0');
  const char* env6 = getenv("KL_FUSED_STEP");
  h->fused_step = !(env6 && env6[0] == '0');
  const char* env6k = getenv("KL_INC_TILE");
  h->inc_tile = !(env6k && env6k[0] == '0');
  const char* env6l = getenv("KL_TILE_VAR");
  h->tile_var = env6l ? atoi(env6l) : 0;
  const char* env6r = getenv("KL_TILE_ROWS");
  if (env6r) h->tile_rows = atoi(env6r);
  const char* env6m = getenv("KL_OUT_FUSED");
  h->out_fused = !(env6m && env6m[0] == '0');
  const char* env6n = getenv("KL_OUT_FUSED_MIN");
  if (env6n) h->out_fused_min = atoi(env6n);
  const char* env6o = getenv("KL_INC_SMALL_MIN");
  if (env6o) h->inc_small_min = atoi(env6o);
  const char* env6p = getenv("KL_HOST_KERNARG");
  h->host_kernarg = !(env6p && env6p[0] == '0');
  const char* env8 = getenv("KL_SCAN2");
  if (env8) h->scan2 = atoi(env8) != 0;
  const char* env8b = getenv("KL_SCAN2_ROWS");
  if (env8b) h->scan2_rows = atoi(env8b);
  const char* env8c = getenv("KL_SCAN2_PF");
  if (env8c) h->scan2_pf = atoi(env8c);
  const char* env8e = getenv("KL_SCAN2_BF16");
  if (env8e) h->scan2_bf16 = atoi(env8e) != 0;
  const char* env8i = getenv("KL_LOGITS_WS");
  if (env8i) h->logits_ws = atoi(env8i) != 0;
  const char* env8h = getenv("KL_PROJ_WS");
  if (env8h) h->proj_ws = atoi(env8h) != 0;
  const char* env8p = getenv("KL_W128");
  if (env8p) h->w128 = atoi(env8p) != 0;
  const char* env8s = getenv("KL_W128_FUSE");
  if (env8s) h->w128_fuse = atoi(env8s) != 0;
  const char* env8t = getenv("KL_W128_TABLES");
  if (env8t) h->w128_tables = atoi(env8t) != 0;
  const char* env8mm = getenv("KL_W128_MULTI");
  if (env8mm) h->w128_multi = atoi(env8mm) != 0;
  const char* env8q = getenv("KL_W128_MIN");
  if (env8q) h->w128_min = atoi(env8q);
  const char* env8m = getenv("KL_FWD8");
  if (env8m) { h->fwd8 = atoi(env8m) != 0; h->fwd8_all = atoi(env8m) == 2; }
  const char* env8r = getenv("KL_FWD8_LS");
  if (env8r) h->fwd8_ls = atoi(env8r) != 0;
  const char* env8n = getenv("KL_FWD8_LOCAL");
  if (env8n) h->fwd8_local = atoi(env8n) != 0;
  const char* env8o = getenv("KL_FWD8_PF");
  if (env8o) h->fwd8_pf = atoi(env8o);
  const char* env8l = getenv("KL_RT_LOCAL");
  if (env8l) h->rt_local = atoi(env8l) != 0;
  const char* env8k = getenv("KL_REGTILE");
  if (env8k) h->regtile = atoi(env8k) != 0;
  const char* env8g = getenv("KL_SCAN2_FLAGS");
  if (env8g) h->scan2_flags = atoi(env8g) != 0;
  h->flags_zeroed = false;
  const char* env8f = getenv("KL_FUSE_WG");
  if (env8f) h->fuse_wg = atoi(env8f) != 0;
  const char* env8d = getenv("KL_SCAN2_PFB");
  if (env8d) h->scan2_pfb = atoi(env8d);
  const char* env5 = getenv("KL_WIDE_FWD_MIN");
  if (env5) h->wide_fwd_min = atoi(env5);
  return kl_zero_page_ready();
}

int kl_prepare(kl_handle* h, int precision, void* stream) {
  if (!h) return KL_ERR_ARG;
  if (!h->params) return KL_ERR_STATE;
  if (precision != KL_PREC_BF16 && precision != KL_PREC_SPLIT) return KL_ERR_ARG;
  return prepare_impl(h, precision, (hipStream_t)stream);
}

size_t kl_window_workspace_bytes(const kl_handle* h, int B, int T, int training) {
  if (!h || B < 1 || T < 1) return 0;
  return carve_window(h, nullptr, B, T, training, nullptr);
}

static int forward_window_body(kl_handle* h, int B, int T, const int32_t* idx, const int32_t* ctx, const int32_t* tgt,
                      float* states, float* probs, float* loss_acc, void* ws, size_t ws_bytes, void* stream) {
  if (!h || !idx || !states || !ws || B < 1 || T < 1) return KL_ERR_ARG;
  if (h->cfg.n_ctx > 0 && !ctx) return KL_ERR_ARG;
  if (!h->precision) return KL_ERR_STATE;
  hipStream_t s = (hipStream_t)stream;
  WindowWs w;
  const int W = h->cfg.width, V = h->cfg.voc_size, L = h->cfg.depth;
  // Windows in bf16 precision (validation after each epoch, rating.py:300-306: a fifth of every epoch's data) with
  // a training-size workspace take the TRAINING forward -- persistent wide / fused scans and the big GEMMs -- without
  // the backward, instead of the launch-per-step inference kernels (1024 x 256 characters: 41 ms -> 4.6 ms).
  if (h->precision == KL_PREC_BF16 && ws_bytes >= carve_window(h, nullptr, B, T, 1, nullptr)) {
    carve_window(h, ws, B, T, 1, &w);
    const int BT = B * T;
    KL_TRY(kl_zero_async(w.scan_status, (4 + 256) * sizeof(unsigned), s));
    w.km_plan = true;                  // (no transposed outputs: nothing is going to contract over the rows)
    w.scan2_rows = plan_scan2(h, B, T, true, false);
    KL_TRY(forward_impl(h, B, T, idx, ctx, states, nullptr, 1, w, s));
    const bf16_t* Htop = (const bf16_t*)w.H[L - 1] + (size_t)B * W;
    KL_TRY(kl_launch_gemm_tn(Htop, h->d.E_hi, w.logits, nullptr, BT, V, W, W, W, V, 0, 1, 1.f, s));
    const int mean_rows = (h->loss_rows > 0 && h->loss_rows <= B) ? h->loss_rows : B;      // (a padded batch: kl_set_loss_rows)
    KL_TRY(kl_launch_softmax_ce(w.logits, V, BT, V, tgt, B, T, 1.0f / (h->last_only ? (float)mean_rows : (float)mean_rows * (float)T),
                                nullptr, 0, tgt ? loss_acc : nullptr, w.rowstat, 1, s, h->last_only));
    if (probs) {
      if (B == 1) KL_TRY(hip_ok(hipMemcpyAsync(probs, w.logits, (size_t)T * V * sizeof(float), hipMemcpyDeviceToDevice, s)));
      else KL_TRY(kl_launch_rows_tm_to_bm(w.logits, V, probs, B, T, V, s));
    }
    if (loss_acc) hipLaunchKernelGGL(scan_status_kernel, dim3(1), dim3(64), 0, s, w.scan_status, loss_acc);
    return hip_ok(hipGetLastError());
  }
  if (ws_bytes < carve_window(h, ws, B, T, 0, &w)) return KL_ERR_WORKSPACE;
  KL_TRY(kl_zero_async(w.scan_status, (4 + 256) * sizeof(unsigned), s));
  KL_TRY(forward_impl(h, B, T, idx, ctx, states, nullptr, 0, w, s));
  KlOperand op;
  memset(&op, 0, sizeof(op));
  op.A = (float*)w.H[L - 1] + (size_t)B * W; op.lda = W; op.a_is_f32 = 1;
  op.WT_hi = h->d.E_hi; op.WT_lo = h->precision == 3 ? h->d.E_lo : nullptr; op.ldw = W; op.K = W;
  KL_TRY(kl_launch_thin_gemm(&op, B * T, V, w.logits, V, nullptr, h->precision, s));
  KL_TRY(kl_launch_softmax_ce(w.logits, V, B * T, V, tgt, B, T, 1.0f / (h->last_only ? (float)B : (float)B * T), nullptr, 0,
                              tgt ? loss_acc : nullptr, w.rowstat, 1, s, h->last_only));
  if (probs) {
    if (B == 1) KL_TRY(hip_ok(hipMemcpyAsync(probs, w.logits, (size_t)T * V * sizeof(float), hipMemcpyDeviceToDevice, s)));
    else KL_TRY(kl_launch_rows_tm_to_bm(w.logits, V, probs, B, T, V, s));
  }
  // a timed-out hand-off in the persistent scan surfaces as loss_acc[3] != 0
  if (loss_acc) hipLaunchKernelGGL(scan_status_kernel, dim3(1), dim3(64), 0, s, w.scan_status, loss_acc);
  return hip_ok(hipGetLastError());
}

static int train_window_body(kl_handle* h, int B, int T, const int32_t* idx, const int32_t* ctx, const int32_t* tgt,
                    float* states, const float* masks, float* grads, float* loss_acc, void* ws, size_t ws_bytes,
                    void* stream) {
  if (!h || !idx || !tgt || !states || !grads || !ws || B < 1 || T < 1) return KL_ERR_ARG;
  if (h->cfg.n_ctx > 0 && !ctx) return KL_ERR_ARG;
  if (h->precision != KL_PREC_BF16) return KL_ERR_STATE;
  hipStream_t s = (hipStream_t)stream;
  WindowWs w;
  if (ws_bytes < carve_window(h, ws, B, T, 1, &w)) return KL_ERR_WORKSPACE;
  const kl_config& c = h->cfg;
  const int W = c.width, V = c.voc_size, Vp = h->Vp, L = c.depth;
  const int BT = B * T, BTp = round_up_i(BT, 8);
  const size_t BW = (size_t)B * W;
  const float* P = h->params;
  Derived& d = h->d;
  const int ksplit = BT >= 4096 ? 8 : (BT >= 1024 ? 4 : 1);

  KL_TRY(kl_zero_async(grads, h->n_params * sizeof(float), s));
  KL_TRY(kl_zero_async(w.scan_status, (4 + 256) * sizeof(unsigned), s));
  // (M = 4W rows of dZ as the K-major A operand: W % 64 == 0, T*B % 64 == 0)
  w.km_plan = h->gemm_an && BTp == BT && kl_gemm_an_applicable(4 * W, W, BT, 4 * W);
  w.scan2_rows = plan_scan2(h, B, T, w.km_plan, true);
  w.scan2_bwd = w.scan2_rows != 0;
  KL_TRY(forward_impl(h, B, T, idx, ctx, states, masks, 1, w, s));

  // F5/F6: logits over the (masked) top-layer outputs, softmax, CE, dlogits
  const bool top_masked = masks != nullptr && L > 1;
  const bf16_t* Htop = top_masked ? w.Hd[L - 1] : (const bf16_t*)w.H[L - 1] + BW;
  // (one kernel where it applies -- V = 256, width 512: the logits never reach memory --, else GEMM + softmax)
  const int mean_rows = (h->loss_rows > 0 && h->loss_rows <= B) ? h->loss_rows : B;      // (a padded batch: the real streams)
  const float inv_count = 1.0f / (h->last_only ? (float)mean_rows : (float)mean_rows * (float)T);
  int fe = KL_ERR_SHAPE;
  if (h->logits_ws && loss_acc != nullptr && w.rowstat != nullptr && V == Vp)
    fe = kl_launch_logits_ce_ws(Htop, d.E_hi, tgt, w.dlogits, w.rowstat, B, T, W, V, Vp, inv_count, h->last_only, s);
  // (width 128: ... and dH = dlogits . E in the same pass -- lstm_scan_w128.hip; the scans there take dH as f32 rows)
  bool dh_done = false;
  if (fe == KL_ERR_SHAPE && h->logits_ws && W == 128 && loss_acc != nullptr && w.rowstat != nullptr && !w.scan2_bwd) {
    fe = kl_launch_logits_ce_w128(Htop, d.E_hi, d.ET, tgt, w.dlogits, w.dH, w.rowstat, B, T, W, V, Vp, inv_count, h->last_only, s);
    dh_done = fe == 0;
  }
  if (fe == 0) {
    KL_TRY(kl_launch_rowstat_reduce(w.rowstat, BT, loss_acc, s));
  } else if (fe == KL_ERR_SHAPE) {
    KL_TRY(kl_launch_gemm_tn(Htop, d.E_hi, w.logits, nullptr, BT, V, W, W, W, V, 0, 1, 1.f, s));
    KL_TRY(kl_launch_softmax_ce(w.logits, V, BT, V, tgt, B, T, inv_count, w.dlogits, Vp, loss_acc, w.rowstat, 1, s, h->last_only));
  } else {
    return fe;
  }
  // B1: dH = dlogits . E ; dE += dlogits^T . Htop
  // (second-generation backward scan: dH travels as bf16)
  const int dh_mode = w.scan2_bwd ? 1 : 0;
  int de = KL_ERR_SHAPE;
  if (dh_mode == 1 && h->logits_ws) de = kl_launch_dh_ws(w.dlogits, d.ET, reinterpret_cast<bf16_t*>(w.dH), BT, W, Vp, s);
  if (dh_done) de = 0;
  if (de == KL_ERR_SHAPE) de = kl_launch_gemm_tn(w.dlogits, d.ET, w.dH, nullptr, BT, W, Vp, Vp, Vp, W, dh_mode, 1, 1.f, s);
  KL_TRY(de);
  if (BTp != BT) {
    KL_TRY(kl_zero_async(w.dlogitsT, (size_t)Vp * BTp * sizeof(bf16_t), s));
    KL_TRY(kl_zero_async(w.HT, (size_t)W * BTp * sizeof(bf16_t), s));
    KL_TRY(kl_zero_async(w.dZT, (size_t)4 * W * BTp * sizeof(bf16_t), s));
  }
  const long ldtf = (long)(T + 1) * B;
  const bool km_e = w.km_plan && Vp == V && kl_gemm_an_applicable(V, W, BT, Vp);
  if (!km_e) KL_TRY(kl_launch_transpose_bf16(w.dlogits, Vp, w.dlogitsT, BTp, BT, Vp, s));
  if (km_e) {
    // dE += dlogits^T . Htop with both operands as they lie in memory
    KL_TRY(kl_launch_gemm_an(w.dlogits, Htop, grads + h->off_E, V, W, BT, Vp, W, W, 0, s, 1));
  } else if (w.ht_ready) {
    const bf16_t* HtopT = top_masked ? w.HdT[L - 1] : w.HTf[L - 1] + B;
    KL_TRY(kl_launch_gemm_tn(w.dlogitsT, HtopT, grads + h->off_E, nullptr, V, W, BT, BTp, top_masked ? (long)BT : ldtf, W, 2, ksplit, 1.f, s));
  } else {
    KL_TRY(kl_launch_transpose_bf16(Htop, W, w.HT, BTp, BT, W, s));
    KL_TRY(kl_launch_gemm_tn(w.dlogitsT, w.HT, grads + h->off_E, nullptr, V, W, BTp, BTp, BTp, W, 2, ksplit, 1.f, s));
  }

  // B3: reverse recurrence -- persistent scan where the shape allows it, else the
  // launch-per-step layer wavefront
  std::vector<char> wg_done(L, 0);
  // B4/B5: weight gradients of one layer, K = B*T contractions over transposed activations
  // dz_km: the contractions over the T*B rows read dZ and the activations K-major, as the scans wrote them
  // (kl_launch_gemm_an, the hardware transpose read) -- no transposed copies at all
  auto weight_grads = [&](int l, bool dzt_ready, bool db_done, bool dz_km) -> int {
    if (!dzt_ready && !dz_km) KL_TRY(kl_launch_transpose_bf16(w.dZ[l], 4 * W, w.dZT, BTp, BT, 4 * W, s));
    // dU_l = Hprev^T . dZ   (Hprev = H blocks 0..T-1)
    // (dz_km: every product over the layer's dZ rows in as few passes over them as possible -- KlGemmSecond, gemm.hip)
    // (a pair has twice the column tiles of a single product; where the launcher does not take it -- KL_ERR_SHAPE -- the
    //  products go out one by one, as km_plan has checked they can)
    bool pair_uk = dz_km && l > 0 && h->fuse_wg && (W % 128) == 0 && kl_gemm_an_applicable(4 * W, 2 * W, BT, 4 * W);
    if (pair_uk) {
      const bool masked_in = masks != nullptr && (l - 1) > 0;
      const bf16_t* X = masked_in ? w.Hd[l - 1] : (const bf16_t*)w.H[l - 1] + BW;
      const int pe = kl_launch_gemm_an2(w.dZ[l], (const bf16_t*)w.H[l], grads + h->off_U[l], 4 * W, W, BT, 4 * W, W, 4 * W, 1,
                                        X, grads + h->off_K[l], W, W, 4 * W, 1, s, 1);
      if (pe == KL_ERR_SHAPE) pair_uk = false;
      else KL_TRY(pe);
    }
    if (pair_uk) {
      // (dU and dK done in one pass)
    } else if (dz_km) {
      KL_TRY(kl_launch_gemm_an(w.dZ[l], (const bf16_t*)w.H[l], grads + h->off_U[l], 4 * W, W, BT, 4 * W, W, 4 * W, 1, s, 1));
    } else if (w.ht_ready) {
      KL_TRY(kl_launch_gemm_tn(w.HTf[l], w.dZT, grads + h->off_U[l], nullptr, W, 4 * W, BT, ldtf, BTp, 4 * W, 2, ksplit, 1.f, s));
    } else {
      KL_TRY(kl_launch_transpose_bf16((const bf16_t*)w.H[l], W, w.HT, BTp, BT, W, s));
      KL_TRY(kl_launch_gemm_tn(w.HT, w.dZT, grads + h->off_U[l], nullptr, W, 4 * W, BTp, BTp, BTp, 4 * W, 2, ksplit, 1.f, s));
    }
    if (!db_done) KL_TRY(kl_launch_colsum_bf16(w.dZ[l], 4 * W, BT, 4 * W, grads + h->off_b[l], s));   // (the wide backward scan sums db itself)
    if (l > 0) {
      // dK_l = X^T . dZ with X = (masked) outputs of layer l-1
      const bool masked_in = masks != nullptr && (l - 1) > 0;
      const bf16_t* X = masked_in ? w.Hd[l - 1] : (const bf16_t*)w.H[l - 1] + BW;
      if (pair_uk) {
        // (done with dU above)
      } else if (dz_km) {
        KL_TRY(kl_launch_gemm_an(w.dZ[l], X, grads + h->off_K[l], 4 * W, W, BT, 4 * W, W, 4 * W, 1, s, 1));
      } else if (w.ht_ready) {
        const bf16_t* XT = masked_in ? w.HdT[l - 1] : w.HTf[l - 1] + B;
        KL_TRY(kl_launch_gemm_tn(XT, w.dZT, grads + h->off_K[l], nullptr, W, 4 * W, BT, masked_in ? (long)BT : ldtf, BTp, 4 * W, 2, ksplit, 1.f, s));
      } else {
        KL_TRY(kl_launch_transpose_bf16(X, W, w.HT, BTp, BT, W, s));
        KL_TRY(kl_launch_gemm_tn(w.HT, w.dZT, grads + h->off_K[l], nullptr, W, 4 * W, BTp, BTp, BTp, 4 * W, 2, ksplit, 1.f, s));
      }
    } else {
      // layer 0 through the look-up tables: dEK^T = dZ^T . OneHot ; dCtxK_n^T likewise
      if (kl_launch_onehot_dense(idx, B, T, Vp, 0, 1, w.OHT, BTp, s) == KL_ERR_SHAPE) {
        KL_TRY(kl_zero_async(w.OHT, (size_t)Vp * BTp * sizeof(bf16_t), s));
        KL_TRY(kl_launch_onehot_t(idx, B, T, V, 0, 1, w.OHT, BTp, s));
      }
      KL_TRY(kl_zero_async(w.dEKT, (size_t)4 * W * Vp * sizeof(float), s));
      // (the first context variable's one-hot product rides along with the characters': one pass over dZ for both)
      bool pair_ctx = dz_km && h->fuse_wg && c.n_ctx >= 1 && (Vp % 128) == 0 &&
                      kl_gemm_an_applicable(4 * W, Vp + c.ctx_vocab, BT, 4 * W);
      if (pair_ctx) {
        if (kl_launch_onehot_dense(ctx, B, T, c.ctx_vocab, 0, c.n_ctx, w.OHC[0], BTp, s) == KL_ERR_SHAPE) {
          KL_TRY(kl_zero_async(w.OHC[0], (size_t)c.ctx_vocab * BTp * sizeof(bf16_t), s));
          KL_TRY(kl_launch_onehot_t(ctx, B, T, c.ctx_vocab, 0, c.n_ctx, w.OHC[0], BTp, s));
        }
        KL_TRY(kl_zero_async(w.dCtxKT[0], (size_t)4 * W * c.ctx_vocab * sizeof(float), s));
        const int pe = kl_launch_gemm_an2(w.dZ[l], w.OHT, w.dEKT, 4 * W, Vp, BT, 4 * W, BTp, Vp, 0,
                                          w.OHC[0], w.dCtxKT[0], c.ctx_vocab, BTp, c.ctx_vocab, 0, s, 0);
        if (pe == KL_ERR_SHAPE) pair_ctx = false;      // (nothing was launched: both products follow one by one)
        else KL_TRY(pe);
      }
      if (pair_ctx) {
        // (characters and first context variable done in one pass)
      } else if (dz_km) KL_TRY(kl_launch_gemm_an(w.dZ[l], w.OHT, w.dEKT, 4 * W, Vp, BT, 4 * W, BTp, Vp, 0, s));
      else KL_TRY(kl_launch_gemm_tn(w.dZT, w.OHT, w.dEKT, nullptr, 4 * W, Vp, BTp, BTp, BTp, Vp, 2, ksplit, 1.f, s));
      KL_TRY(kl_launch_f32_to_bf16_t(w.dEKT, Vp, 4 * W, Vp, w.dEKT_bf, nullptr, Vp, 0, s));
      KL_TRY(kl_launch_f32_to_bf16_t(w.dEKT, Vp, 4 * W, Vp, w.dEK_bf, nullptr, 4 * W, 1, s));
      // dK0[:W] = E^T . dEK      (C[W][4W] = ET[W][Vp] . dEKT[4W][Vp]^T)
      KL_TRY(kl_launch_gemm_tn(d.ET, w.dEKT_bf, grads + h->off_K[0], nullptr, W, 4 * W, Vp, Vp, Vp, 4 * W, 0, 1, 1.f, s));
      // dE += dEK . K0[:W]^T     (C[V][W] = dEK[V][4W] . Kn0[W][4W]^T)
      KL_TRY(kl_launch_gemm_tn(w.dEK_bf, d.Kn[0], grads + h->off_E, nullptr, V, W, 4 * W, 4 * W, 4 * W, W, 2, 1, 1.f, s));
      for (int n = 0; n < c.n_ctx; ++n) {
        if (!(pair_ctx && n == 0)) {
          if (kl_launch_onehot_dense(ctx, B, T, c.ctx_vocab, n, c.n_ctx, w.OHC[n], BTp, s) == KL_ERR_SHAPE) {
            KL_TRY(kl_zero_async(w.OHC[n], (size_t)c.ctx_vocab * BTp * sizeof(bf16_t), s));
            KL_TRY(kl_launch_onehot_t(ctx, B, T, c.ctx_vocab, n, c.n_ctx, w.OHC[n], BTp, s));
          }
          KL_TRY(kl_zero_async(w.dCtxKT[n], (size_t)4 * W * c.ctx_vocab * sizeof(float), s));
          if (dz_km) KL_TRY(kl_launch_gemm_an(w.dZ[l], w.OHC[n], w.dCtxKT[n], 4 * W, c.ctx_vocab, BT, 4 * W, BTp, c.ctx_vocab, 0, s));
          else KL_TRY(kl_launch_gemm_tn(w.dZT, w.OHC[n], w.dCtxKT[n], nullptr, 4 * W, c.ctx_vocab, BTp, BTp, BTp,
                                        c.ctx_vocab, 2, ksplit, 1.f, s));
        }
        const size_t krow = (size_t)(W + n * c.ctx_dim) * 4 * W;
        KL_TRY(kl_launch_ctx_grads(P + h->off_Ctx[n], P + h->off_K[0] + krow, 4 * W, c.ctx_vocab, c.ctx_dim,
                                   w.dCtxKT[n], c.ctx_vocab, 4 * W, grads + h->off_K[0] + krow, 4 * W,
                                   grads + h->off_Ctx[n], s));
      }
    }
      return 0;
  };
  bool bscanned = false;
  // Many row blocks (B >= 256 at cfg2): the fused two-layer scan is bound by every
  // workgroup re-reading its 16 x 4W dZ tile for BOTH contractions.  Run the layers one
  // after the other instead -- each scan then only carries the recurrent contraction and
  // twice the row groups fit -- and take the from-above term dZ_{l+1} . K_{l+1}^T for all
  // steps at once from the big GEMM.
  const int n_rb_all = (B + 15) / 16, nug = W / 16;
  const bool thin_fits = (n_rb_all + 512 / nug - 1) / (512 / nug) <= (W == 1024 ? 8 : 4) &&
                         (long)T * B * 4 * W * 2 <= 0xfffffff0L;      // (the scans address dZ with unsigned 32-bit offsets)
  const bool wide_fits = h->wide_bwd && kl_scan_bwd_wide_applicable(B, T, W) && BTp == BT && (B & 7) == 0;
  // (width 1024 has no fused scan at all: always layer by layer)
  // (... and deeper than four layers: the fused scan's limit)
  // (width 128: the one-layer scans of lstm_scan_w128.hip, a workgroup per 16-row block with all units)
  const bool seq128 = h->scan_enabled && h->seq_bwd && h->w128 && B >= h->w128_min && kl_scan_w128_applicable(B, T, W) && !w.scan2_bwd;
  const bool sequential = seq128 || (h->scan_enabled && h->seq_bwd &&
                          ((L > 1 && L <= KL_SCAN_MAXL && (W == 512 || W == 256 || W == 128) && n_rb_all > 512 / (L * nug)) ||
                           W == 1024 || (L > KL_SCAN_MAXL && (W == 512 || W == 256 || W == 128 || W == 64))) &&
                          (thin_fits || wide_fits));
  // Width 128, all layers' workgroups fitting the CUs at once: ONE launch, a layer following the one above it a step behind (it
  // polls the dZ rows that one publishes, which therefore start out as sentinels), then every layer's weight gradients.
  if (seq128 && h->w128_fuse && h->w128_multi && kl_scan_w128_multi_fits(B, L)) {
    KlScanBwd a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.T = T; a.W = W; a.L = L;
    for (int l = 0; l < L; ++l) {
      a.Un[l] = d.Un[l];
      a.Kn[l] = d.Kn[l];
      a.G[l] = w.G[l];
      a.C[l] = w.C[l];
      a.dZ[l] = w.dZ[l];
      a.mask[l] = (masks != nullptr && l > 0) ? masks + (size_t)l * BW : nullptr;
      a.db_l[l] = grads + h->off_b[l];
    }
    a.dH = w.dH;
    a.status = w.scan_status + 1;
    const bool roll = h->sentinel_roll && T >= 3;      // (rolling sentinels: only the last two steps start out armed)
    a.sentinel = roll ? 2 : 1;
    for (int l = 1; l < L; ++l)
      KL_TRY(kl_fill_u32_async(w.dZ[l] + (roll ? (size_t)(T - 2) * BW * 4 : 0), (size_t)(roll ? 2 : T) * BW * 4 * sizeof(bf16_t), 0xFFFFFFFFu, s));
    h->trace_begin(1, s);
    const int e = kl_launch_scan_bwd_w128_multi(a, s);
    if (e != KL_ERR_SHAPE) {
      KL_TRY(e);
      h->trace_persistent[1] = true;
      h->trace_name[1] = "lstm_scan_bwd_w128_multi_kernel";
      h->trace_flops[1] = (double)(2 * L - 1) * B * T * (2.0 * W * 4.0 * W);      // (recurrent contractions + the from-above ones)
      h->trace_end(1, s);
      for (int l = L - 1; l >= 0; --l) {
        KL_TRY(weight_grads(l, false, true, w.km_plan));
        wg_done[l] = 1;
      }
      bscanned = true;
    }
  }
  if (!bscanned && (sequential || w.scan2_bwd)) {
    for (int l = L - 1; l >= 0; --l) {
      // (width 128: the scan of lstm_scan_w128.hip contracts the layer above's dZ rows itself)
      const bool fuse_dx = seq128 && h->w128_fuse && l < L - 1;
      if (l < L - 1 && !fuse_dx)   // dX_l = dZ_{l+1} . K_{l+1}^T  -> w.dH (free once the layer above has been scanned)
        KL_TRY(kl_launch_gemm_tn(w.dZ[l + 1], d.Kn[l + 1], w.dH, nullptr, BT, W, 4 * W, 4 * W, 4 * W, W, dh_mode, 1, 1.f, s));
      KlScanBwd a;
      memset(&a, 0, sizeof(a));
      a.B = B; a.T = T; a.W = W; a.L = 1;
      a.Un[0] = d.Un[l];
      a.G[0] = w.G[l];
      a.C[0] = w.C[l];
      a.dZ[0] = w.dZ[l];
      a.mask[0] = (masks != nullptr && l > 0) ? masks + (size_t)l * BW : nullptr;
      a.dH = w.dH;
      a.dHb = reinterpret_cast<const bf16_t*>(w.dH);
      a.Cb = (w.scan2_bwd && h->scan2_bf16) ? w.Cb[l] : nullptr;
      a.counters = w.scan_cnt;
      a.status = w.scan_status + 1;
      // (sentinels pay with several row blocks per workgroup, where the next tile is prefetched; with one
      // block the cheap counter poll beats re-fetching 64 KiB tiles; XCD-local publishes measured slower here)
      a.sentinel = (w.scan2_bwd || (h->sentinel_bwd && wide_fits && (kl_scan_wide_blocks_per_wg(B, W) > 1 || h->sentinel_bwd_all))) ? 1 : 0;
      // (measured at B = 1024 .. 3072: the request behind the MFMA phase is the best or tied everywhere; at the top of the block
      //  one tile in ten is requested before it is published, and the re-fetch costs more than the earlier request saves)
      a.pf_mode = h->scan2_pfb >= 0 ? h->scan2_pfb : 1;
      a.xcc_slots = (a.sentinel && h->xcd_local_bwd) ? w.scan_status + 4 : nullptr;
      a.gen = (unsigned)(1 + L + l);
      const int np_b = w.scan2_bwd ? kl_scan_wide2_phases(B, T, W, 16, 6) : 0;
      const bool by_flags = w.scan2_bwd && h->scan2_flags && h->flags_zeroed && np_b >= 3;
      // (from five blocks per step: eight waves, the tile through registers two blocks ahead -- lstm_scan_bwd_regtile_kernel)
      const bool rt = by_flags && h->regtile && h->scan2_bf16 && np_b >= kl_scan_bwd_regtile_min_np();
      if (by_flags) {
        // hand-off by flags (lstm_scan_bwd_wide2_kernel): nothing to arm, the epoch moves on
        a.flags = d.scan_flags;
        a.epoch = d.scan_flags + KL_SCAN_FLAGS;
        // (flags tell before the request whether a tile is there, so it can be asked for TWO blocks ahead, behind the epilogue,
        //  without the re-fetches that cost the sentinel form -- once five or more blocks lie between a publish and its use)
        if (h->scan2_pfb < 0) a.pf_mode = kl_scan_wide2_phases(B, T, W, 16, 6) >= 5 ? 2 : 1;
        else if (a.pf_mode == 0) a.pf_mode = 1;
        KL_TRY(kl_launch_scan_epoch(d.scan_flags, KL_SCAN_FLAGS, d.scan_flags + KL_SCAN_FLAGS, (unsigned)T + 2u, s));
      } else if (a.sentinel && h->sentinel_roll && T >= 3) {
        // rolling sentinels: the scan re-arms step t - 2 while it publishes step t; only the first two start armed
        a.sentinel = 2;
        KL_TRY(kl_fill_u32_async(w.dZ[l] + (size_t)(T - 2) * BW * 4, (size_t)2 * BW * 4 * sizeof(bf16_t), 0xFFFFFFFFu, s));
      } else if (a.sentinel)   // hand-off by data: the steps the scan is going to publish start out as sentinels
        KL_TRY(kl_fill_u32_async(w.dZ[l], (size_t)T * BW * 4 * sizeof(bf16_t), 0xFFFFFFFFu, s));
      else
        KL_TRY(kl_zero_coherent_async(w.scan_cnt, (size_t)n_rb_all * T, s));
      h->trace_begin(1, s);
      // wide (64-unit) workgroups share the dZ tile through LDS; the weight-gradient GEMMs read dZ K-major
      // as it is (dz_km), else the scan also writes dZ^T
      a.dZT = (!w.km_plan && BTp == BT && (B & 7) == 0) ? w.dZT : nullptr;
      a.ldt = BTp;
      a.db = grads + h->off_b[l];
      if (rt) a.xcc_slots = h->rt_local ? w.scan_status + 4 : nullptr;
      int e = w.scan2_bwd ? (rt ? kl_launch_scan_bwd_regtile(a, s) : kl_launch_scan_bwd_wide2(a, s)) : (h->wide_bwd ? kl_launch_scan_bwd_wide(a, s) : KL_ERR_SHAPE);
      bool w32 = false, w128k = false;
      if (e == KL_ERR_SHAPE && seq128) {
        KlScanBwd a1 = a;
        a1.dZT = nullptr;
        a1.sentinel = 0;
        if (fuse_dx) { a1.Kn[1] = d.Kn[l + 1]; a1.dZ[1] = w.dZ[l + 1]; }
        e = kl_launch_scan_bwd_w128(a1, s);
        if (e == KL_ERR_SHAPE && fuse_dx) return KL_ERR_SHAPE;      // (the dX product was skipped for it)
        w128k = e == 0;
        if (w128k) a.dZT = nullptr;
      }
      if (e == KL_ERR_SHAPE && !w.scan2_bwd && h->w32 && n_rb_all >= h->w32_min_rb && kl_scan_w32_applicable(B, T, W)) {
        // width 1024: eight-wave workgroups of 32 units (lstm_scan_w32.hip); every step starts out as sentinels
        if (a.sentinel != 1) KL_TRY(kl_fill_u32_async(w.dZ[l], (size_t)T * BW * 4 * sizeof(bf16_t), 0xFFFFFFFFu, s));
        a.sentinel = 1;
        a.dZT = nullptr;
        a.xcc_slots = h->w32_local ? w.scan_status + 4 : nullptr;
        a.gen = (unsigned)(1 + L + l);
        e = kl_launch_scan_bwd_w32(a, s);
        w32 = e == 0;
      }
      const bool wide = e == 0;
      if (e == KL_ERR_SHAPE && w.scan2_bwd) return KL_ERR_SHAPE;      // (the forward scans wrote gate-interleaved G: planned together, plan_scan2)
      if (e == KL_ERR_SHAPE) {
        a.dZT = nullptr;
        a.db = nullptr;
        if (a.sentinel) KL_TRY(kl_zero_coherent_async(w.scan_cnt, (size_t)n_rb_all * T, s));   // (the thin scan counts)
        e = kl_launch_scan_bwd(a, s);
      }
      if (e != 0) return e;
      {
        h->trace_persistent[1] = true;
        h->trace_name[1] = w.scan2_bwd ? (rt ? "lstm_scan_bwd_regtile_kernel" : "lstm_scan_bwd_wide2_kernel")
                                       : (w128k ? "lstm_scan_bwd_w128_kernel" : w32 ? "lstm_scan_bwd_w32_kernel" : (wide ? "lstm_scan_bwd_wide_kernel" : "lstm_scan_bwd_kernel"));
        h->trace_flops[1] = (double)B * T * (2.0 * W * 4.0 * W);   // one layer's recurrent contraction
        h->trace_end(1, s);
      }
      // dZ^T lives in ONE buffer: this layer's weight gradients before the next layer's scan
      KL_TRY(weight_grads(l, wide && a.dZT != nullptr, wide, w.km_plan));
      wg_done[l] = 1;
    }
    bscanned = true;
  }
  if (!bscanned && h->scan_enabled && L <= KL_SCAN_MAXL) {
    KlScanBwd a;
    memset(&a, 0, sizeof(a));
    a.B = B; a.T = T; a.W = W; a.L = L;
    for (int l = 0; l < L; ++l) {
      a.Un[l] = d.Un[l];
      a.Kn[l] = d.Kn[l];
      a.G[l] = w.G[l];
      a.C[l] = w.C[l];
      a.dZ[l] = w.dZ[l];
      a.mask[l] = (masks != nullptr && l > 0) ? masks + (size_t)l * BW : nullptr;
    }
    a.dH = w.dH;
    a.counters = w.scan_cnt;
    a.status = w.scan_status + 1;
    KL_TRY(kl_zero_coherent_async(w.scan_cnt, (size_t)L * ((B + 15) / 16) * T, s));
    h->trace_begin(1, s);
    const int e = kl_launch_scan_bwd(a, s);
    if (e == 0) {
      bscanned = true;
      h->trace_persistent[1] = true;
      h->trace_name[1] = "lstm_scan_bwd_kernel";
      h->trace_flops[1] = (double)B * T * (2.0 * (2.0 * L - 1.0) * W * 4.0 * W);
      h->trace_end(1, s);
    } else if (e != KL_ERR_SHAPE) {
      return e;
    }
  }
  for (int dgl = 0; !bscanned && dgl < T + L - 1; ++dgl) {
    KlBwdStep steps[16];     // one per layer (config_ok: depth <= 16); launched in packs of 4 below
    int ns = 0;
    for (int j = 0; j < L; ++j) {
      const int l = L - 1 - j;
      const int t = T - 1 - (dgl - j);
      if (t < 0 || t >= T) continue;
      KlBwdStep& S = steps[ns++];
      memset(&S, 0, sizeof(S));
      S.n_rows = B; S.W = W;
      int p = 0;
      if (l < L - 1) {   // gradient from the layer above: dZ_{l+1}[t] . K_{l+1}^T
        KlOperand& o = S.op[p++];
        o.A = w.dZ[l + 1] + (size_t)t * B * 4 * W; o.lda = 4 * W; o.a_is_f32 = 0;
        o.WT_hi = d.Kn[l + 1]; o.ldw = 4 * W; o.K = 4 * W;
        if (masks != nullptr && l > 0) { S.op0_mask = masks + (size_t)l * BW; S.op0_mask_ld = W; }
      } else {
        S.dh_in = w.dH + (size_t)t * BW; S.dh_in_ld = W;
        if (top_masked) { S.dh_mask = masks + (size_t)l * BW; S.dh_mask_ld = W; }
      }
      if (t < T - 1) {   // recurrent: dZ_l[t+1] . U_l^T
        KlOperand& o = S.op[p++];
        o.A = w.dZ[l] + (size_t)(t + 1) * B * 4 * W; o.lda = 4 * W; o.a_is_f32 = 0;
        o.WT_hi = d.Un[l]; o.ldw = 4 * W; o.K = 4 * W;
        // an op0 mask must only scale the from-above contribution: keep that one first
      }
      S.n_ops = p;
      if (S.op0_mask && !(l < L - 1)) S.op0_mask = nullptr;
      S.gates = w.G[l] + (size_t)t * B * 4 * W; S.gates_ld = 4 * W;
      S.c = w.C[l] + (size_t)(t + 1) * BW; S.c_ld = W;
      S.c_prev = w.C[l] + (size_t)t * BW; S.c_prev_ld = W;
      float* dc_rd = (t & 1) ? w.dc1[l] : w.dc0[l];
      float* dc_wr = (t & 1) ? w.dc0[l] : w.dc1[l];
      if (t < T - 1) { S.dc_in = dc_rd; S.dc_in_ld = W; }
      S.dc_out = dc_wr; S.dc_out_ld = W;
      S.dz_out = w.dZ[l] + (size_t)t * B * 4 * W; S.dz_ld = 4 * W;
    }
    for (int i = 0; i < ns; i += 4) {
      const bool steady = (ns == L) && dgl >= L && dgl + 8 < T;
      if (steady && dgl % 8 == 0) h->trace_begin(1, s);
      KL_TRY(kl_launch_bwd_steps(steps + i, ns - i < 4 ? ns - i : 4, s));
      if (steady && dgl % 8 == 7) h->trace_end(1, s);
    }
  }

  for (int l = L - 1; l >= 0; --l)
    if (!wg_done[l]) KL_TRY(weight_grads(l, false, false, w.km_plan));

  // F7: embedding regularisers (training phase only)
  std::vector<const float*> ctabs(c.n_ctx);
  std::vector<float*> gctabs(c.n_ctx);
  for (int n = 0; n < c.n_ctx; ++n) { ctabs[n] = P + h->off_Ctx[n]; gctabs[n] = grads + h->off_Ctx[n]; }
  KL_TRY(kl_launch_regulariser_grads(P + h->off_E, V, W, ctabs.data(), c.n_ctx, c.ctx_vocab, c.ctx_dim,
                                     grads + h->off_E, gctabs.data(), loss_acc, w.reg_scratch, s));
  // a timed-out hand-off in a persistent scan surfaces as loss_acc[3] != 0
  hipLaunchKernelGGL(scan_status_kernel, dim3(1), dim3(64), 0, s, w.scan_status, loss_acc);
  return hip_ok(hipGetLastError());
}

int kl_adam_step_scaled(kl_handle* h, const float* grads, float grad_scale, float* m, float* v, int t, float lr, float b1,
                        float b2, float eps, float clip, void* stream) {
  if (!h || !grads || !m || !v || t < 1) return KL_ERR_ARG;
  if (!h->params) return KL_ERR_STATE;
  hipStream_t s = (hipStream_t)stream;
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t));
  KL_TRY(kl_launch_adam(h->params, grads, m, v, h->n_params, (float)lr_t, b1, b2, eps, clip, grad_scale, s));
  return prepare_impl(h, h->precision ? h->precision : KL_PREC_BF16, s);
}

int kl_adam_step(kl_handle* h, const float* grads, float* m, float* v, int t, float lr, float b1, float b2,
                 float eps, float clip, void* stream) {
  return kl_adam_step_scaled(h, grads, 1.0f, m, v, t, lr, b1, b2, eps, clip, stream);
}

size_t kl_step_workspace_bytes(const kl_handle* h, int n) {
  if (!h || n < 1) return 0;
  Carver cv(nullptr);
  cv.take<float>((size_t)n * 4 * h->cfg.width);          // P rows when n_ctx != 1
  cv.take<float>((size_t)n * 4 * h->cfg.width);          // z of the big-n path
  cv.take<bf16_t>((size_t)n * 3 * 2 * h->cfg.width);     // [hi | lo | hi] activation rows
  for (int l = 0; l < h->cfg.depth; ++l)                 // ... of every layer at once (fused path)
    cv.take<bf16_t>((size_t)n * 3 * (l == 0 ? 1 : 2) * h->cfg.width);
  return align_up(cv.off, 256);
}

int kl_step_batch(kl_handle* h, int n, const int32_t* idx, const int32_t* ctx, float* pool, const int32_t* slot_in,
                  const int32_t* slot_out, float* probs, void* ws, size_t ws_bytes, void* stream) {
  if (!h || !idx || !pool || !slot_in || !slot_out || !probs || n < 1) return KL_ERR_ARG;
  if (h->cfg.n_ctx > 0 && !ctx) return KL_ERR_ARG;
  if (!h->precision) return KL_ERR_STATE;
  hipStream_t s = (hipStream_t)stream;
  const kl_config& c = h->cfg;
  const int W = c.width, L = c.depth, V = c.voc_size;
  const long slot_ld = (long)2 * L * W;
  const float* P = h->params;
  Derived& d = h->d;
  const int split = h->precision;
  float* prow = nullptr;
  if (c.n_ctx != 1) {
    if (!ws || ws_bytes < kl_step_workspace_bytes(h, n)) return KL_ERR_WORKSPACE;
    prow = reinterpret_cast<float*>(ws);
    std::vector<const float*> ctxk(c.n_ctx);
    for (int k = 0; k < c.n_ctx; ++k) ctxk[k] = d.CtxK[k];
    // rows are hypotheses: treat as B = n streams, T = 1
    KL_TRY(kl_launch_p1_gather(d.EK, ctxk.data(), c.n_ctx, P + h->off_b[0], idx, ctx, n, 1, 4 * W, prow, s));
  }
  // the output layer of every path below: logits over the tied embedding from the top layer's new h, softmax in place
  auto output_layer = [&]() -> int {
    if (h->out_fused && n >= h->out_fused_min && d.EF) {
      if (!h->inc_ready) KL_TRY(prepare_incremental(h, s));
      const int e = kl_launch_out_softmax(pool, slot_ld, slot_out, 2 * (L - 1) * W, d.EF, split, n, W, V, probs, V, s);
      if (e != KL_ERR_SHAPE) return e;
    }
    KlOperand op;
    memset(&op, 0, sizeof(op));
    op.A = pool + (size_t)2 * (L - 1) * W; op.lda = slot_ld; op.row_index = slot_out; op.a_is_f32 = 1;
    op.WT_hi = d.E_hi; op.WT_lo = split == 3 ? d.E_lo : nullptr; op.ldw = W; op.K = W;
    KL_TRY(kl_launch_thin_gemm(&op, n, V, probs, V, nullptr, split, s));
    return kl_launch_softmax_ce(probs, V, n, V, nullptr, n, 1, 1.f, nullptr, 0, nullptr, nullptr, 0, s);
  };
  auto cell_args = [&](int l) {
    KlIncCellArgs a;
    memset(&a, 0, sizeof(a));
    a.n = n; a.W = W; a.split = split;
    a.pool = pool; a.slot_ld = slot_ld; a.slot_in = slot_in; a.slot_out = slot_out;
    a.h_off = 2 * l * W; a.c_off = (2 * l + 1) * W; a.x_off = l > 0 ? 2 * (l - 1) * W : -1;
    a.UF = d.UF[l]; a.KF = d.KF[l];
    if (l == 0) {
      if (prow) { a.T1 = prow; }
      else { a.T1 = d.EK; a.i1 = idx; a.T2 = d.CtxK[0]; a.i2 = ctx; a.bias = P + h->off_b[0]; }
    } else {
      a.bias = P + h->off_b[l];
    }
    return a;
  };
  // 256 hypotheses or more, widths of 256, 384, 512, ...: one launch per layer, tiles of 64 hypotheses x 32 units with the
  // state rows read through the pool slots and the weights from the hi / lo arrays as they are (step_tile.hip)
  // (width 1024 already from 129 hypotheses: inc_cell_kernel cannot hold two row tiles of K = 2048 in LDS there, and with one
  //  its W / 16 x n / 16 workgroups read every weight 16 times at 250 hypotheses -- 90 us per step against 51 on 64-row tiles)
  if ((n >= KL_BIG_STEP_N || (W >= 1024 && n > 128)) && h->inc_tile && V < 1024 && d.EF) {
    if (!h->inc_ready) KL_TRY(prepare_incremental(h, s));
    int e = 0;
    for (int l = 0; l < L && e == 0; ++l) {
      e = kl_launch_inc_tile(cell_args(l), h->tile_var, s, h->tile_rows);
      if (e == KL_ERR_SHAPE && l > 0) return e;      // (layer 0 decides for all: the shapes are the same)
    }
    if (e == 0) return output_layer();
    if (e != KL_ERR_SHAPE) return e;
  }
  // 16 hypotheses and more (the reference's callers feed at most 128 / 256 rows; also whatever the tile kernel does not take:
  // widths 64 and 128): coalesced state rows through LDS, 16-unit workgroups (step_small.hip); KL_ERR_SHAPE, or fewer rows:
  // the gather + GEMM path / the launch-per-layer kernels below
  if (h->inc_small && n >= h->inc_small_min && (n < KL_BIG_STEP_N || V < 1024)) {
    int e = d.EF ? 0 : KL_ERR_SHAPE;
    if (e == 0 && !h->inc_ready) KL_TRY(prepare_incremental(h, s));
    for (int l = 0; l < L && e == 0; ++l) {
      e = kl_launch_inc_cell(cell_args(l), s);
      if (e == KL_ERR_SHAPE && l > 0) return e;      // (layer 0 decides for all: the shapes are the same)
    }
    if (e == 0) return output_layer();
    if (e != KL_ERR_SHAPE) return e;
  }
  if (n >= KL_BIG_STEP_N && ws && ws_bytes >= kl_step_workspace_bytes(h, n)) {
    if (!h->big_ready) KL_TRY(prepare_big_step(h, s));
    // big-tile path: gather+split -> one bf16 GEMM over the 3x contraction -> gates
    Carver cv(ws);
    cv.take<float>((size_t)n * 4 * W);
    float* z = cv.take<float>((size_t)n * 4 * W);
    bf16_t* A3 = cv.take<bf16_t>((size_t)n * 3 * 2 * W);
    const int nb = split == 3 ? 3 : 1;
    // fused form: one gather of all layers' recurrent halves, then per layer ONE GEMM whose
    // epilogue is the cell (gates, c', h', and h' as the next layer's input rows)
    bool fusable = h->fused_step && (W & 31) == 0 && L <= KL_SCAN_MAXL && V < 1024;
    for (int l = 0; l < L && fusable; ++l) fusable = d.WTperm[l] != nullptr && ((nb * (l == 0 ? W : 2 * W)) % 64) == 0;
    if (fusable) {
      KlGatherRec g;
      memset(&g, 0, sizeof(g));
      g.pool = pool; g.slot_ld = slot_ld; g.slot_in = slot_in; g.n = n; g.W = W; g.nb = nb;
      for (int l = 0; l < L; ++l) g.out[l] = cv.take<bf16_t>((size_t)n * 3 * (l == 0 ? 1 : 2) * W);
      KL_TRY(kl_launch_gather_recurrent(g, L, s));
      for (int l = 0; l < L; ++l) {
        const int Kl = l == 0 ? W : 2 * W;
        const bool tab = l == 0 && !prow;
        KlGateEpi e;
        memset(&e, 0, sizeof(e));
        if (l == 0) { e.T1 = prow ? prow : d.EK; e.i1 = tab ? idx : nullptr; e.T2 = tab ? d.CtxK[0] : nullptr; e.i2 = tab ? ctx : nullptr; }
        e.bias = (l > 0 || tab) ? P + h->off_b[l] : nullptr;
        e.c_prev = pool + (size_t)(2 * l + 1) * W; e.c_ld = slot_ld; e.slot_in = slot_in;
        e.c_out = pool + (size_t)(2 * l + 1) * W; e.h_out = pool + (size_t)2 * l * W; e.out_ld = slot_ld; e.slot_out = slot_out;
        if (l + 1 < L) { e.xn = g.out[l + 1]; e.ldn = 3L * 2 * W; e.kn = 2 * W; e.nbn = nb; }
        e.W = W;
        KL_TRY(kl_launch_gemm_gates(g.out[l], d.WTperm[l], n, W, nb * Kl, 3L * Kl, &e, s));
      }
      KlOperand op;
      memset(&op, 0, sizeof(op));
      op.A = pool + (size_t)2 * (L - 1) * W; op.lda = slot_ld; op.row_index = slot_out; op.a_is_f32 = 1;
      op.WT_hi = d.E_hi; op.WT_lo = split == 3 ? d.E_lo : nullptr; op.ldw = W; op.K = W;
      KL_TRY(kl_launch_thin_gemm(&op, n, V, probs, V, nullptr, split, s));
      KL_TRY(kl_launch_softmax_ce(probs, V, n, V, nullptr, n, 1, 1.f, nullptr, 0, nullptr, nullptr, 0, s));
      return 0;
    }
    for (int l = 0; l < L; ++l) {
      const int Kl = l == 0 ? W : 2 * W;
      if (l == 0) {
        KL_TRY(kl_launch_split_gather(pool, slot_ld, slot_in, W, nullptr, 0, nullptr, 0, n, nb, A3, 3L * Kl, s));
      } else {
        KL_TRY(kl_launch_split_gather(pool + (size_t)2 * (l - 1) * W, slot_ld, slot_out, W, pool + (size_t)2 * l * W, slot_ld,
                                      slot_in, W, n, nb, A3, 3L * Kl, s));
      }
      KL_TRY(kl_launch_gemm_tn(A3, d.WTcat[l], z, nullptr, n, 4 * W, nb * Kl, 3L * Kl, 3L * Kl, 4 * W, 0, 1, 1.f, s));
      const bool tab = l == 0 && !prow;
      KL_TRY(kl_launch_gates_rows(z, 4 * W, n, W, l == 0 ? (prow ? prow : d.EK) : nullptr, tab ? idx : nullptr,
                                  tab ? d.CtxK[0] : nullptr, tab ? ctx : nullptr,
                                  (l > 0 || tab) ? P + h->off_b[l] : nullptr, pool + (size_t)(2 * l + 1) * W, slot_ld, slot_in,
                                  pool + (size_t)(2 * l + 1) * W, pool + (size_t)2 * l * W, slot_ld, slot_out, s));
    }
    if (V >= 1024) {   // wide vocabulary: big tiles pay off for the output projection too
      KL_TRY(kl_launch_split_gather(pool + (size_t)2 * (L - 1) * W, slot_ld, slot_out, W, nullptr, 0, nullptr, 0, n, nb, A3, 3L * W, s));
      KL_TRY(kl_launch_gemm_tn(A3, d.Ecat, probs, nullptr, n, V, nb * W, 3L * W, 3L * W, V, 0, 1, 1.f, s));
    } else {           // V x W is small: the thin kernel spreads it over n/32 x V/16 workgroups
      KlOperand op;
      memset(&op, 0, sizeof(op));
      op.A = pool + (size_t)2 * (L - 1) * W; op.lda = slot_ld; op.row_index = slot_out; op.a_is_f32 = 1;
      op.WT_hi = d.E_hi; op.WT_lo = split == 3 ? d.E_lo : nullptr; op.ldw = W; op.K = W;
      KL_TRY(kl_launch_thin_gemm(&op, n, V, probs, V, nullptr, split, s));
    }
    KL_TRY(kl_launch_softmax_ce(probs, V, n, V, nullptr, n, 1, 1.f, nullptr, 0, nullptr, nullptr, 0, s));
    return 0;
  }
  for (int l = 0; l < L; ++l) {
    KlFwdStep S;
    memset(&S, 0, sizeof(S));
    S.n_rows = n; S.W = W; S.split = split;
    int p = 0;
    if (l == 0) {
      if (prow) {
        S.T1 = prow; S.t1_ld = 4 * W;
      } else {
        S.T1 = d.EK; S.i1 = idx; S.t1_ld = 4 * W;
        S.T2 = d.CtxK[0]; S.i2 = ctx; S.t2_ld = 4 * W;
        S.bias = P + h->off_b[0];
      }
    } else {
      KlOperand& o = S.op[p++];
      o.A = pool + (size_t)2 * (l - 1) * W; o.lda = slot_ld; o.row_index = slot_out; o.a_is_f32 = 1;
      o.WT_hi = d.KT_hi[l]; o.WT_lo = split == 3 ? d.KT_lo[l] : nullptr; o.ldw = W; o.K = W;
      S.bias = P + h->off_b[l];
    }
    {
      KlOperand& o = S.op[p++];
      o.A = pool + (size_t)2 * l * W; o.lda = slot_ld; o.row_index = slot_in; o.a_is_f32 = 1;
      o.WT_hi = d.UT_hi[l]; o.WT_lo = split == 3 ? d.UT_lo[l] : nullptr; o.ldw = W; o.K = W;
    }
    S.n_ops = p;
    S.c_prev = pool + (size_t)(2 * l + 1) * W; S.c_prev_ld = slot_ld; S.c_prev_index = slot_in;
    S.out_index = slot_out;
    S.c_out = pool + (size_t)(2 * l + 1) * W; S.c_out_ld = slot_ld;
    S.h_out_f32 = pool + (size_t)2 * l * W; S.h_out_f32_ld = slot_ld;
    KL_TRY(kl_launch_fwd_steps(&S, 1, s));
  }
  KlOperand op;
  memset(&op, 0, sizeof(op));
  op.A = pool + (size_t)2 * (L - 1) * W; op.lda = slot_ld; op.row_index = slot_out; op.a_is_f32 = 1;
  op.WT_hi = d.E_hi; op.WT_lo = split == 3 ? d.E_lo : nullptr; op.ldw = W; op.K = W;
  KL_TRY(kl_launch_thin_gemm(&op, n, V, probs, V, nullptr, split, s));
  KL_TRY(kl_launch_softmax_ce(probs, V, n, V, nullptr, n, 1, 1.f, nullptr, 0, nullptr, nullptr, 0, s));
  return 0;
}

// ---- the incremental step as a beam search issues it: one step per character, the GPU idle in between ----------------
// (rating.py:809-826 rate_best, :689-691 generate: predict -> look at the probabilities -> decide -> predict ...)
void* kl_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) return nullptr;
  memset(p, 0, bytes);
  return p;
}

void kl_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

size_t kl_step_host_workspace_bytes(const kl_handle* h, int n) {
  if (!h || n < 1) return 0;
  Carver cv(nullptr);
  cv.take<unsigned>(64);                                             // the finish launch's ticket counter (zero between steps)
  cv.take<float>((size_t)n * h->cfg.voc_size);                      // logits / probabilities
  cv.take<int32_t>((size_t)n * (4 + (h->cfg.n_ctx > 0 ? h->cfg.n_ctx : 1)));      // indices on the device (n > 256 and fall-backs)
  return align_up(cv.off, 256) + kl_step_workspace_bytes(h, n);
}

int kl_step_batch_host(kl_handle* h, int n, const int32_t* idx, const int32_t* ctx, const int32_t* slot_in,
                       const int32_t* slot_out, const int32_t* target, float* pool, int head_k, float* probs_host,
                       float* heads_host, uint32_t* done_host, uint32_t ticket, void* ws, size_t ws_bytes, void* stream) {
  if (!h || !idx || !pool || !slot_in || !slot_out || !probs_host || !done_host || n < 1) return KL_ERR_ARG;
  if (h->cfg.n_ctx > 0 && !ctx) return KL_ERR_ARG;
  if (head_k < 0 || head_k > 2 * h->cfg.depth || (head_k > 0 && !heads_host)) return KL_ERR_ARG;
  if (!h->precision) return KL_ERR_STATE;
  if (!ws || ws_bytes < kl_step_host_workspace_bytes(h, n)) return KL_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const kl_config& c = h->cfg;
  const int W = c.width, L = c.depth, V = c.voc_size, C = c.n_ctx;
  const long slot_ld = (long)2 * L * W;
  Derived& d = h->d;
  Carver cv(ws);
  unsigned* counter = cv.take<unsigned>(64);
  float* logits = cv.take<float>((size_t)n * V);
  int32_t* dev_idx = cv.take<int32_t>((size_t)n * (4 + (C > 0 ? C : 1)));
  unsigned char* rest = reinterpret_cast<unsigned char*>(ws) + align_up(cv.off, 256);
  const size_t rest_bytes = ws_bytes - align_up(cv.off, 256);
  if (!h->host_step_ready) {      // (the counter of a fresh workspace; afterwards every finish launch leaves it zero)
    KL_TRY(kl_zero_async(counter, 64 * sizeof(unsigned), s));
    h->host_step_ready = ws;
  } else if (h->host_step_ready != ws) {
    KL_TRY(kl_zero_async(counter, 64 * sizeof(unsigned), s));
    h->host_step_ready = ws;
  }
  KlStepFinish f;
  memset(&f, 0, sizeof(f));
  f.n = n; f.V = V; f.W = W; f.logits = logits; f.ld = V;
  f.by_target = target != nullptr; f.head_k = head_k;
  f.pool = pool; f.slot_ld = slot_ld;
  f.probs_host = probs_host; f.heads_host = heads_host; f.done_host = done_host; f.ticket = ticket; f.counter = counter;
  // ---- up to 256 hypotheses on the 16-unit cell kernels: every index travels in the kernel arguments
  bool in_range = n <= KL_HOST_STEP_MAX && V <= 65535 && C == 1 && h->inc_small && d.EF && h->host_kernarg;
  for (int i = 0; i < n && in_range; ++i)
    in_range = idx[i] >= 0 && idx[i] < V && ctx[i] >= 0 && ctx[i] < c.ctx_vocab && (!target || (target[i] >= 0 && target[i] < V));
  if (in_range && ((W & 255) == 0 || W == 64 || W == 128) && !(W >= 1024 && n > 128)) {
    static thread_local KlHostIdx hx;
    static thread_local KlHostTargets tx;
    for (int i = 0; i < n; ++i) {
      hx.slot_in[i] = slot_in[i]; hx.slot_out[i] = slot_out[i];
      hx.idx[i] = (unsigned short)idx[i]; hx.ctx[i] = (unsigned short)ctx[i];
      if (target) tx.t[i] = (unsigned short)target[i];
    }
    if (!h->inc_ready) KL_TRY(prepare_incremental(h, s));
    const int split = h->precision;
    const float* P = h->params;
    int e = 0;
    for (int l = 0; l < L && e == 0; ++l) {
      KlIncCellArgs a;
      memset(&a, 0, sizeof(a));
      a.n = n; a.W = W; a.split = split;
      a.pool = pool; a.slot_ld = slot_ld;
      a.h_off = 2 * l * W; a.c_off = (2 * l + 1) * W; a.x_off = l > 0 ? 2 * (l - 1) * W : -1;
      a.UF = d.UF[l]; a.KF = d.KF[l];
      a.bias = P + h->off_b[l];
      if (l == 0) {      // (i1 / i2 non-null = "table rows by index"; the values are hx's)
        a.T1 = d.EK; a.i1 = dev_idx; a.T2 = d.CtxK[0]; a.i2 = dev_idx;
      }
      e = kl_launch_inc_cell(a, s, &hx, l == 0 ? dev_idx : nullptr);
      if (e == KL_ERR_SHAPE && l > 0) return e;
    }
    if (e == 0) {
      KlOperand op;
      memset(&op, 0, sizeof(op));
      op.A = pool + (size_t)2 * (L - 1) * W; op.lda = slot_ld; op.row_index = dev_idx; op.a_is_f32 = 1;
      op.WT_hi = d.E_hi; op.WT_lo = split == 3 ? d.E_lo : nullptr; op.ldw = W; op.K = W;
      KL_TRY(kl_launch_thin_gemm(&op, n, V, logits, V, nullptr, split, s));
      f.softmax = 1; f.slot_out = dev_idx;
      return kl_launch_step_finish(f, target ? &tx : nullptr, s);
    }
    if (e != KL_ERR_SHAPE) return e;
  }
  // ---- everything else: ONE copy of the packed indices to the device, the device-pointer step, delivery by the finish launch
  {
    const int Cc = C > 0 ? C : 1;
    std::vector<int32_t>& pk = h->host_pack;
    pk.resize((size_t)n * (4 + Cc));
    memcpy(pk.data(), idx, (size_t)n * 4);
    memcpy(pk.data() + n, slot_in, (size_t)n * 4);
    memcpy(pk.data() + 2 * (size_t)n, slot_out, (size_t)n * 4);
    if (target) memcpy(pk.data() + 3 * (size_t)n, target, (size_t)n * 4);
    else memset(pk.data() + 3 * (size_t)n, 0, (size_t)n * 4);
    if (C > 0) memcpy(pk.data() + 4 * (size_t)n, ctx, (size_t)n * C * 4);
    if (hipMemcpyAsync(dev_idx, pk.data(), pk.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess) return KL_ERR_LAUNCH;
    KL_TRY(kl_step_batch(h, n, dev_idx, C > 0 ? dev_idx + 4 * (size_t)n : nullptr, pool, dev_idx + n, dev_idx + 2 * (size_t)n, logits,
                         rest, rest_bytes, stream));
    f.softmax = 0; f.slot_out = dev_idx + 2 * (size_t)n;
    f.target = target ? dev_idx + 3 * (size_t)n : nullptr;
    return kl_launch_step_finish(f, nullptr, s);
  }
}

int kl_step_wait(const uint32_t* done_host, uint32_t ticket, double timeout_s) {
  if (!done_host) return KL_ERR_ARG;
  const volatile uint32_t* flag = done_host;
  struct timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned spins = 0;; ++spins) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == ticket) return 0;
    __builtin_ia32_pause();
    if ((spins & 0x3ff) == 0x3ff) {
      struct timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > timeout_s) {
        // (a failed launch never writes the word: report what the runtime says rather than spin for ever)
        return hipGetLastError() == hipSuccess ? KL_ERR_STATE : KL_ERR_LAUNCH;
      }
    }
  }
}

int kl_state_dist2(const kl_handle* h, int n, const float* pool, const int32_t* a, const int32_t* b, int k, float* out,
                   void* stream) {
  if (!h || !pool || !a || !b || !out || n < 1) return KL_ERR_ARG;
  if (k < 0 || k >= 2 * h->cfg.depth) return KL_ERR_ARG;
  const int W = h->cfg.width;
  hipLaunchKernelGGL(state_dist2_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, pool,
                     (long)2 * h->cfg.depth * W, W, k, a, b, n, out);
  return hip_ok(hipGetLastError());
}

int kl_test_gemm_tn(const uint16_t* A, const uint16_t* B, void* C, const float* bias, int M, int N, int K, long lda,
                    long ldb, long ldc, int out_mode, int splits, void* stream) {
  return kl_launch_gemm_tn(A, B, C, bias, M, N, K, lda, ldb, ldc, out_mode, splits, 1.f, (hipStream_t)stream);
}

int kl_test_gemm_an(const uint16_t* A_km, const uint16_t* B, float* C, int M, int N, int K, long lda_km, long ldb,
                    long ldc, int c_transposed, int b_km, void* stream) {
  return kl_launch_gemm_an(A_km, B, C, M, N, K, lda_km, ldb, ldc, c_transposed, (hipStream_t)stream, b_km);
}

int kl_test_thin_gemm(const float* A, long lda, const uint16_t* WT_hi, const uint16_t* WT_lo, long ldw, int M, int N,
                      int K, float* C, long ldc, int split, void* stream) {
  KlOperand op;
  memset(&op, 0, sizeof(op));
  op.A = A; op.lda = lda; op.a_is_f32 = 1; op.WT_hi = WT_hi; op.WT_lo = WT_lo; op.ldw = ldw; op.K = K;
  return kl_launch_thin_gemm(&op, M, N, C, ldc, nullptr, split, (hipStream_t)stream);
}

}  // extern "C"

// Tuning hook: launch `iters` identical forward cell steps (bf16 A, split 1) on raw
// buffers, `fused` copies per launch.  Used by tools/probe_step.py only.
extern "C" int kl_test_fwd_step(const uint16_t* A, long lda, const uint16_t* WT, long ldw, int K, int n_ops,
                                int n_rows, int W, int fused, int iters, float* c_out, uint16_t* h_out,
                                void* stream) {
  KlFwdStep st[4];
  if (fused < 1 || fused > 4 || n_ops < 1 || n_ops > 2) return KL_ERR_ARG;
  for (int f = 0; f < fused; ++f) {
    KlFwdStep& S = st[f];
    memset(&S, 0, sizeof(S));
    S.n_rows = n_rows; S.W = W; S.split = 1; S.n_ops = n_ops;
    for (int p = 0; p < n_ops; ++p) {
      KlOperand& o = S.op[p];
      o.A = A; o.lda = lda; o.WT_hi = WT + (size_t)(f * 2 + p) * (size_t)4 * W * ldw; o.ldw = ldw;
      o.K = K; o.a_is_f32 = 0;
    }
    S.c_out = c_out + (size_t)f * n_rows * W; S.c_out_ld = W;
    S.h_out_bf16 = h_out + (size_t)f * n_rows * W; S.h_out_bf16_ld = W;
  }
  for (int i = 0; i < iters; ++i) {
    int e = kl_launch_fwd_steps(st, fused, (hipStream_t)stream);
    if (e) return e;
  }
  return 0;
}


// ---- whole-window hipGraph replay ------------------------------------------------
// A window is 2(T+L-1) dependent step launches plus ~40 helper launches; issued
// eagerly the host launch rate (~4.5 us per launch) bounds it.  The sequence is
// captured once per (shape, pointer set) and replayed.  Inputs are first staged
// into fixed workspace slots so that replays read the new batch.
namespace {

int run_graphed(kl_handle* h, const kl_handle::GraphKey& key, hipStream_t s, const std::function<int()>& body) {
  if (!h->graphs_enabled || h->trace_on || s == nullptr) return body();   // (the legacy default stream cannot be captured)
  for (auto& g : h->graphs)
    if (g.first == key) return hip_ok(hipGraphLaunch(g.second, s));
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    h->graphs_enabled = false;
    return body();
  }
  const int e = body();
  hipGraph_t graph = nullptr;
  const hipError_t ce = hipStreamEndCapture(s, &graph);
  if (e != 0 || ce != hipSuccess || graph == nullptr) {
    if (graph) (void)hipGraphDestroy(graph);
    (void)hipGetLastError();
    h->graphs_enabled = false;
    return e != 0 ? e : body();
  }
  hipGraphExec_t exec = nullptr;
  const hipError_t ie = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ie != hipSuccess || exec == nullptr) {
    (void)hipGetLastError();
    h->graphs_enabled = false;
    return body();
  }
  if (h->graphs.size() >= 16) h->drop_graphs();
  h->graphs.emplace_back(key, exec);
  return hip_ok(hipGraphLaunch(exec, s));
}

}  // namespace

extern "C" int kl_set_loss_rows(kl_handle* h, int rows) {
  if (!h || rows < 0) return KL_ERR_ARG;
  h->loss_rows = rows;
  return 0;
}

extern "C" int kl_set_window_mode(kl_handle* h, int last_only) {
  if (!h) return KL_ERR_ARG;
  h->last_only = last_only ? 1 : 0;
  return 0;
}

extern "C" int kl_forward_window(kl_handle* h, int B, int T, const int32_t* idx, const int32_t* ctx, const int32_t* tgt,
                                 float* states, float* probs, float* loss_acc, void* ws, size_t ws_bytes,
                                 void* stream) {
  if (!h || !idx || !states || !ws || B < 1 || T < 1) return KL_ERR_ARG;
  if (h->cfg.n_ctx > 0 && !ctx) return KL_ERR_ARG;
  if (!h->precision) return KL_ERR_STATE;
  hipStream_t s = (hipStream_t)stream;
  WindowWs w;
  // (bf16 windows on a training-size workspace run the training forward: the staging slots must come from the
  // same layout the body is going to carve -- see forward_window_body)
  const int layout = (h->precision == KL_PREC_BF16 && ws_bytes >= carve_window(h, nullptr, B, T, 1, nullptr)) ? 1 : 0;
  if (ws_bytes < carve_window(h, ws, B, T, layout, &w)) return KL_ERR_WORKSPACE;
  const size_t BT = (size_t)B * T;
  KL_TRY(hip_ok(hipMemcpyAsync(w.s_idx, idx, BT * sizeof(int), hipMemcpyDeviceToDevice, s)));
  if (h->cfg.n_ctx > 0)
    KL_TRY(hip_ok(hipMemcpyAsync(w.s_ctx, ctx, BT * h->cfg.n_ctx * sizeof(int), hipMemcpyDeviceToDevice, s)));
  if (tgt) KL_TRY(hip_ok(hipMemcpyAsync(w.s_tgt, tgt, BT * sizeof(int), hipMemcpyDeviceToDevice, s)));
  kl_handle::GraphKey key{0, B, T, (tgt ? 1 : 0) | (probs ? 2 : 0) | (h->last_only ? 4 : 0) | (h->loss_rows << 3), h->precision, states, loss_acc, ws, nullptr, layout};      // (loss_rows: baked into the captured launches)
  KL_TRY(run_graphed(h, key, s, [&]() {
    return forward_window_body(h, B, T, w.s_idx, w.s_ctx, tgt ? w.s_tgt : nullptr, states, probs ? w.s_probs : nullptr,
                               loss_acc, ws, ws_bytes, stream);
  }));
  if (probs)
    KL_TRY(hip_ok(hipMemcpyAsync(probs, w.s_probs, BT * h->cfg.voc_size * sizeof(float), hipMemcpyDeviceToDevice, s)));
  return 0;
}

extern "C" int kl_train_window(kl_handle* h, int B, int T, const int32_t* idx, const int32_t* ctx, const int32_t* tgt,
                               float* states, const float* masks, float* grads, float* loss_acc, void* ws,
                               size_t ws_bytes, void* stream) {
  if (!h || !idx || !tgt || !states || !grads || !ws || B < 1 || T < 1) return KL_ERR_ARG;
  if (h->cfg.n_ctx > 0 && !ctx) return KL_ERR_ARG;
  if (h->precision != KL_PREC_BF16) return KL_ERR_STATE;
  hipStream_t s = (hipStream_t)stream;
  WindowWs w;
  if (ws_bytes < carve_window(h, ws, B, T, 1, &w)) return KL_ERR_WORKSPACE;
  const size_t BT = (size_t)B * T;
  KL_TRY(hip_ok(hipMemcpyAsync(w.s_idx, idx, BT * sizeof(int), hipMemcpyDeviceToDevice, s)));
  if (h->cfg.n_ctx > 0)
    KL_TRY(hip_ok(hipMemcpyAsync(w.s_ctx, ctx, BT * h->cfg.n_ctx * sizeof(int), hipMemcpyDeviceToDevice, s)));
  KL_TRY(hip_ok(hipMemcpyAsync(w.s_tgt, tgt, BT * sizeof(int), hipMemcpyDeviceToDevice, s)));
  if (masks)
    KL_TRY(hip_ok(hipMemcpyAsync(w.s_masks, masks, (size_t)h->cfg.depth * B * h->cfg.width * sizeof(float),
                                 hipMemcpyDeviceToDevice, s)));
  kl_handle::GraphKey key{1, B, T, (masks ? 1 : 0) | (h->last_only ? 4 : 0), h->precision, states, loss_acc, ws, grads, h->loss_rows};      // (last field: the count baked into the captured launches)
  return run_graphed(h, key, s, [&]() {
    return train_window_body(h, B, T, w.s_idx, w.s_ctx, w.s_tgt, states, masks ? w.s_masks : nullptr, grads, loss_acc,
                             ws, ws_bytes, stream);
  });
}


// ---- per-launch timing of the step kernels (bench.py roofline leg) -----------------
extern "C" int kl_trace_enable(kl_handle* h, int on) {
  if (!h) return KL_ERR_ARG;
  h->trace_on = on != 0;
  h->trace_used[0] = h->trace_used[1] = 0;
  h->trace_persistent[0] = h->trace_persistent[1] = false;
  h->trace_name[0] = "lstm_fwd_step_kernel";
  h->trace_name[1] = "lstm_bwd_step_kernel";
  return 0;
}

// kind 0 = forward cell-step launches, 1 = backward; call after synchronising the stream
extern "C" const char* kl_trace_kernel_name(kl_handle* h, int kind) {
  if (!h || kind < 0 || kind > 1) return "";
  return h->trace_name[kind];
}

extern "C" int kl_trace_read(kl_handle* h, int kind, int* n_launches, float* total_ms, int* persistent,
                             double* flops_per_launch) {
  if (!h || kind < 0 || kind > 1 || !n_launches || !total_ms) return KL_ERR_ARG;
  float total = 0.f;
  int n = 0;
  for (size_t i = 0; i < h->trace_used[kind]; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->trace_ev[kind][i].first, h->trace_ev[kind][i].second) == hipSuccess) {
      total += ms;
      ++n;
    }
  }
  // a pair brackets one persistent scan launch, or 8 consecutive launch-per-step launches
  *n_launches = h->trace_persistent[kind] ? n : 8 * n;
  *total_ms = total;
  if (persistent) *persistent = h->trace_persistent[kind] ? 1 : 0;
  // persistent scans report the contractions they actually carry; the step path carries
  // every layer's cell (recurrent + input contraction above layer 0) for B rows per launch
  if (flops_per_launch) *flops_per_launch = h->trace_persistent[kind] ? h->trace_flops[kind] : 0.0;
  return 0;
}
