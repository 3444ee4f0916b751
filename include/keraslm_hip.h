/* keraslm_hip.h -- C ABI of the MI355X (gfx950) Rater hot path.
 *
 * This is the drop-in boundary BELOW the reference's Python `Rater` class.  In
 * the reference the boundary is the Keras model object `Rater.model`
 * (ocrd_keraslm/lib/rating.py:56, built at rating.py:61-179); every entry point
 * here replaces one family of calls the reference makes on that object:
 *
 *   reference call site (ocrd_keraslm/lib/rating.py)        entry point here
 *   -------------------------------------------------------  -------------------
 *   Model(...) / compile, :171-178                           kl_create, kl_bind, kl_prepare
 *   model.get_weights / set_weights, :398-412, :435-458      the caller-owned flat f32 parameter vector
 *                                                            (layout: kl_param_layout) + kl_prepare
 *   model.predict_generator (windows), :516                  kl_forward_window (tgt = NULL)
 *   model.evaluate_generator, :490                           kl_forward_window (tgt != NULL)
 *   model.predict_on_batch, stateful (1,1) step, :566        kl_forward_window with T = 1
 *   model.predict_on_batch, incremental + states, :631       kl_step_batch
 *   model.fit_generator -> train_on_batch, :292-298          kl_train_window + kl_adam_step
 *   model.reset_states, :475, :555, callbacks.py:58,69       caller zeroes its state rows
 *
 * Conventions: plain pointers and sizes only; every pointer marked "device" is
 * HBM memory owned by the CALLER (this library never allocates or frees device
 * memory and never retains a pointer beyond what kl_bind documents); every call
 * is asynchronous on the given hipStream_t (passed as void*); return value 0 =
 * success, otherwise a KL_ERR_* code (kl_error_string).  Nothing here throws.
 *
 * Data layout in HBM
 *   parameters   one flat f32 vector in Keras weight order: char embedding E [V][W],
 *                context embeddings Ctx_n [200][10], then per layer kernel K_l
 *                [D_l][4W], recurrent kernel U_l [W][4W], bias b_l [4W]; gate column
 *                order i,f,c,o (rating.py:103-145).  Gradients / Adam moments use
 *                the same layout.
 *   states       [rows][2L][W] f32, per row h1,c1,...,hL,cL (the order of
 *                Rater.predict's state lists, rating.py:622-629).  A "row" is a
 *                stateful stream (windows) or a state-pool slot (hypotheses).
 *   idx / tgt    int32 [B][T]; tgt = -1 marks an all-zero one-hot row (the padded
 *                tail of the last window, rating.py:1096-1102); ctx int32 [B][T][n_ctx].
 *   probs        f32 [B][T][V] (windows) or [n][V] (steps).
 */
#ifndef KERASLM_HIP_H
#define KERASLM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KL_ABI_VERSION 1

enum {
  KL_OK = 0,
  KL_ERR_SHAPE = 1,      /* unsupported or inconsistent dimensions */
  KL_ERR_LAUNCH = 2,     /* HIP launch / runtime error */
  KL_ERR_STATE = 3,      /* call order violated (e.g. not bound / not prepared) */
  KL_ERR_WORKSPACE = 4,  /* workspace too small */
  KL_ERR_ARG = 5         /* null or invalid argument */
};

/* precision of the contractions */
#define KL_PREC_BF16 1   /* bf16 MFMA operands, f32 accumulate (training)          */
#define KL_PREC_SPLIT 3  /* hi/lo split bf16 (3 MFMAs), ~f32 accuracy (rating)     */

typedef struct kl_config {
  int32_t depth;      /* L  (Rater.depth, rating.py:40)                            */
  int32_t width;      /* W  (Rater.width, rating.py:39); multiple of 32            */
  int32_t voc_size;   /* V  (Rater.voc_size, rating.py:59)                         */
  int32_t n_ctx;      /* number of context variables (1 in the reference)          */
  int32_t ctx_vocab;  /* 200 (rating.py:111)                                       */
  int32_t ctx_dim;    /* 10  (rating.py:111)                                       */
} kl_config;

typedef struct kl_handle kl_handle;

int kl_abi_version(void);
const char* kl_error_string(int code);

/* Parameter vector layout.  kl_param_layout iterates over the weight arrays in
 * Keras order; returns KL_ERR_ARG past the end. */
size_t kl_param_count(const kl_config* cfg);
int kl_param_layout(const kl_config* cfg, int index, char* name, size_t name_cap, size_t* offset, size_t* rows,
                    size_t* cols);

/* Host-only object: configuration + launch plans.  (rating.py:61-179) */
kl_handle* kl_create(const kl_config* cfg);
void kl_destroy(kl_handle* h);

/* Bind the caller's parameter vector and a scratch area for derived weights
 * (bf16 transposed copies, hi/lo splits, layer-0 look-up tables).  Both must
 * stay valid until the next kl_bind / kl_destroy. */
size_t kl_derived_bytes(const kl_handle* h);
int kl_bind(kl_handle* h, float* params /*device*/, void* derived /*device*/, size_t derived_bytes);

/* Recompute the derived weights from the bound parameters.  Call after loading
 * weights and after every parameter change made outside kl_adam_step. */
int kl_prepare(kl_handle* h, int precision, void* stream);

/* Workspace (device bytes) needed by the window calls for at most B x T. */
size_t kl_window_workspace_bytes(const kl_handle* h, int B, int T, int training);

/* Windowed forward (rating.py:490, 516, 566): B stateful streams x T steps.
 * states [B][2L][W] is read as the carried-in state and overwritten with the
 * state after step T-1.  probs (may be NULL) receives [B][T][V].  If tgt != NULL,
 * loss_acc[0] += mean categorical cross-entropy over all B*T positions and
 * loss_acc[1] += accuracy (Keras semantics, rating.py:178); loss_acc is f32[4]. */
/* Window mode of kl_forward_window / kl_train_window.  0 (default) = the reference's stateful graph:
 * a target at every position, loss and accuracy are means over all B*T positions (rating.py:161-167,
 * TimeDistributed output).  1 = its stateless graph (top LSTM layer without return_sequences,
 * rating.py:126-129, 1123-1126): one target per row, at the LAST position (tgt[b][T-1]); loss and
 * accuracy are means over the B rows, the other positions carry no gradient. */
int kl_set_window_mode(kl_handle* h, int last_only);

/* Rows the means of kl_train_window (and of kl_forward_window in bf16 precision on a training-size workspace: validation
 * windows) are taken over: 0 (default) = its B.  A caller that PADS a batch with dummy streams
 * (targets -1: no loss, no gradient) up to a stream count the persistent scans are instantiated for passes the real count
 * here, so that loss, accuracy and gradient stay those of the reference's mean over the real B*T positions
 * (rating.py:178, Keras' mean over the batch). */
int kl_set_loss_rows(kl_handle* h, int rows);

int kl_forward_window(kl_handle* h, int B, int T, const int32_t* idx, const int32_t* ctx, const int32_t* tgt,
                      float* states, float* probs, float* loss_acc, void* ws, size_t ws_bytes, void* stream);

/* One training batch, forward + backward (rating.py:292-298 -> train_on_batch):
 * writes the gradient of (mean CE + embedding regularisers, rating.py:187-246)
 * into grads (param layout; overwritten), advances states, and accumulates
 * loss_acc[0] CE, [1] accuracy, [2] regulariser value.  dropout_masks is NULL or
 * f32 [L][B][W] keep-masks already scaled by 1/0.9 (entry 0 unused; time-constant
 * dropout after every layer with index > 0, rating.py:146-152). */
int kl_train_window(kl_handle* h, int B, int T, const int32_t* idx, const int32_t* ctx, const int32_t* tgt,
                    float* states, const float* dropout_masks, float* grads, float* loss_acc, void* ws,
                    size_t ws_bytes, void* stream);

/* Keras-2.3 Adam with clipvalue (rating.py:178): g <- clip(g,-clip,clip);
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v updates; p -= lr_t*m/(sqrt(v)+eps); then
 * re-derives the bf16 weights (as kl_prepare).  t starts at 1. */
int kl_adam_step(kl_handle* h, const float* grads, float* m, float* v, int t, float lr, float b1, float b2,
                 float eps, float clip, void* stream);
/* The same with the gradients read as grads[i] * grad_scale (before the clip).  Data-parallel training (not a
 * reference feature: rating.py:295 trains with workers=1) all-reduces the flat gradient vector with SUM and passes
 * 1 / world size here, so the mean over the ranks costs no pass of its own. */
int kl_adam_step_scaled(kl_handle* h, const float* grads, float grad_scale, float* m, float* v, int t, float lr,
                        float b1, float b2, float eps, float clip, void* stream);

/* Incremental step for n hypotheses with explicit states (rating.py:578-639):
 * row i reads its state from pool slot slot_in[i] and writes the new state to
 * slot_out[i] (slot_out must not alias any slot_in of the same call); probs
 * receives [n][V].  pool is [n_slots][2L][W] f32.  ctx is [n][n_ctx]. */
size_t kl_step_workspace_bytes(const kl_handle* h, int n);
int kl_step_batch(kl_handle* h, int n, const int32_t* idx, const int32_t* ctx, float* pool, const int32_t* slot_in,
                  const int32_t* slot_out, float* probs, void* ws, size_t ws_bytes, void* stream);

/* The same step as a BEAM SEARCH issues it (rating.py:809-826 rate_best, :689-691 generate, each through
 * Rater.predict, rating.py:578-639): one call per character with the GPU idle in between, so what counts is the time
 * from the call to the numbers being in the caller's hands, not the kernels' own.  Differences to kl_step_batch:
 *   - idx, ctx, slot_in, slot_out (and target) are HOST arrays; up to 256 hypotheses they travel inside the kernel
 *     arguments of the launches (no copy to the device, no index array for the kernels to chase), beyond that in one copy;
 *   - the results are delivered into HOST memory the device can write (kl_host_alloc): probs_host receives [n][V], or --
 *     target != NULL -- [n]: per row the probability of character target[i] alone (all a lattice decoder looks at,
 *     rating.py:838-843); head_k > 0 also delivers the first head_k state vectors of every new state, [n][head_k][W]
 *     (history clustering compares exactly those, rating.py:887-916);
 *   - nothing has to synchronise with the stream: *done_host becomes `ticket` once everything above has arrived
 *     (kl_step_wait spins on that word; use a different ticket for every step).
 * pool, ws: device.  ws_bytes >= kl_step_host_workspace_bytes(h, n); one workspace per handle and stream. */
void* kl_host_alloc(size_t bytes);      /* zeroed, page-locked, device-visible host memory; NULL on failure */
void kl_host_free(void* p);
size_t kl_step_host_workspace_bytes(const kl_handle* h, int n);
int kl_step_batch_host(kl_handle* h, int n, const int32_t* idx, const int32_t* ctx, const int32_t* slot_in,
                       const int32_t* slot_out, const int32_t* target, float* pool, int head_k, float* probs_host,
                       float* heads_host, uint32_t* done_host, uint32_t ticket, void* ws, size_t ws_bytes, void* stream);
/* returns 0 once *done_host == ticket; KL_ERR_LAUNCH / KL_ERR_STATE after timeout_s seconds without it */
int kl_step_wait(const uint32_t* done_host, uint32_t ticket, double timeout_s);

/* Squared L2 distances between state vectors of pool slots, for beam history
 * clustering (rating.py:887-916): out[i] = || pool[a[i]][k] - pool[b[i]][k] ||^2
 * for state entry k (0 = h1, 1 = c1, ...). */
int kl_state_dist2(const kl_handle* h, int n, const float* pool, const int32_t* a, const int32_t* b, int k,
                   float* out, void* stream);

/* Per-launch timing of the recurrence kernels with HIP events on the caller's
 * stream (used by bench.py for the roofline figure).  While enabled, windows are
 * issued eagerly and the recurrence launches are bracketed by event pairs: each
 * whole-window persistent scan launch, or -- on the launch-per-step path -- runs of
 * 8 consecutive steady-state step launches.  kl_trace_read (after a stream
 * synchronise) returns how many launches were timed, their summed duration and
 * whether they were persistent scans (then also the LSTM contraction FLOPs one such
 * launch carries); kind 0 = forward, 1 = backward recurrence. */
int kl_trace_enable(kl_handle* h, int on);
int kl_trace_read(kl_handle* h, int kind, int* n_launches, float* total_ms, int* persistent,
                  double* flops_per_launch);
/* name of the kernel whose launches kind 0 (forward) / 1 (backward) timed, as rocprofv3 lists it */
const char* kl_trace_kernel_name(kl_handle* h, int kind);

/* Test hooks: the bare contraction kernels on caller buffers. */
int kl_test_gemm_tn(const uint16_t* A, const uint16_t* B, void* C, const float* bias, int M, int N, int K, long lda,
                    long ldb, long ldc, int out_mode, int splits, void* stream);
/* the weight-gradient contraction with a K-major A operand: C[m][n] (or C[n][m] if c_transposed) +=
 * sum_k A_km[k][m] * B[n][k] (b_km: B_km[k][n]); KL_ERR_SHAPE where it does not apply (M % 256, K % 64) */
int kl_test_gemm_an(const uint16_t* A_km, const uint16_t* B, float* C, int M, int N, int K, long lda_km, long ldb,
                    long ldc, int c_transposed, int b_km, void* stream);
int kl_test_thin_gemm(const float* A, long lda, const uint16_t* WT_hi, const uint16_t* WT_lo, long ldw, int M, int N,
                      int K, float* C, long ldc, int split, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KERASLM_HIP_H */
