// Forward scan of one layer, width 512, 32-row phases -- EIGHT waves and NO workgroup barrier inside the loop (round 4).
//
// What the stamps of lstm_scan_fwd_wide2_kernel said at 3072 streams (9 200 cycles per 32-row phase): 3 200 in the MFMA
// phase, which is bound by LDS reads (16 waves x the whole 32 KiB tile per phase = 512 KiB at 128 B/clk, against 2 048 cycles
// of MFMA work per SIMD), 2 100 in two workgroup barriers (the slowest of 16 waves sets the pace, twice per phase), 1 300 in
// the stores behind the second barrier, 1 100 in gate math -- one phase after the other, nothing overlapped.
//
// Here:
//  * 8 waves; wave w owns the EIGHT hidden units 8w .. 8w + 7 of the workgroup's 64 (two MFMA column tiles of 4 units x 4 gates
//    against one A fragment: half the LDS reads per MFMA, 128 weight registers per lane);
//  * the lane quad transpose leaves a lane with two ADJACENT units of one row (tile u holds units 2k + u), so the gate inputs
//    arrive and the gate activations leave as 16-byte pieces straight from / to registers (a 4 x 4 lane transpose makes the four
//    pieces of a row's 64 bytes consecutive lanes);
//  * h, c and the masked h go through shared staging tiles (three sets): the LAST of the eight waves to arrive (an LDS counter)
//    publishes the phase's 32 x 128-byte h lines at once -- whole lines per store as before, nobody waits for anybody --, the cell
//    states and masked outputs leave TWO phases later, four rows per wave (by then every wave has long written its part);
//  * the state tile lives in a ring of three 32 KiB buffers; a wave fetches four rows of it by LDS-DMA, and two LDS counters
//    per buffer replace the barriers: `landed` (this wave's rows of the tile are there and valid -- a reader may multiply) and
//    `released` (this wave has read its last fragment -- the zone may be armed again);
//  * NOTHING in the loop waits on vmcnt: a landing zone is armed with 0xFFFFFFFE words (no pair of finite bf16, and not the
//    producers' sentinel 0xFFFFFFFF), and the wave that asked for the rows polls the zone itself -- still armed: in flight;
//    a sentinel: the producer was late, arm and ask again; else valid.  (A counted wait is no proof with stores in the queue,
//    and vmcnt(0) waits for the write-through publishes, ~3 us each.)  First cut of this kernel, with counted waits +
//    vmcnt(0) on a miss and a per-fragment validity check: 0.66 misses and 0.73 repeated MFMA phases per phase, 4.1 ms per
//    launch against 3.3 of the 16-wave kernel;
//  * so the two waves of a SIMD drift apart and one's gate math runs under the other's MFMAs.
// Gate inputs come as bf16 P rows only (layer 0: gathered by p_gather_il_kernel in front of the scan), hand-off by rolling
// sentinels (a.sentinel == 2), XCD-local publishes where the row group's workgroups share an XCD (checked per launch).
// Restates what lstm_scan_fwd_wide2_kernel computes (the Keras LSTM cell of rating.py:130-145: gates i, f, c, o, sigmoid /
// tanh, h = o tanh(c), time-constant dropout mask on the output that feeds the layer above).
#include <stdlib.h>
#include <string.h>

#include "kl_common.h"
#include "kl_kernels.h"

namespace {

#define KL_STAMP_ARRAY kl_fwd8_stamps
#include "kl_scan_common.h"

#include "kl_scan2_helpers.h"

constexpr int F8_RING = 3;
constexpr int F8_ST_LD = 144;       // bytes per row of a staging tile (128 + 16: the quads' writes fall on distinct banks)
constexpr int F8_ST_TILE = 32 * F8_ST_LD, F8_ST_BUF = 3 * F8_ST_TILE;      // h (published) | c | masked h of one phase
// LDS map (bytes): tile [3][32][1024] | zin [8 waves][2][1024] | staging [3 phases][3 tiles][32][144] | counters [16] | row numbers [8 waves][4][32]
constexpr int F8_TILE = 0, F8_ZIN = F8_RING * 32 * 1024, F8_ST = F8_ZIN + 8 * 2 * 1024, F8_CTR = F8_ST + F8_RING * F8_ST_BUF, F8_IDS = F8_CTR + 64,
              F8_LDS = F8_IDS + 8 * 4 * 128;      // (table mode: the row numbers of four phases per wave, 32 x 4 bytes each)
// counters (words): [0, 3) released, [3] ok flag, [4, 7) landed, [8, 11) arrived at the publish, [11] XCD-local verdict

// one LDS word, re-read on every call (as asm with the LDS byte address: a volatile access through a generic pointer becomes a
// FLAT load followed by s_waitcnt vmcnt(0), i.e. a wait for every store in flight)
__device__ __forceinline__ unsigned lds_read_u32(unsigned lds_addr) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr) : "memory");
  return v;
}

constexpr unsigned F8_ARMED = 0xFFFFFFFEu;
__device__ __forceinline__ void arm16e(unsigned char* p) {
  *reinterpret_cast<uint4*>(p) = uint4{F8_ARMED, F8_ARMED, F8_ARMED, F8_ARMED};
}
// state of a 16-byte landing piece: bit 0 = still armed somewhere (in flight), bit 1 = a producer's sentinel dword
__device__ __forceinline__ unsigned piece_state(const unsigned char* p) {
  asm volatile("" ::: "memory");
  const u32x4 v = *reinterpret_cast<const u32x4*>(p);
  const bool armed = v.x == F8_ARMED || v.y == F8_ARMED || v.z == F8_ARMED || v.w == F8_ARMED;
  const bool sent = v.x == 0xFFFFFFFFu || v.y == 0xFFFFFFFFu || v.z == 0xFFFFFFFFu || v.w == 0xFFFFFFFFu;
  return (armed ? 1u : 0u) | (sent ? 2u : 0u);
}

// LS ("lockstep"): the same eight waves, two unit tiles per wave, polled landing zones and coalesced stores, but the waves
// meet at TWO workgroup barriers per phase (tile complete / staging complete) instead of going through the counters: no
// `landed` / `released` / arrival bookkeeping, no looks ahead, waves 0-3 publish and waves 4-7 store the strips of the SAME
// phase behind the second barrier.
template <int NP, bool LS>
__global__ __launch_bounds__(512, 1) void lstm_scan_fwd8_kernel(const KlScanFwdWide a) {
  constexpr int KSTEPS = 16, W = 512, NWG_RB = W / 64, ROWS = 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_rg = a.n_rg, B = a.B, T = a.T;
  const int xcd = blockIdx.x & 7, yy = blockIdx.x >> 3;
  const int cg = yy % NWG_RB, rq = yy / NWG_RB, rg = xcd * ((n_rg + 7) >> 3) + rq;
  if (rg >= n_rg) return;
  const int u0 = cg * 64, uw = u0 + 8 * wave;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned* const ctr = reinterpret_cast<unsigned*>(smem + F8_CTR);
  const unsigned lds_base = (unsigned)(size_t)(lds_void_t*)smem;
  const unsigned lds_zin = lds_base + (unsigned)(F8_ZIN + wave * 2048);
  const unsigned lds_ctr = lds_base + (unsigned)F8_CTR;
  const unsigned char* const my_zin = smem + F8_ZIN + wave * 2048;

  // cells of this lane inside a 16-row block: row 4 (lane >> 4) + (lane & 3), units uw + 2 ((lane >> 2) & 3) + {0, 1}
  const int jr = lane & 3, a4 = (lane >> 2) & 3, q4 = lane >> 4;
  const int crow = 4 * q4 + jr;

  // ---- resident weights: B fragments of 2 x 16 columns (tile u: column c = gate c & 3 of unit uw + 2 (c >> 2) + u), all of K
  u32x4 bu[2][KSTEPS];
  {
    const int col = lane & 15;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const long wrow = ((long)(col & 3) * W + uw + 2 * (col >> 2) + u) * W + (lane >> 4) * 8;
#pragma unroll
      for (int j = 0; j < KSTEPS; ++j) bu[u][j] = *reinterpret_cast<const u32x4*>(a.UT + wrow + j * 32);
    }
  }
  // cell state and dropout keep-mask of this lane's cells, per phase, block and unit (phase 0 = the current one: rotated)
  float cst[NP][2][2], mk[NP][2][2];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const long row = (long)(rg + p * n_rg) * ROWS + s * 16 + crow;
        const long at = row * W + uw + 2 * a4 + u;
        cst[p][s][u] = a.C[at];
        mk[p][s][u] = a.mask ? a.mask[at] : 1.f;
        if (a.Cb) a.Cb[at] = f2bf(cst[p][s][u]);      // (block 0: the backward scan reads every c_{t-1} as bf16)
      }
  const long BW = (long)B * W;
  const __amdgpu_buffer_rsrc_t rs_h = make_rsrc(a.H, (long)(T + 1) * BW * 2);
  const __amdgpu_buffer_rsrc_t rs_hnull = make_rsrc(a.H, 0);
  const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(a.C, (long)(T + 1) * BW * 4);
  const __amdgpu_buffer_rsrc_t rs_cb = make_rsrc(a.Cb, a.Cb ? (long)(T + 1) * BW * 2 : 0);
  const __amdgpu_buffer_rsrc_t rs_cnull = make_rsrc(a.C, 0);
  const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(a.G, (long)T * BW * 4 * 2);
  const __amdgpu_buffer_rsrc_t rs_hd = make_rsrc(a.Hd, a.Hd ? (long)T * BW * 2 : 0);          // zero records: stores dropped
  unsigned* status = a.status;
  if (tid < 16) ctr[tid] = tid == 3 ? 1u : 0u;
  // the compiler's own loads end here: everything it knows about has landed before the first asm operation is issued
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j) asm volatile("" : "+v"(bu[u][j]));
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int u = 0; u < 2; ++u) asm volatile("" : "+v"(cst[p][s][u]), "+v"(mk[p][s][u]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // XCD-local hand-off: plain publishes that stay in the shared L2 + streaming tile loads, if all column groups of this row
  // group were verified to sit on this workgroup's XCD
  bool local = false;
  if (a.xcc_slots)
    local = __builtin_amdgcn_readfirstlane(
                xcd_local_group(a.xcc_slots, a.gen, NWG_RB, [&](int j) { return xcd + 8 * (rq * NWG_RB + j); }, reinterpret_cast<int*>(ctr + 11), status) ? 1 : 0) != 0;
  SSTAMP_INIT(0);

  // wave-uniform: false once a bounded wait of this workgroup has expired (stores are dropped from then on, nothing waits any more).
  // (The helpers below RETURN whether they succeeded: a lambda that assigns to a captured variable, or captures another lambda,
  // puts its closure into scratch memory.)
  bool alive = true;
#define F8_GIVE_UP()                                                                       \
  do {                                                                                     \
    __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);            \
    ctr[3] = 0;                                                                            \
  } while (0)
  auto ctr_now = [&](int which) __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    return (unsigned)__builtin_amdgcn_readfirstlane(ctr[which]);
  };
  // wait until an LDS counter has reached `want` (bounded; a wait that expired elsewhere in the workgroup ends this one early);
  // false: given up
  auto wait_ctr = [&](int which, unsigned want) __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    if ((unsigned)__builtin_amdgcn_readfirstlane(ctr[which]) >= want) return true;
    for (unsigned spin = 0; spin < SPIN_LIMIT; ++spin) {
      __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
      if ((unsigned)__builtin_amdgcn_readfirstlane(ctr[which]) >= want) return true;
      if ((spin & 255) == 255 && __builtin_amdgcn_readfirstlane(ctr[3]) == 0) return false;
    }
    F8_GIVE_UP();
    return false;
  };
  auto bump = [&](int which) __attribute__((always_inline)) {
    if (lane == 0) __hip_atomic_fetch_add(ctr + which, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };

  // Tile image as in lstm_scan_fwd_wide2_kernel: one 1 KiB piece = one ROW of the tile, the 16-byte chunks of row r XOR-swizzled at
  // the source (chunk c at position c ^ (r & 15)).  Wave w fetches rows w, w + 8, w + 16, w + 24 of a phase.
  const unsigned dma_lane0 = (unsigned)((lane ^ wave) * 16), dma_lane1 = (unsigned)((lane ^ (wave + 8)) * 16);
  // (a macro, not a lambda: lambdas that call lambdas put their closures -- and with them the kernel arguments -- into scratch)
#define F8_PIECE_REQUEST(t_, r0_, buf_, k_)                                                       \
  do {                                                                                            \
    const int p_ = wave + 8 * (k_);                                                               \
    const unsigned soff_ = (unsigned)((((long)(t_) * B + (r0_) + p_) * W) * 2);                   \
    const unsigned dst_ = lds_base + (unsigned)(((buf_) * 32 + p_) * 1024);                       \
    if (local) glds16_nt_s(rs_h, ((k_) & 1) ? dma_lane1 : dma_lane0, soff_, dst_);                \
    else glds16_sc1_s(rs_h, ((k_) & 1) ? dma_lane1 : dma_lane0, soff_, dst_);                     \
  } while (0)
  auto arm_tile = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 4; ++k) arm16e(smem + (buf * 32 + wave + 8 * k) * 1024 + lane * 16);
  };
  auto request_tile = [&](int t, int r0, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 4; ++k) F8_PIECE_REQUEST(t, r0, buf, k);
  };
  // ONE look at this wave's four landing rows of a tile (see the head of the file): true = all valid.  A piece that has landed
  // with a producer's sentinel in it is armed and asked for again on the spot.
  auto look = [&](int t, int r0, int buf) __attribute__((always_inline)) {
    asm volatile("" ::: "memory");
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const u32x4*>(smem + (buf * 32 + wave + 8 * k) * 1024 + lane * 16);
    // (both patterns -- armed 0xFFFFFFFE, sentinel 0xFFFFFFFF -- are the two largest dwords there are, and no pair of finite bf16 comes
    //  near them: the maximum over the sixteen dwords says whether ANY piece needs a closer look -- eight v_max3 and one compare
    //  instead of thirty-two compares and eight wave votes; the look was 2 400 cycles of a 10 000-cycle phase, stamps)
    {
      unsigned m = max(max(v[0].x, v[0].y), v[0].z);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (k > 0) m = max(max(m, v[k].x), v[k].y);
        else m = max(m, v[0].w);
        if (k > 0) m = max(max(m, v[k].z), v[k].w);
      }
      if (!__any(m >= F8_ARMED)) return true;
    }
    bool valid = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool armed = v[k].x == F8_ARMED || v[k].y == F8_ARMED || v[k].z == F8_ARMED || v[k].w == F8_ARMED;
      const bool sent = v[k].x == 0xFFFFFFFFu || v[k].y == 0xFFFFFFFFu || v[k].z == 0xFFFFFFFFu || v[k].w == 0xFFFFFFFFu;
      const bool flying = __any(armed), late = __any(sent);
      if (flying || late) valid = false;
      if (late && !flying) {      // landed, but (partly) sentinels: the producer had not published yet -- once more
#ifdef KL_STAMP
        if (blockIdx.x == STAMP_WG && threadIdx.x == KL_STAMP_TID) stamp_lds[12] += 1;
#endif
        arm16e(smem + (buf * 32 + wave + 8 * k) * 1024 + lane * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        F8_PIECE_REQUEST(t, r0, buf, k);
      }
    }
    return valid;
  };
  // ... until they are all there, then the wave counts itself in as "landed" (a macro: see `alive`)
#define F8_OWN_ROWS(t_, r0_, buf_)                                                                                                  \
  do {                                                                                                                              \
    bool ok_ = false;                                                                                                               \
    for (unsigned spin = 0; spin < SPIN_LIMIT; ++spin) {                                                                            \
      if (look(t_, r0_, buf_)) { ok_ = true; break; }                                                                               \
      if ((spin & 255) == 255 && (ctr_now(3) == 0 || __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) break; \
      __builtin_amdgcn_s_sleep(1);                                                                                                  \
    }                                                                                                                               \
    if (!ok_) { F8_GIVE_UP(); alive = false; }                                                                                      \
    bump(4 + (buf_));                                                                                                               \
  } while (0)
  const unsigned frag_lane = (unsigned)((lane & 15) * 1024 + (((lane >> 4) ^ (lane & 3)) * 16) + 64 * ((lane >> 2) & 3));
  // gate inputs of a phase: 16 rows x 64 bytes (8 units x 4 gates, bf16) per block = one 1 KiB piece; lane -> (row lane >> 2, 16 bytes lane & 3)
  const unsigned z_lane = (unsigned)((((lane >> 2) * 4 * W + uw * 4) * 2) + (lane & 3) * 16);
  auto arm_zin = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 2; ++s) arm16e(const_cast<unsigned char*>(my_zin) + s * 1024 + lane * 16);
  };
  // TABLE MODE (layer 0, a.ids_tm != null): a.P is the table of ALL gate-input rows layer 0 can ask for -- row v * ctx_vocab + c =
  // EK[v] + CtxK[c] + bias, bf16, laid out like a P row -- and a.ids_tm [T][B] the row number of every position.  The 32 row
  // numbers of a phase come into LDS by a 128-byte LDS-DMA three phases ahead (armed, polled like every landing zone here: nothing
  // waits on vmcnt), each lane then asks for its row's 64 bytes of this wave's units.  (A gather kernel writing P rows first costs
  // 1.25 ms per window at 3072 streams; the table is 200 MiB, rebuilt in 0.06 ms after every update.)
  const bool tab = a.ids_tm != nullptr;
  const __amdgpu_buffer_rsrc_t rs_ids = make_rsrc(a.ids_tm, tab ? (long)T * B * 4 : 0);
  const __amdgpu_buffer_rsrc_t rs_comb = make_rsrc(a.P, tab ? (long)a.V * a.ctx_vocab * 4 * W * 2 : 0);
  unsigned char* const my_ids = smem + F8_IDS + wave * 512;
  const unsigned lds_ids = lds_base + (unsigned)(F8_IDS + wave * 512);
  const unsigned z_piece = (unsigned)(uw * 4 * 2 + (lane & 3) * 16);
  auto arm_ids = [&](int slot) __attribute__((always_inline)) {
    if (lane < 32) *reinterpret_cast<unsigned*>(my_ids + slot * 128 + lane * 4) = F8_ARMED;
  };
  auto request_ids = [&](int slot, int t, int r0) __attribute__((always_inline)) {
    if (lane < 32) glds4_plain_s(rs_ids, (unsigned)(lane * 4), (unsigned)(((long)t * B + r0) * 4), lds_ids + (unsigned)(slot * 128));
  };
  auto request_zin = [&](int t, int r0, int slot) __attribute__((always_inline)) {
    if (tab) {
      unsigned id[2] = {0u, 0u};
      bool ok = false;
      for (unsigned spin = 0; spin < SPIN_LIMIT; ++spin) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < 2; ++s) id[s] = *reinterpret_cast<const unsigned*>(my_ids + slot * 128 + (s * 16 + (lane >> 2)) * 4);
        if (!__any(id[0] == F8_ARMED || id[1] == F8_ARMED)) { ok = true; break; }
        if ((spin & 255) == 255 && ctr_now(3) == 0) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (!ok) { F8_GIVE_UP(); alive = false; return; }
#pragma unroll
      for (int s = 0; s < 2; ++s) glds16_plain(rs_comb, id[s] * (unsigned)(4 * W * 2) + z_piece, lds_zin + (unsigned)(s * 1024));
      return;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const __amdgpu_buffer_rsrc_t rs_p = make_rsrc(reinterpret_cast<const bf16_t*>(a.P) + ((long)t * B + r0 + s * 16) * 4 * W, (long)16 * 4 * W * 2);
      glds16_plain(rs_p, z_lane, lds_zin + (unsigned)(s * 1024));
    }
  };
  // G store after the lane transpose: lane d = 16 q + 4 j + k takes the piece of lane 16 q + 4 k + j and holds row 4 q + j, unit pair k
  const int g_src = ((lane & 48) | ((lane & 3) << 2) | ((lane >> 2) & 3)) * 4;
  const unsigned g_lane = (unsigned)(((4 * (lane >> 4) + ((lane >> 2) & 3)) * W + uw + 2 * (lane & 3)) * 8);
  // cell states (bf16, block t + 1 of Cb) and masked outputs of a finished phase from its staging tiles: wave w stores rows
  // 4 w .. 4 w + 3, lanes 0 .. 31 the cell states, lanes 32 .. 63 the masked outputs, 16 bytes each
  const int st_i = lane & 31, st_row = 4 * wave + (st_i >> 3), st_seg = st_i & 7;
  auto strips_read = [&](int sbuf) __attribute__((always_inline)) {
    return *reinterpret_cast<const uint4*>(smem + F8_ST + sbuf * F8_ST_BUF + (lane < 32 ? 1 : 2) * F8_ST_TILE + st_row * F8_ST_LD + st_seg * 16);
  };
  // (the descriptors are PARAMETERS: a choice between two captured variables inside a lambda becomes an indexed access into its
  // closure, which then lives in scratch -- and the kernel arguments with it)
  auto strips_store = [&](unsigned trow_p, uint4 v, __amdgpu_buffer_rsrc_t to_cb, __amdgpu_buffer_rsrc_t to_hd) __attribute__((always_inline)) {
    const unsigned off = (unsigned)((st_row * W + u0 + st_seg * 8) * 2);
    if (lane < 32) {
      if (a.Cb) store16(to_cb, off, (trow_p + B) * (unsigned)(W * 2), v);
    } else {
      store16(to_hd, off, trow_p * (unsigned)(W * 2), v);
    }
  };
  unsigned prev_trow[2] = {0u, 0u};      // first time-major row of the last phase and of the one before
  // where the tile of a later phase is requested (a.pf_mode): 1 = two phases ahead behind this phase's MFMAs (three or more
  // phases per step: its rows were published at least a phase before), 2 = two ahead at the top, 0 = one ahead at the top,
  // 3 = one ahead behind the MFMAs (two phases per step: the rows it asks for are those of the phase just published)
  const int ahead = (a.pf_mode == 1 || a.pf_mode == 2) ? 2 : 1;
  const bool at_top = a.pf_mode == 0 || a.pf_mode == 2;
  const int n_phases = T * NP;
  auto phase_tr = [&](int m, int& t, int& r0) __attribute__((always_inline)) {
    t = m / NP;
    r0 = (rg + (m - t * NP) * n_rg) * ROWS;
  };
  // ---- prologue: (table mode) the row numbers of the first three phases; the tiles of the first `ahead` phases, the gate inputs of the first
  if (tab) {
    for (int m = 0; m < 3 && m < n_phases; ++m) arm_ids(m);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int m = 0; m < 3 && m < n_phases; ++m) {
      int t, r0;
      phase_tr(m, t, r0);
      request_ids(m, t, r0);
    }
  }
  for (int m = 0; m < ahead && m < n_phases; ++m) {
    int t, r0;
    phase_tr(m, t, r0);
    arm_tile(m % F8_RING);
    if (m == 0) arm_zin();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    request_tile(t, r0, m % F8_RING);
    if (m == 0) request_zin(0, rg * ROWS, 0);
  }

  bool mine = false;      // this wave's rows of the phase at hand have been seen valid (and counted) already
  int n = 0;
  for (int t = 0; t < T; ++t) {
#pragma unroll 1
    for (int ip = 0; ip < NP; ++ip, ++n) {
      const int buf = n % F8_RING, buf1 = (n + 1) % F8_RING;
      const int r0 = (rg + ip * n_rg) * ROWS;
      int t1 = t, ip1 = ip + 1;
      if (ip1 >= NP) { ip1 = 0; t1 = t + 1; }
      const int r1 = (rg + ip1 * n_rg) * ROWS;
      const bool have_next = t1 < T;
      int ta, ra;
      phase_tr(n + ahead, ta, ra);
      const bool ask = n + ahead < n_phases;
      const int bufa = (n + ahead) % F8_RING;
      SSTAMP(0);
      if (at_top && alive && ask) {
        // (the buffer was last read in phase n + ahead - 3: all eight waves must have released it -- LS: they met at a barrier since)
        if (!LS) alive = wait_ctr(bufa, 8u * (unsigned)((n + ahead) / F8_RING));
        arm_tile(bufa);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (alive) request_tile(ta, ra, bufa);
      }
      // ---- this wave's own rows of the tile (normally seen during the phase before), then everybody's
      if (alive && (LS || !mine)) F8_OWN_ROWS(t, r0, buf);
      SSTAMP(8);
      if (LS) {
        __syncthreads();
        alive = ctr_now(3) != 0;
      } else if (alive) {
        alive = wait_ctr(4 + buf, 8u * (unsigned)(n / F8_RING + 1));
      }
      SSTAMP(1);
      // ---- MFMA phase: two row blocks x two unit tiles against the resident weights, all of K
      f32x4 acc[2][2];
      {
        const unsigned char* tb = smem + buf * 32 * 1024;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int u = 0; u < 2; ++u) acc[s][u] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 fr[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) fr[0][s] = *reinterpret_cast<const u32x4*>(tb + frag_lane + s * 16 * 1024);
#pragma unroll
        for (int q = 0; q < KSTEPS; ++q) {
          if (q + 1 < KSTEPS) {
            const int q1 = q + 1;
            const unsigned char* ap = tb + (frag_lane ^ (unsigned)(64 * (q1 >> 2))) + 256 * (q1 & 3);
#pragma unroll
            for (int s = 0; s < 2; ++s) fr[q1 & 1][s] = *reinterpret_cast<const u32x4*>(ap + s * 16 * 1024);
          }
          __builtin_amdgcn_sched_barrier(0);
          const int j = 4 * (q & 3) + (q >> 2);
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < 2; ++u)
              acc[s][u] = mfma16(__builtin_bit_cast(bf16x8, fr[q & 1][s]), __builtin_bit_cast(bf16x8, bu[u][j]), acc[s][u]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (!LS) bump(buf);      // released: this wave's last fragment read of the buffer is in the LDS queue in front of this
      SSTAMP(2);
      // ---- the phase TWO before this one: all eight waves have long arrived at its publish (i.e. written their staging rows --
      // checked, not assumed), its cell states and masked outputs leave now, whole 128-byte lines, four rows per wave.  (In front
      // of the looks below: a wave that finds `landed` complete for a phase knows that everybody has read the staging set that
      // phase will overwrite.)
      if (!LS && n > 1 && alive) {
        alive = wait_ctr(8 + (n - 2) % F8_RING, 8u * (unsigned)((n - 2) / F8_RING + 1));
        if (alive) strips_store(prev_trow[1], strips_read((n - 2) % F8_RING), rs_cb, rs_hd);
      }
      SSTAMP(11);
      // ---- a first look at this wave's rows of the NEXT phase (asked for behind the MFMAs of the phase before): a wave that is
      // ahead does not hold the others up at the top of the next phase
      // (only if they HAVE been asked for: with requests one phase ahead the buffer still holds the tile of three phases ago)
      mine = false;
      if (!LS && have_next && alive && ahead == 2) {
        mine = look(t1, r1, buf1);
        if (mine) bump(4 + buf1);
      }
      SSTAMP(9);
      // ---- gate inputs of this phase (asked for a phase ago): 16 bytes per block = the 4 gates of this lane's two units
      u32x4 zi[2];
      {
        bool ok = false;
        for (unsigned spin = 0; spin < SPIN_LIMIT; ++spin) {
          asm volatile("" ::: "memory");
#pragma unroll
          for (int s = 0; s < 2; ++s) zi[s] = *reinterpret_cast<const u32x4*>(my_zin + s * 1024 + crow * 64 + a4 * 16);
          // (the armed pattern is larger than any pair of finite bf16: one maximum over the eight dwords)
          const unsigned zm = max(max(max(zi[0].x, zi[0].y), max(zi[0].z, zi[0].w)), max(max(zi[1].x, zi[1].y), max(zi[1].z, zi[1].w)));
          if (!__any(zm >= F8_ARMED)) { ok = true; break; }
#ifdef KL_STAMP
          if (blockIdx.x == STAMP_WG && threadIdx.x == KL_STAMP_TID && spin == 0) stamp_lds[13] += 1;
#endif
          if ((spin & 255) == 255 && ctr_now(3) == 0) break;
          __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) { F8_GIVE_UP(); alive = false; }
      }
      SSTAMP(10);
      // ---- requests: the tile `ahead` phases on (its buffer must have been released by all eight waves), the next phase's gate inputs
      {
        const bool ask_tile = !at_top && alive && ask;
        if (ask_tile) {
          if (!LS) alive = wait_ctr(bufa, 8u * (unsigned)((n + ahead) / F8_RING));
          arm_tile(bufa);
        }
        if (have_next) arm_zin();
        const bool ids_next = tab && n + 3 < n_phases;      // (slot (n + 3) & 3 held phase n - 1's numbers, used a phase ago)
        if (ids_next) arm_ids((n + 3) & 3);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(zi[0]), "+v"(zi[1])::"memory");
        if (ask_tile && alive) request_tile(ta, ra, bufa);
        if (have_next && alive) request_zin(t1, r1, (n + 1) & 3);
        if (ids_next && alive) {
          int t3, r3;
          phase_tr(n + 3, t3, r3);
          request_ids((n + 3) & 3, t3, r3);
        }
      }
      SSTAMP(3);
      // ---- epilogue on the accumulators: after the quad transpose lane = (row, unit pair), registers = gates
      const unsigned trow = (unsigned)(t * B + r0);          // first time-major row of this phase
      unsigned char* const stage = smem + F8_ST + buf * F8_ST_BUF;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float hv[2], cv[2], hm[2];
        u32x4 gp;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          quad_transpose(acc[s][u], jr);
          const unsigned p01 = u ? zi[s].z : zi[s].x, p23 = u ? zi[s].w : zi[s].y;
          const float z0 = acc[s][u][0] + u2f(p01 << 16), z1 = acc[s][u][1] + u2f(p01 & 0xffff0000u);
          const float z2 = acc[s][u][2] + u2f(p23 << 16), z3 = acc[s][u][3] + u2f(p23 & 0xffff0000u);
          const float gi = fast_sigmoid(z0), gf = fast_sigmoid(z1), gg = fast_tanh(z2), go = fast_sigmoid(z3);
          const float c = gf * cst[0][s][u] + gi * gg;
          cst[0][s][u] = c;
          const float h = go * fast_tanh(c);
          hv[u] = h; cv[u] = c; hm[u] = h * mk[0][s][u];
          const unsigned g01 = (unsigned)f2bf(gi) | ((unsigned)f2bf(gf) << 16), g23 = (unsigned)f2bf(gg) | ((unsigned)f2bf(go) << 16);
          if (u) { gp.z = g01; gp.w = g23; } else { gp.x = g01; gp.y = g23; }
        }
        const int row = s * 16 + crow;
        // gate activations: 16 bytes per lane of G[trow + row][uw + 2 a4 ..][4].  The quad transpose leaves consecutive lanes
        // in different ROWS; a 4 x 4 lane transpose (row-in-4 <-> unit pair) puts the four pieces of a row's 64 bytes into
        // consecutive lanes, so the address unit sends 16 requests of 64 bytes instead of 64 of 16
#pragma unroll
        for (int k = 0; k < 4; ++k) gp[k] = (unsigned)__builtin_amdgcn_ds_bpermute(g_src, (int)gp[k]);
        store16(rs_g, g_lane + (unsigned)(s * 16 * W * 8), trow * (unsigned)(W * 8), uint4{gp.x, gp.y, gp.z, gp.w});
        const int at = row * F8_ST_LD + (8 * wave + 2 * a4) * 2;
        *reinterpret_cast<unsigned*>(stage + at) = (unsigned)f2bf(hv[0]) | ((unsigned)f2bf(hv[1]) << 16);
        *reinterpret_cast<unsigned*>(stage + F8_ST_TILE + at) = (unsigned)f2bf(cv[0]) | ((unsigned)f2bf(cv[1]) << 16);
        *reinterpret_cast<unsigned*>(stage + 2 * F8_ST_TILE + at) = (unsigned)f2bf(hm[0]) | ((unsigned)f2bf(hm[1]) << 16);
        if (t == T - 1) {      // the carried-out state stays f32 (block T of C)
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, cv[0]), __builtin_bit_cast(unsigned, cv[1])},
                                                alive ? rs_c : rs_cnull, (int)((row * W + uw + 2 * a4) * 4), (int)((trow + B) * (unsigned)(W * 4)), 0);
        }
      }
      prev_trow[1] = prev_trow[0];
      prev_trow[0] = trow;
      SSTAMP(4);
      // ---- publish: the last wave to arrive stores the phase's 32 rows x 128 bytes (and arms block t + 3).  (The staging writes
      // above and the counter update pass through the LDS queue in order: no wait in between.)
      if (LS) {
        // second barrier: the staging tiles are complete; waves 0-3 publish (one 16-byte piece per lane, 8 rows per wave) and arm
        // block t + 3, waves 4-7 store the cell states and masked outputs of this same phase
        __syncthreads();
        if (wave < 4) {
          const int i = wave * 64 + lane, prow = i >> 3, seg = i & 7;
          const uint4 v = *reinterpret_cast<const uint4*>(stage + prow * F8_ST_LD + seg * 16);
          const unsigned off = (unsigned)((prow * W + seg * 8) * 2) + ((trow + B) * W + u0) * 2u;
          if (!alive) store16(rs_hnull, 0u, 0u, v);
          else if (local) store16(rs_h, off, 0u, v);
          else store16_sc1(rs_h, off, v);
          const uint4 ones = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
          const unsigned off3 = off + (unsigned)((long)2 * B * W * 2);
          if (!alive || t + 3 > T) store16(rs_hnull, 0u, 0u, ones);
          else if (local) store16(rs_h, off3, 0u, ones);
          else store16_sc1(rs_h, off3, ones);
        } else {
          const int i = (wave - 4) * 64 + lane, prow = i >> 3, seg = i & 7;
          const unsigned off = (unsigned)((prow * W + u0 + seg * 8) * 2);
          const uint4 vc = *reinterpret_cast<const uint4*>(stage + F8_ST_TILE + prow * F8_ST_LD + seg * 16);
          const uint4 vh = *reinterpret_cast<const uint4*>(stage + 2 * F8_ST_TILE + prow * F8_ST_LD + seg * 16);
          if (a.Cb) store16(alive ? rs_cb : rs_cnull, off, (trow + B) * (unsigned)(W * 2), vc);
          store16(alive ? rs_hd : rs_cnull, off, trow * (unsigned)(W * 2), vh);
        }
      } else {
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(ctr + 8 + buf, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old == 8u * (unsigned)(n / F8_RING) + 7u) {
          SSTAMP(5);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int i = k * 64 + lane, prow = i >> 3, seg = i & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(stage + prow * F8_ST_LD + seg * 16);
            const unsigned off = (unsigned)((prow * W + seg * 8) * 2) + ((trow + B) * W + u0) * 2u;
            if (!alive) store16(rs_hnull, 0u, 0u, v);
            else if (local) store16(rs_h, off, 0u, v);        // stays in this XCD's L2, where all its readers are
            else store16_sc1(rs_h, off, v);
            // rolling sentinels: the same lanes arm block t + 3 while they publish block t + 1 (blocks 1 and 2 are pre-filled)
            const uint4 ones = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            const unsigned off3 = off + (unsigned)((long)2 * B * W * 2);
            if (!alive || t + 3 > T) store16(rs_hnull, 0u, 0u, ones);
            else if (local) store16(rs_h, off3, 0u, ones);
            else store16_sc1(rs_h, off3, ones);
          }
          SSTAMP(6);
        }
      }
      // a second look at the next phase's rows
      if (!LS && have_next && alive && !mine) {
        mine = look(t1, r1, buf1);
        if (mine) bump(4 + buf1);
      }
      SSTAMP(7);
      // the next phase's cells move to slot 0
      if (NP > 1) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const float c0 = cst[0][s][u], m0 = mk[0][s][u];
#pragma unroll
            for (int p = 0; p + 1 < NP; ++p) { cst[p][s][u] = cst[p + 1][s][u]; mk[p][s][u] = mk[p + 1][s][u]; }
            cst[NP - 1][s][u] = c0; mk[NP - 1][s][u] = m0;
          }
      }
    }
  }
  // the last phase's cell states and masked outputs
  __syncthreads();
  if (!LS && alive && ctr_now(3) != 0) {
    if (n > 1) strips_store(prev_trow[1], strips_read((n - 2) % F8_RING), rs_cb, rs_hd);
    strips_store(prev_trow[0], strips_read((n - 1) % F8_RING), rs_cb, rs_hd);
  }
  SSTAMP_FLUSH();
}

#undef F8_OWN_ROWS
#undef F8_PIECE_REQUEST
#undef F8_GIVE_UP

}  // namespace

#ifdef KL_STAMP
extern "C" int kl_test_fwd8_stamps(unsigned long long* out, int reset) {
  if (reset) {
    unsigned long long zeros[32] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(kl_fwd8_stamps), zeros, sizeof(zeros)) == hipSuccess ? 0 : KL_ERR_LAUNCH;
  }
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(kl_fwd8_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : KL_ERR_LAUNCH;
}
#endif

// KL_ERR_SHAPE = not applicable (the caller takes lstm_scan_fwd_wide2_kernel)
int kl_launch_scan_fwd8(KlScanFwdWide a, hipStream_t stream, bool lockstep) {
  if (a.W != 512 || !a.P || !a.p_bf16 || a.sentinel != 2 || a.HT || a.HdT || !a.G || !a.H || !a.C) return KL_ERR_SHAPE;
  if (a.ids_tm && (a.V < 1 || a.ctx_vocab < 1 || (long)a.V * a.ctx_vocab * 4 * a.W * 2 > 0xfffffff0L)) return KL_ERR_SHAPE;      // (table mode: a.P = the table of all rows)
  const int np = kl_scan_wide2_phases(a.B, a.T, a.W, 32, 4);
  if (np < 2) return KL_ERR_SHAPE;
  a.n_rb = a.B / 32;
  a.n_rg = a.n_rb / np;
  if (a.pf_mode < 0 || a.pf_mode > 3) return KL_ERR_ARG;
  if (np == 2 && (a.pf_mode == 1 || a.pf_mode == 2)) a.pf_mode = 3;      // (two phases per step: the rows of phase n + 2 are this phase's own)
  dim3 grid(8 * (a.W / 64) * ((a.n_rg + 7) / 8)), block(512);
  const size_t lds = (size_t)F8_LDS;
#define KL_F8_CASE1(NP_, LS_)                                                                                            \
  do {                                                                                                                   \
    static KlLdsGrant grant;                                                                                             \
    if (kl_grant_lds(grant, reinterpret_cast<const void*>(&lstm_scan_fwd8_kernel<NP_, LS_>), lds)) return KL_ERR_LAUNCH; \
    hipLaunchKernelGGL((lstm_scan_fwd8_kernel<NP_, LS_>), grid, block, lds, stream, a);                                  \
  } while (0)
#define KL_F8_CASE(NP_) do { if (lockstep) KL_F8_CASE1(NP_, true); else KL_F8_CASE1(NP_, false); } while (0)
  if (np == 2) KL_F8_CASE(2);
  else if (np == 3) KL_F8_CASE(3);
  else KL_F8_CASE(4);
#undef KL_F8_CASE
#undef KL_F8_CASE1
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}
