// Second generation of the layer-sequential WIDE scans (many streams: two or more row phases per
// workgroup and step).  Same decomposition as lstm_scan.hip's wide kernels -- one layer per launch,
// 1024-thread workgroups of 64 hidden units with the recurrent weights resident in registers, the
// state tile fetched once per workgroup by LDS-DMA and shared, hand-off by data sentinels -- but
// re-cut around what the stamps of the first generation showed: a 16-row block
// cost 6600 (forward) / 9500 (backward) cycles of which the MFMA pipe was busy for 1000, the rest
// being three workgroup barriers, an f32 partial-tile round trip through LDS (the K split over the
// waves), the wait for a tile that was only requested after the previous block's MFMA phase, and
// 7-8 small global loads per thread.
//
// Forward (lstm_scan_fwd_wide2_kernel):
//  * no K split: wave w owns ALL of K for the 4 gates x 4 hidden units 4w..4w+3 (16 MFMA columns in
//    the order unit-major, gate-minor), so a finished accumulator tile holds complete gate
//    pre-activations; a 4 x 4 transpose inside each lane quad (DPP) turns "4 rows x 1 gate" per
//    lane into "1 row x 4 gates" and the cell update runs straight on the accumulator registers:
//    no partial tiles in LDS, one barrier less;
//  * 32 rows (two row blocks) per phase against the same register-resident weights: barriers, waits
//    and stores are paid once per 32 rows;
//  * the state tile is double-buffered in LDS and the next phase's tile is requested at the TOP of a
//    phase (it was published a whole phase earlier), so it lands behind MFMA phase and epilogue;
//  * every asynchronous vector-memory operation (tile DMA, gate-input rows, table-row ids) is issued
//    through inline asm and waited for by COUNT (s_waitcnt vmcnt(n) with n = operations issued
//    since): nothing ever waits for a younger operation, in particular not for the write-through
//    publish of the phase before;
//  * gate-interleaved layouts ([row][unit][4 gates]) for the gate-input rows P, the layer-0 tables
//    and the stored gate activations G: one 16-byte load / one 8-byte LDS write per cell.
// Backward (lstm_scan_bwd_wide2_kernel): the K split stays (K = 4W: the partial tiles are small), but the
// dZ tile is double-buffered and requested a whole block ahead, the cell state is carried in a
// register from step to step (one load of C per step instead of two), the gates arrive as one
// 8-byte load, and all loads are counted the same way.
//
// Hand-off protocol, sentinel re-arming and the bounded spins are those of lstm_scan.hip.
#include <stdlib.h>
#include <string.h>

#include "kl_common.h"
#include "kl_kernels.h"

#ifndef KL_FWD_VAR
#define KL_FWD_VAR 0      /* timing experiments on the forward scan only (tools/gpu_variants.sh) */
#endif

namespace {

#define KL_STAMP_ARRAY kl_scan2_stamps
#include "kl_scan_common.h"

#include "kl_scan2_helpers.h"

// ---------------------------------------------------------------- forward
// LDS map (bytes): tile [2][NB*KSTEPS][1024] | zin [16 waves][NZ][1024] | st_g [ROWS][544] | st_c [ROWS][272] |
// pub [ROWS][144] | st_hd [ROWS][144] (P mode only) | ids [2][256] (table mode only) | flags.
// The row strides of the staging buffers are padded so that the epilogue's writes (lane = row-in-4 fastest,
// then unit) fall on distinct banks.
constexpr int F2_G_LD = 544, F2_C_LD = 272, F2_H_LD = 144;
constexpr int fwd2_lds_bytes(int ksteps, int nb, bool tab) {
  return 2 * nb * ksteps * 1024 + 16 * (tab ? 2 * nb : nb) * 1024 + 16 * nb * (F2_G_LD + F2_C_LD + F2_H_LD) +
         (tab ? 512 : 16 * nb * F2_H_LD) + 16;
}

// NB: row blocks of 16 per phase; NP: phases per workgroup and step (>= 2: while one phase computes, the other's
// publish travels); TAB: layer 0, gate inputs from the look-up tables instead of P rows.
// Nothing asynchronous ever lands in a register: tiles, gate-input pieces and table-row ids are all brought in by
// LDS-DMA and read from LDS behind a counted wait (the compiler is free to move or copy registers it believes to
// hold data, which it did with asm loads whose data had not landed yet).
template <int NB, int NP, bool TAB>
__global__ __launch_bounds__(1024, 1) void lstm_scan_fwd_wide2_kernel(const KlScanFwdWide a) {
  // (width 512 only: one tile row = one 1 KiB DMA piece, and the lane constants of the tile image -- dma_lane, frag_lane -- are
  //  written for 1 KiB rows.  A width-256 instantiation of round 2 gave a wrong recurrent gradient and was never found; the
  //  width is no template parameter any more, so no other instantiation can be built.)
  constexpr int KSTEPS = 16;
  constexpr int W = KSTEPS * 32;
  constexpr int NWG_RB = W / 64;
  constexpr int ROWS = 16 * NB;
  constexpr int NPIECE = NB * KSTEPS;            // 1 KiB pieces of a phase's tile
  constexpr int NPC = (NPIECE + 15) / 16;        // ... fetched per wave (at most)
  constexpr int NZ = TAB ? 2 * NB : NB;          // gate-input pieces per phase and wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n_rg = a.n_rg, B = a.B, T = a.T;
  // placement as in the first generation: XCD x = blockIdx % 8 hosts whole row groups (all their column groups)
  const int xcd = blockIdx.x & 7, yy = blockIdx.x >> 3;
  const int cg = yy % NWG_RB, rq = yy / NWG_RB, rg = xcd * ((n_rg + 7) >> 3) + rq;
  if (rg >= n_rg) return;
  const int u0 = cg * 64;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const zin_l = smem + 2 * NPIECE * 1024;
  unsigned char* const st_g = zin_l + 16 * NZ * 1024;
  unsigned char* const st_c = st_g + ROWS * F2_G_LD;
  unsigned char* const pub = st_c + ROWS * F2_C_LD;
  unsigned char* const st_hd = pub + ROWS * F2_H_LD;                       // (P mode)
  unsigned char* const ids_l = pub + ROWS * F2_H_LD;                       // (table mode: [2][256])
  int& ok_flag = *reinterpret_cast<int*>(pub + ROWS * F2_H_LD + (TAB ? 512 : ROWS * F2_H_LD));
  const unsigned lds_base = (unsigned)(size_t)(lds_void_t*)smem;
  const unsigned lds_zin = lds_base + (unsigned)(2 * NPIECE * 1024 + wave * NZ * 1024);      // this wave's landing slots
  const unsigned lds_ids = lds_base + (unsigned)(ids_l - smem);
  const unsigned char* const my_zin = zin_l + wave * NZ * 1024 + lane * 16;

  // cell of this lane inside a 16-row block: row 4 (lane >> 4) + (lane & 3), unit 4 wave + ((lane >> 2) & 3)
  const int jr = lane & 3, a4 = (lane >> 2) & 3, q4 = lane >> 4;
  const int crow = 4 * q4 + jr, cunit = 4 * wave + a4;

  // ---- resident weights: B fragments of the 16 columns (unit-major, gate-minor), all of K
  u32x4 bu[KSTEPS];
  {
    const int col = lane & 15;
    const long wrow = ((long)(col & 3) * W + u0 + 4 * wave + (col >> 2)) * W + (lane >> 4) * 8;
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j) bu[j] = *reinterpret_cast<const u32x4*>(a.UT + wrow + j * 32);
  }
  // cell state and dropout keep-mask of this lane's cells, per phase and block (phase 0 = the current one: rotated)
  float cst[NP][NB];
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int s = 0; s < NB; ++s) {
      const long row = (long)(rg + p * n_rg) * ROWS + s * 16 + crow;
      cst[p][s] = a.C[row * W + u0 + cunit];
      if (a.Cb) a.Cb[row * W + u0 + cunit] = f2bf(cst[p][s]);      // (block 0: the backward scan reads every c_{t-1} as bf16)
    }
  const long BW = (long)B * W;
  const __amdgpu_buffer_rsrc_t rs_h = make_rsrc(a.H, (long)(T + 1) * BW * 2);
  const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(a.C, (long)(T + 1) * BW * 4);
  const __amdgpu_buffer_rsrc_t rs_cb = make_rsrc(a.Cb, a.Cb ? (long)(T + 1) * BW * 2 : 0);
  const __amdgpu_buffer_rsrc_t rs_cnull = make_rsrc(a.C, 0);
  const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(a.G, (long)T * BW * 4 * 2);
  const __amdgpu_buffer_rsrc_t rs_hd = make_rsrc(a.Hd, a.Hd ? (long)T * BW * 2 : 0);          // zero records: stores dropped
  const __amdgpu_buffer_rsrc_t rs_ek = make_rsrc(a.EK, TAB ? (long)a.V * 4 * W * 4 : 0);
  const __amdgpu_buffer_rsrc_t rs_ck = make_rsrc(a.CtxK[0], (TAB && a.n_ctx > 0) ? (long)a.ctx_vocab * 4 * W * 4 : 0);   // (no context: zeros land)
  const __amdgpu_buffer_rsrc_t rs_id = make_rsrc(a.ids_tm, TAB ? (long)(T + 1) * B * 8 : 0);
  const bool has_ctx = TAB && a.n_ctx > 0;
  unsigned* status = a.status;
  bool alive = true;
  if (tid == 0) ok_flag = 1;
  // the compiler's own loads end here: everything it knows about has landed before the first asm operation
  // is issued, so it never places a wait of its own inside the loop
#pragma unroll
  for (int j = 0; j < KSTEPS; ++j) asm volatile("" : "+v"(bu[j]));
#pragma unroll
  for (int p = 0; p < NP; ++p)
#pragma unroll
    for (int s = 0; s < NB; ++s) asm volatile("" : "+v"(cst[p][s]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // XCD-local hand-off (opt-in): plain publishes that stay in the shared L2 + streaming tile loads, if all column groups
  // of this row group were verified to sit on this workgroup's XCD
  bool local = false;
  if (a.xcc_slots)
    local = __builtin_amdgcn_readfirstlane(
                xcd_local_group(a.xcc_slots, a.gen, NWG_RB, [&](int j) { return xcd + 8 * (rq * NWG_RB + j); }, &ok_flag + 1, status) ? 1 : 0) != 0;
  SSTAMP_INIT(0);

  // vector-memory queue of this wave: vq counts the operations issued so far, seq_x = vq right after x was
  // issued; "x has landed" = s_waitcnt vmcnt(vq - seq_x)
  int vq = 0, seq_tile[2] = {0, 0}, seq_z = 0, seq_id = 0;
  // Tile image in LDS: one 1 KiB piece = one ROW of the tile (W = 512 bf16), fetched by ONE fully coalesced DMA
  // instruction (8 whole cache lines; the fragment-order image of the first generation asked the address unit for
  // 64 different lines of which it used 16 bytes each, and the scans ran at ~10 bytes per clock and CU).  The
  // MFMA fragment reads (lane = row l & 15, k group l >> 4) would then meet on one bank group, so the 16-byte
  // chunks of row r are XOR-swizzled AT THE SOURCE: lane l fetches chunk l ^ (r & 15) and lands at position l,
  // i.e. chunk c sits at position c ^ r -- conflict-free for every 16-lane group the LDS serves together.
  static_assert(KSTEPS == 16, "one tile row = one 1 KiB DMA piece");
  const unsigned dma_lane = (unsigned)(((lane ^ (wave & 15)) * 16));
  auto issue_tile = [&](int t, int r0, int buf) __attribute__((always_inline)) {      // tile = H block t (the state before step t), rows r0 .. r0 + ROWS
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
      const int p = wave + 16 * k;
      if (p < NPIECE) arm16(smem + (buf * NPIECE + p) * 1024 + lane * 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
      const int p = wave + 16 * k;                     // row p of the phase (p & 15 == wave)
      if (p < NPIECE) {
        const unsigned soff = (unsigned)((((long)t * B + r0 + p) * W) * 2);
        if (local) glds16_nt_s(rs_h, dma_lane, soff, lds_base + (unsigned)((buf * NPIECE + p) * 1024));
        else glds16_sc1_s(rs_h, dma_lane, soff, lds_base + (unsigned)((buf * NPIECE + p) * 1024));
        ++vq;
      }
    }
    seq_tile[buf] = vq;
  };
  // lane part of the fragment address of k-step j: chunk 4 j + g of row r at position chunk ^ r, i.e.
  // frag_lane ^ (64 (j & 3)) + 256 (j >> 2) with frag_lane = 1024 r + 16 (g ^ (r & 3)) + 64 ((r >> 2) & 3)
  const unsigned frag_lane = (unsigned)((lane & 15) * 1024 + (((lane >> 4) ^ (lane & 3)) * 16) + 64 * ((lane >> 2) & 3));
  // gate-input pieces of a phase: P mode = this lane's 16 bytes (4 gates) of each block's P rows; table mode = EK row +
  // context row pieces, addressed through the ids in LDS
  const unsigned z_lane = (unsigned)((crow * 4 * W + (u0 + cunit) * 4) * 4);
  const unsigned tab_lane = (unsigned)((u0 + cunit) * 16);
  const unsigned zb_lane = (unsigned)((crow * 4 * W + (u0 + (cunit & ~1)) * 4) * 2);      // (bf16 P: the 16 bytes of units 2k, 2k + 1)
  auto issue_ids = [&](int t, int r0, int slot) __attribute__((always_inline)) {       // (wave 0: the 256 bytes of (EK, context) row offsets from row r0 of block t on)
    if (wave == 0) {
      *reinterpret_cast<unsigned*>(ids_l + slot * 256 + lane * 4) = 0xFFFFFFFFu;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      glds4_plain_s(rs_id, (unsigned)(lane * 4), (unsigned)(((long)t * B + r0) * 8), lds_ids + (unsigned)(slot * 256));
      ++vq;
      seq_id = vq;
    }
  };
  auto issue_zin = [&](int t, int r0, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NZ; ++i) arm16(const_cast<unsigned char*>(my_zin) + i * 1024);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (TAB) {
#pragma unroll
      for (int s = 0; s < NB; ++s) {
        const uint2 idv = *reinterpret_cast<const uint2*>(ids_l + slot * 256 + (s * 16 + crow) * 8);
        glds16_plain(rs_ek, idv.x + tab_lane, lds_zin + (unsigned)((2 * s) * 1024));
        ++vq;
        if (has_ctx) {      // (an out-of-range LDS-DMA writes nothing: without a context variable there is no second piece)
          glds16_plain(rs_ck, idv.y + tab_lane, lds_zin + (unsigned)((2 * s + 1) * 1024));
          ++vq;
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < NB; ++s) {
        if (a.p_bf16) {      // 8 bytes per cell: a lane fetches the 16 bytes of its pair of units and uses its half
          const __amdgpu_buffer_rsrc_t rs_p = make_rsrc(reinterpret_cast<const bf16_t*>(a.P) + ((long)t * B + r0 + s * 16) * 4 * W, (long)16 * 4 * W * 2);
          glds16_plain(rs_p, zb_lane, lds_zin + (unsigned)(s * 1024));
        } else {
          const __amdgpu_buffer_rsrc_t rs_p = make_rsrc(a.P + ((long)t * B + r0 + s * 16) * 4 * W, (long)16 * 4 * W * 4);
          glds16_plain(rs_p, z_lane, lds_zin + (unsigned)(s * 1024));
        }
        ++vq;
      }
    }
    seq_z = vq;
  };

  // all pieces of this wave's gate inputs have landed (checked, not assumed: see arm16)
  auto zin_landed = [&]() __attribute__((always_inline)) {
    bool ok = false;
    for (unsigned spin = 0; spin < SPIN_LIMIT; ++spin) {
      bool v = true;
#pragma unroll
      for (int i = 0; i < NZ; ++i)
        if (!TAB || (i & 1) == 0 || has_ctx) v = v && landed16(my_zin + i * 1024);
      if (__all(v)) { ok = true; break; }
#ifdef KL_STAMP
      if (blockIdx.x == STAMP_WG && threadIdx.x == 0) stamp_lds[13] += 1;
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!ok) { __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok_flag = 0; }
  };

  // ---- prologue: tile and gate inputs of the first phase; (tables) ids of the first two phases
  {
    const int r0 = rg * ROWS;
    issue_tile(0, r0, 0);
    if (a.pf_mode == 2) issue_tile(0, (rg + n_rg) * ROWS, 1);      // (NP >= 2: the second phase is another row group of step 0)
    if (TAB) {
      issue_ids(0, r0, 0);
      issue_ids(0, (rg + n_rg) * ROWS, 1);      // (NP >= 2: the second phase is another row group of step 0)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    issue_zin(0, r0, 0);
  }

  int n = 0;
  for (int t = 0; t < T; ++t) {
#pragma unroll 1
    for (int ip = 0; ip < NP; ++ip, ++n) {
      const int buf = n & 1;
      const int r0 = (rg + ip * n_rg) * ROWS;
      // the phase after this one, and the one after that
      int t1 = t, ip1 = ip + 1;
      if (ip1 >= NP) { ip1 = 0; t1 = t + 1; }
      int t2 = t1, ip2 = ip1 + 1;
      if (ip2 >= NP) { ip2 = 0; t2 = t1 + 1; }
      const int r1 = (rg + ip1 * n_rg) * ROWS, r2 = (rg + ip2 * n_rg) * ROWS;
      SSTAMP(0);
      // dropout keep-masks of this phase's cells (constant over the steps, served by L2): asm loads of this iteration,
      // consumed in the epilogue (no younger stores between: the count is exact)
      unsigned mkin[NB];
      int seq_mk = 0;
#pragma unroll
      for (int s = 0; s < NB; ++s) mkin[s] = 0x3f800000u;      // (1.0f)
      if (!TAB && a.mask) {
#pragma unroll
        for (int s = 0; s < NB; ++s) {
          aload4_glb(mkin[s], a.mask + ((long)r0 + s * 16) * W, (unsigned)((crow * W + u0 + cunit) * 4));
          ++vq;
        }
        seq_mk = vq;
      }
      // ---- request the next phase's tile (its rows were published a whole phase ago), then wait for this one's
      // Where the next tile is requested (a.pf_mode): 0 = here, a phase ahead; 1 = behind this phase's MFMAs; 2 = TWO
      // phases ahead, behind the barrier after the MFMAs of the phase that last read its buffer (NP >= 3: the rows were
      // published at least a phase before) -- the request then also sits in front of this phase's stores in the queue.
      if (a.pf_mode == 0 && alive && t1 < T) issue_tile(t1, r1, buf ^ 1);
      if (alive) {
        wait_vm(vq - seq_tile[buf]);
        SSTAMP(8);
        bool ok = true;
#pragma unroll
        for (int k = 0; k < NPC; ++k) {
          const int p = wave + 16 * k;
          if (p < NPIECE) ok = ok && piece_there(smem + (buf * NPIECE + p) * 1024 + lane * 16);
        }
        ok = __all(ok);
        if (!ok) {
          // a producer was late: re-fetch this wave's pieces until all their granules are there (bounded)
#ifdef KL_STAMP
          if (blockIdx.x == STAMP_WG && threadIdx.x == 0) stamp_lds[12] += 1;
#endif
          for (unsigned spin = 0; spin < SPIN_LIMIT && !ok; ++spin) {
#pragma unroll
            for (int k = 0; k < NPC; ++k) {
              const int p = wave + 16 * k;
              if (p < NPIECE && spin > 0) {       // (first round: only wait until everything issued has landed)
                const unsigned soff = (unsigned)((((long)t * B + r0 + p) * W) * 2);
                if (local) glds16_nt_s(rs_h, dma_lane, soff, lds_base + (unsigned)((buf * NPIECE + p) * 1024));
                else glds16_sc1_s(rs_h, dma_lane, soff, lds_base + (unsigned)((buf * NPIECE + p) * 1024));
                ++vq;
              }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ok = true;
#pragma unroll
            for (int k = 0; k < NPC; ++k) {
              const int p = wave + 16 * k;
              if (p < NPIECE) ok = ok && piece_there(smem + (buf * NPIECE + p) * 1024 + lane * 16);
            }
            ok = __all(ok);
            if (!ok) {
              if ((spin & 63) == 63 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
              __builtin_amdgcn_s_sleep(2);
            }
          }
          if (!ok) {
            __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok_flag = 0;
          }
        }
      }
      if (TAB && wave == 0) {      // the ids the waves will read behind the next barrier have landed (slot of phase n + 1)
        wait_vm(vq - seq_id);
        if (t1 < T) {
          bool ok = false;
          for (unsigned spin = 0; spin < SPIN_LIMIT; ++spin) {
            asm volatile("" ::: "memory");
            const unsigned v = *reinterpret_cast<const unsigned*>(ids_l + ((n + 1) & 1) * 256 + lane * 4);
            if (__all(v != 0xFFFFFFFFu || lane * 4 >= ROWS * 8)) { ok = true; break; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          }
          if (!ok) { __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok_flag = 0; }
        }
      }
      SSTAMP(1);
      __syncthreads();
      SSTAMP(2);
      alive = __builtin_amdgcn_readfirstlane(ok_flag) != 0;
      // ---- MFMA phase: NB row blocks against the resident weights, all of K
      // The gate inputs of this phase (requested a whole phase ago) start the accumulators, in the MFMA's own layout (the
      // quad transpose is its own inverse); their landing slots are free again at once, so the next phase's pieces are
      // requested here, a whole phase ahead (table mode: through the ids in LDS; wave 0 then fetches the ids of the
      // phase after that)
      f32x4 acc[NB];
      wait_vm(vq - seq_z);
      zin_landed();
#pragma unroll
      for (int s = 0; s < NB; ++s) {
        if (TAB) {
          acc[s] = *reinterpret_cast<const f32x4*>(my_zin + (2 * s) * 1024);
          if (has_ctx) acc[s] += *reinterpret_cast<const f32x4*>(my_zin + (2 * s + 1) * 1024);
        } else if (a.p_bf16) {
          const uint2 pb = *reinterpret_cast<const uint2*>(my_zin + s * 1024 + (cunit & 1) * 8);
          acc[s] = f32x4{u2f(pb.x << 16), u2f(pb.x & 0xffff0000u), u2f(pb.y << 16), u2f(pb.y & 0xffff0000u)};
        } else {
          acc[s] = *reinterpret_cast<const f32x4*>(my_zin + s * 1024);
        }
        quad_transpose(acc[s], jr);
      }
      asm volatile("" : "+v"(acc[0]));
      if (NB > 1) asm volatile("" : "+v"(acc[NB - 1]));
      if (t1 < T) issue_zin(t1, r1, (n + 1) & 1);
      if (TAB && t2 < T) issue_ids(t2, r2, n & 1);      // (slot of this phase's ids: every wave read them a phase ago)
      {
        // k-steps in the order j = 4 (q & 3) + (q >> 2), q = 0..15: one lane address per group of four; the fragments of
        // step q + 1 are requested before the MFMAs of step q (pinned: left alone, the compiler serialised read -> wait ->
        // MFMA on a single fragment buffer)
        const unsigned char* tb = smem + buf * NPIECE * 1024;
        u32x4 fr[2][NB];
#pragma unroll
        for (int s = 0; s < NB; ++s) fr[0][s] = *reinterpret_cast<const u32x4*>(tb + frag_lane + s * 16 * 1024);
#pragma unroll
        for (int q = 0; q < KSTEPS; ++q) {
#if KL_FWD_VAR & 1       /* timing experiments only (tools/gpu_variants.sh): 1 = no MFMA phase, 2 = half the fragment reads, 4 = half the MFMAs */
          continue;
#endif
          if (q + 1 < KSTEPS && !((KL_FWD_VAR & 2) && (q & 1))) {
            const int q1 = q + 1;
            const unsigned char* ap = tb + (frag_lane ^ (unsigned)(64 * (q1 >> 2))) + 256 * (q1 & 3);
#pragma unroll
            for (int s = 0; s < NB; ++s) fr[q1 & 1][s] = *reinterpret_cast<const u32x4*>(ap + s * 16 * 1024);
          }
          __builtin_amdgcn_sched_barrier(0);
          const bf16x8 fb = __builtin_bit_cast(bf16x8, bu[4 * (q & 3) + (q >> 2)]);
          if (!((KL_FWD_VAR & 4) && (q & 1))) {
#pragma unroll
            for (int s = 0; s < NB; ++s) acc[s] = mfma16(__builtin_bit_cast(bf16x8, fr[q & 1][s]), fb, acc[s]);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      SSTAMP(3);
      if (a.pf_mode == 1 && alive && t1 < T) issue_tile(t1, r1, buf ^ 1);     // (its buffer was last read a phase ago)
      // ---- epilogue on the accumulators: lane = (row, unit), registers = gates
#pragma unroll
      for (int s = 0; s < NB; ++s) quad_transpose(acc[s], jr);
      float z[NB][4];
#pragma unroll
      for (int s = 0; s < NB; ++s)
#pragma unroll
        for (int g = 0; g < 4; ++g) z[s][g] = acc[s][g];
      SSTAMP(4);
      if (!TAB && a.mask) {
        wait_vm(vq - seq_mk);
#pragma unroll
        for (int s = 0; s < NB; ++s) use_regs(mkin[s]);
      }
#pragma unroll
      for (int s = 0; s < NB; ++s) {
        const float gi = fast_sigmoid(z[s][0]), gf = fast_sigmoid(z[s][1]), gg = fast_tanh(z[s][2]), go = fast_sigmoid(z[s][3]);
        const float c = gf * cst[0][s] + gi * gg;
        cst[0][s] = c;
        const float h = go * fast_tanh(c);
        const int row = s * 16 + crow;
        *reinterpret_cast<bf16_t*>(pub + row * F2_H_LD + cunit * 2) = f2bf(h);
        if (!TAB) *reinterpret_cast<bf16_t*>(st_hd + row * F2_H_LD + cunit * 2) = f2bf(h * u2f(mkin[s]));
        *reinterpret_cast<float*>(st_c + row * F2_C_LD + cunit * 4) = c;
        uint2 gp;
        gp.x = (unsigned)f2bf(gi) | ((unsigned)f2bf(gf) << 16);
        gp.y = (unsigned)f2bf(gg) | ((unsigned)f2bf(go) << 16);
        *reinterpret_cast<uint2*>(st_g + row * F2_G_LD + cunit * 8) = gp;
      }
      SSTAMP(5);
      __syncthreads();
      SSTAMP(6);
      if (a.pf_mode == 2 && alive && t2 < T) issue_tile(t2, r2, buf);       // (every wave has finished this phase's MFMAs)
      // ---- stores: the publish first (write-through, whole 128-byte lines per row), then what only later launches read
      {
        int stid = tid;
        asm volatile("" : "+v"(stid));
        const unsigned trow = (unsigned)(t * B + r0);          // first time-major row of this phase
        if (wave < 2 * NB) {       // h[t] -> H block t + 1: ROWS x 8 pieces of 8 units
          const int prow = stid >> 3, seg = stid & 7;
          const uint4 v = *reinterpret_cast<const uint4*>(pub + prow * F2_H_LD + seg * 16);
          const unsigned off = (unsigned)((prow * W + seg * 8) * 2) + ((trow + B) * W + u0) * 2u;
          if (!alive) store16(make_rsrc(a.H, 0), 0u, 0u, v);
          else if (local) store16(rs_h, off, 0u, v);        // stays in this XCD's L2, where all its readers are
          else store16_sc1(rs_h, off, v);
          ++vq;
          if (a.sentinel == 2) {
            // rolling sentinels (as the backward scan's): the same lanes arm block t + 3 while they publish block t + 1, so only
            // blocks 1 and 2 are pre-filled in front of the launch instead of all T (0.8 GB per layer at the bench shape).
            // The arming store and the publish of block t + 3 two steps later come from the same lanes to the same
            // addresses, i.e. in order; nobody reads block t + 3 before it has been published.
            const uint4 ones = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            const unsigned off3 = off + (unsigned)((long)2 * B * W * 2);
            if (!alive || t + 3 > T) store16(make_rsrc(a.H, 0), 0u, 0u, ones);
            else if (local) store16(rs_h, off3, 0u, ones);
            else store16_sc1(rs_h, off3, ones);
            ++vq;
          }
        }
        if (NB == 2 || wave >= 2) {
          // gates: ROWS x 32 pieces (2 units x 4 gates); NB == 1: waves 2..9
          const int q = NB == 2 ? stid : stid - 128;
          if (NB == 2 || q < 512) {
            const int prow = q >> 5, seg = q & 31;
            store16(rs_g, (unsigned)((prow * W * 4 + seg * 8) * 2), (trow * W + u0) * 8u,
                    *reinterpret_cast<const uint4*>(st_g + prow * F2_G_LD + seg * 16));
            ++vq;
          }
        }
        {
          // cell state (ROWS x 16 pieces of 4 units) and masked outputs (ROWS x 8 pieces of 8 units)
          const int cq = NB == 2 ? stid - 256 : stid - 640;          // NB == 2: waves 4..11, NB == 1: waves 10..13
          const int hq = NB == 2 ? stid - 768 : stid - 896;          // NB == 2: waves 12..15, NB == 1: waves 14..15
          const int wv_c0 = NB == 2 ? 4 : 10, wv_c1 = NB == 2 ? 12 : 14;
          if (wave >= wv_c0 && wave < wv_c1) {
            const int prow = cq >> 4, seg = cq & 15;
            const uint4 cv = *reinterpret_cast<const uint4*>(st_c + prow * F2_C_LD + seg * 16);
            if (a.Cb) {
              // the backward scan reads bf16 cell states (half the bytes); the carried-out state (block T) stays f32 in C
              uint2 cb;
              cb.x = (unsigned)f2bf(u2f(cv.x)) | ((unsigned)f2bf(u2f(cv.y)) << 16);
              cb.y = (unsigned)f2bf(u2f(cv.z)) | ((unsigned)f2bf(u2f(cv.w)) << 16);
              __builtin_amdgcn_raw_buffer_store_b64(u32x2{cb.x, cb.y}, rs_cb, (int)((prow * W + seg * 4) * 2), (int)(((trow + B) * W + u0) * 2u), 0);
              store16(t == T - 1 ? rs_c : rs_cnull, (unsigned)((prow * W + seg * 4) * 4), ((trow + B) * W + u0) * 4u, cv);
              vq += 2;
            } else {
              store16(rs_c, (unsigned)((prow * W + seg * 4) * 4), ((trow + B) * W + u0) * 4u, cv);
              ++vq;
            }
          } else if (!TAB && wave >= wv_c1) {
            const int prow = hq >> 3, seg = hq & 7;
            store16(rs_hd, (unsigned)((prow * W + seg * 8) * 2), (trow * W + u0) * 2u,
                    *reinterpret_cast<const uint4*>(st_hd + prow * F2_H_LD + seg * 16));
            ++vq;
          }
        }
      }
      SSTAMP(7);
      // the next phase's cells move to slot 0
      if (NP > 1) {
#pragma unroll
        for (int s = 0; s < NB; ++s) {
          const float c0 = cst[0][s];
#pragma unroll
          for (int p = 0; p + 1 < NP; ++p) cst[p][s] = cst[p + 1][s];
          cst[NP - 1][s] = c0;
        }
      }
    }
  }
  SSTAMP_FLUSH();
}

}  // namespace

namespace {

// ---------------------------------------------------------------- gate inputs of the layers above the first
// P[r][4 unit + gate] = X[r][:] . KTp[4 unit + gate][:] + bp  for all T*B rows (width 512), WEIGHT-STATIONARY: the
// 2 MiB of KTp are spread over the register files exactly as the scans spread U (wave w of column group cg keeps its
// 16 output columns for all of K in 64 VGPRs), and only X streams -- 32-row tiles, double-buffered in LDS, fetched through
// registers two tiles ahead (see the kernel).  The ring GEMM
// (gemm.hip) pulls BOTH operands of every 256 x 128 tile from L2 again and again and is bound by that traffic at
// K = 512 (0.65 PFLOP/s, 0.79 with its stores removed); here a CU reads 32 KiB per 8.4 MFLOP instead of 48 KiB per
// 4.2, and nothing depends on anything.
// Grid as the scans': 8 column groups x n_rg row groups, block b on XCD b % 8 so that the eight workgroups reading the
// same rows share an L2; row group rg takes the tiles rg, rg + n_rg, ...  One barrier per tile.
struct KlProjWs {
  const bf16_t* X;       // [M][512]
  const bf16_t* KTp;     // [2048][512], rows in output-column order
  const float* bp;       // [2048]
  bf16_t* P;             // [M][2048]
  int M;                 // rows, a multiple of 32
  int n_rg;              // row groups (workgroups per column group)
  unsigned* status;
};

__global__ __launch_bounds__(1024, 1) void proj_ws_kernel(const KlProjWs a) {
  constexpr int KSTEPS = 16, W = 512, N = 4 * W, NCG = N / 256, ROWS = 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, yy = blockIdx.x >> 3;
  const int cg = yy % NCG, rq = yy / NCG, rg = xcd * ((a.n_rg + 7) >> 3) + rq;
  if (rg >= a.n_rg) return;
  const int c0 = cg * 256;                                  // first output column of this workgroup
  const int n_tiles_all = a.M / ROWS;
  const int my_tiles = rg < n_tiles_all ? (n_tiles_all - rg + a.n_rg - 1) / a.n_rg : 0;
  if (my_tiles == 0) return;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // tile [2][32 rows][1024]

  u32x4 bu[KSTEPS];
  {
    const long wrow = (long)(c0 + 16 * wave + (lane & 15)) * W + (lane >> 4) * 8;
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j) bu[j] = *reinterpret_cast<const u32x4*>(a.KTp + wrow + j * 32);
  }
  const float bias = a.bp[c0 + 16 * wave + (lane & 15)];    // (accumulator layout: one column per lane)
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(a.X, (long)a.M * W * 2);
  const __amdgpu_buffer_rsrc_t rs_p = make_rsrc(a.P, (long)a.M * N * 2);

  // X rows come through REGISTERS, two tiles ahead, and are laid into the LDS tile by the wave that fetched them (rows
  // w and w + 16, their 16-byte chunks XOR-swizzled on the way in: chunk c of row r at position c ^ (r & 15), as in the
  // scans).  An LDS-DMA piece costs the CU ~140 cycles of a resource all waves share (32 pieces per tile were 4,500 of this
  // kernel's 4,800 cycles per tile, with or without the MFMAs); a 1 KiB register load passes the address unit in 16.
  // Everything here is visible to the compiler (no hand-issued asynchronous operation), so its own waits are exact.
  const unsigned frag_lane = (unsigned)((lane & 15) * 1024 + (((lane >> 4) ^ (lane & 3)) * 16) + 64 * ((lane >> 2) & 3));
  const unsigned put_lane = (unsigned)(wave * 1024 + ((lane ^ wave) & 63) * 16);      // position of this lane's chunk in row w (and w + 16)
  const int jr = lane & 3, a4 = (lane >> 2) & 3, q4 = lane >> 4;
  auto fetch = [&](int i, u32x4 (&r)[2]) __attribute__((always_inline)) {
    const long row0 = (long)(rg + (long)i * a.n_rg) * ROWS;
#pragma unroll
    for (int h = 0; h < 2; ++h)
      r[h] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, lane * 16, (int)(unsigned)((row0 + h * 16 + wave) * W * 2), 0));
  };
  auto put = [&](int buf, const u32x4 (&r)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) *reinterpret_cast<u32x4*>(smem + buf * ROWS * 1024 + h * 16 * 1024 + put_lane) = r[h];
  };
  u32x4 ra[2], rb[2];      // tile i + 1 and tile i + 2 on their way
  fetch(0, ra);
  put(0, ra);
  if (my_tiles > 1) fetch(1, ra);
  if (my_tiles > 2) fetch(2, rb);
  // one tile: r1 holds tile i + 1 (fetched two tiles ago) and takes tile i + 3 once it has been laid into LDS; the two
  // register sets swap roles from tile to tile (loop unrolled by two: no moves, which would wait for the younger load)
  auto one_tile = [&](int i, u32x4 (&r1)[2]) __attribute__((always_inline)) {
    const int buf = i & 1;
    __syncthreads();                       // tile i is complete in LDS, and every wave has left tile i - 1
    if (i + 1 < my_tiles) put(buf ^ 1, r1);
    if (i + 3 < my_tiles) fetch(i + 3, r1);
    f32x4 acc[2] = {f32x4{bias, bias, bias, bias}, f32x4{bias, bias, bias, bias}};
    {
      const unsigned char* tb = smem + buf * ROWS * 1024;
      u32x4 fr[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) fr[0][h] = *reinterpret_cast<const u32x4*>(tb + frag_lane + h * 16 * 1024);
#pragma unroll
      for (int q = 0; q < KSTEPS; ++q) {
        if (q + 1 < KSTEPS) {
          const int q1 = q + 1;
          const unsigned char* ap = tb + (frag_lane ^ (unsigned)(64 * (q1 >> 2))) + 256 * (q1 & 3);
#pragma unroll
          for (int h = 0; h < 2; ++h) fr[q1 & 1][h] = *reinterpret_cast<const u32x4*>(ap + h * 16 * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 fb = __builtin_bit_cast(bf16x8, bu[4 * (q & 3) + (q >> 2)]);
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[h] = mfma16(__builtin_bit_cast(bf16x8, fr[q & 1][h]), fb, acc[h]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // out: after the quad transpose lane = (row 4 q4 + jr, columns 4 a4 .. 4 a4 + 3 of this wave's 16): 8 bytes per lane
    const long row0 = (long)(rg + (long)i * a.n_rg) * ROWS;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 v = acc[h];
      quad_transpose(v, jr);
      const u32x2 pk = u32x2{(unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16), (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16)};
      const unsigned off = (unsigned)(((h * 16 + 4 * q4 + jr) * N + c0 + 16 * wave + 4 * a4) * 2);
      __builtin_amdgcn_raw_buffer_store_b64(pk, rs_p, (int)off, (int)(unsigned)(row0 * N * 2), 0);
    }
  };
  for (int i = 0; i < my_tiles; i += 2) {
    one_tile(i, ra);
    if (i + 1 < my_tiles) one_tile(i + 1, rb);
  }
}

// The same with EIGHT waves of 32 output columns each (round 4): every A fragment read from LDS feeds TWO MFMAs (column tiles c = 0, 1),
// so the tile's fragments cross the LDS eight times per tile instead of sixteen -- the 16-wave form above spends 5 000 cycles on a
// 32-row tile whose MFMAs take 2 048 (16 waves x 32 KiB of ds_read_b128 per tile: LDS-read-bound, as the 16-wave forward scan).
// Column tile c's column j is output column 32 w + 8 (j >> 2) + 4 c + (j & 3): after the quad transpose a lane holds eight
// CONSECUTIVE columns of one row over the two tiles -- one 16-byte store, 64 bytes per row and wave.
// STAGE: the tile's 32 x 512 bytes of output leave through 16 KiB of LDS as whole 512-byte row segments (32 consecutive lanes on one
// row) instead of 64 bytes per row and wave.
template <bool STAGE>
__global__ __launch_bounds__(512, 1) void proj_ws8_kernel(const KlProjWs a) {
  constexpr int KSTEPS = 16, W = 512, N = 4 * W, NCG = N / 256, ROWS = 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, yy = blockIdx.x >> 3;
  const int cg = yy % NCG, rq = yy / NCG, rg = xcd * ((a.n_rg + 7) >> 3) + rq;
  if (rg >= a.n_rg) return;
  const int c0 = cg * 256;                                  // first output column of this workgroup
  const int n_tiles_all = a.M / ROWS;
  const int my_tiles = rg < n_tiles_all ? (n_tiles_all - rg + a.n_rg - 1) / a.n_rg : 0;
  if (my_tiles == 0) return;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // tile [2][32 rows][1024]

  u32x4 bu[2][KSTEPS];
  float bias[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int col = c0 + 32 * wave + 8 * ((lane & 15) >> 2) + 4 * c + (lane & 3);
    const long wrow = (long)col * W + (lane >> 4) * 8;
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j) bu[c][j] = *reinterpret_cast<const u32x4*>(a.KTp + wrow + j * 32);
    bias[c] = a.bp[col];                                    // (accumulator layout: one column per lane)
  }
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(a.X, (long)a.M * W * 2);
  const __amdgpu_buffer_rsrc_t rs_p = make_rsrc(a.P, (long)a.M * N * 2);
  // X rows through registers two tiles ahead, laid into the LDS tile by the wave that fetched them: rows w, w + 8, w + 16, w + 24,
  // chunk c of row r at position c ^ (r & 15) -- the layout the 16-wave form reads
  const unsigned frag_lane = (unsigned)((lane & 15) * 1024 + (((lane >> 4) ^ (lane & 3)) * 16) + 64 * ((lane >> 2) & 3));
  unsigned put_lane[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) put_lane[h] = (unsigned)((wave + 8 * h) * 1024 + ((lane ^ ((wave + 8 * h) & 15)) & 63) * 16);
  const int jr = lane & 3, a4 = (lane >> 2) & 3, q4 = lane >> 4;
  auto fetch = [&](int i, u32x4 (&r)[4]) __attribute__((always_inline)) {
    const long row0 = (long)(rg + (long)i * a.n_rg) * ROWS;
#pragma unroll
    for (int h = 0; h < 4; ++h)
      r[h] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, lane * 16, (int)(unsigned)((row0 + 8 * h + wave) * W * 2), 0));
  };
  auto put = [&](int buf, const u32x4 (&r)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 4; ++h) *reinterpret_cast<u32x4*>(smem + buf * ROWS * 1024 + put_lane[h]) = r[h];
  };
  u32x4 ra[4], rb[4];      // tile i + 1 and tile i + 2 on their way
  fetch(0, ra);
  put(0, ra);
  if (my_tiles > 1) fetch(1, ra);
  if (my_tiles > 2) fetch(2, rb);
  auto one_tile = [&](int i, u32x4 (&r1)[4]) __attribute__((always_inline)) {
    const int buf = i & 1;
    __syncthreads();                       // tile i is complete in LDS, and every wave has left tile i - 1
    if (i + 1 < my_tiles) put(buf ^ 1, r1);
    if (i + 3 < my_tiles) fetch(i + 3, r1);
    f32x4 acc[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int h = 0; h < 2; ++h) acc[c][h] = f32x4{bias[c], bias[c], bias[c], bias[c]};
    {
      const unsigned char* tb = smem + buf * ROWS * 1024;
      u32x4 fr[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) fr[0][h] = *reinterpret_cast<const u32x4*>(tb + frag_lane + h * 16 * 1024);
#pragma unroll
      for (int q = 0; q < KSTEPS; ++q) {
        if (q + 1 < KSTEPS) {
          const int q1 = q + 1;
          const unsigned char* ap = tb + (frag_lane ^ (unsigned)(64 * (q1 >> 2))) + 256 * (q1 & 3);
#pragma unroll
          for (int h = 0; h < 2; ++h) fr[q1 & 1][h] = *reinterpret_cast<const u32x4*>(ap + h * 16 * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const bf16x8 fb = __builtin_bit_cast(bf16x8, bu[c][4 * (q & 3) + (q >> 2)]);
#pragma unroll
          for (int h = 0; h < 2; ++h) acc[c][h] = mfma16(__builtin_bit_cast(bf16x8, fr[q & 1][h]), fb, acc[c][h]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // out: after the quad transposes lane = (row 4 q4 + jr, columns 32 w + 8 a4 .. + 7): 16 bytes per lane
    const long row0 = (long)(rg + (long)i * a.n_rg) * ROWS;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 v0 = acc[0][h], v1 = acc[1][h];
      quad_transpose(v0, jr);
      quad_transpose(v1, jr);
      const u32x4 pk = u32x4{(unsigned)f2bf(v0[0]) | ((unsigned)f2bf(v0[1]) << 16), (unsigned)f2bf(v0[2]) | ((unsigned)f2bf(v0[3]) << 16),
                             (unsigned)f2bf(v1[0]) | ((unsigned)f2bf(v1[1]) << 16), (unsigned)f2bf(v1[2]) | ((unsigned)f2bf(v1[3]) << 16)};
      if (STAGE) {
        *reinterpret_cast<u32x4*>(smem + 2 * ROWS * 1024 + (h * 16 + 4 * q4 + jr) * 528 + (32 * wave + 8 * a4) * 2) = pk;
      } else {
        const unsigned off = (unsigned)(((h * 16 + 4 * q4 + jr) * N + c0 + 32 * wave + 8 * a4) * 2);
        __builtin_amdgcn_raw_buffer_store_b128(pk, rs_p, (int)off, (int)(unsigned)(row0 * N * 2), 0);
      }
    }
    if (STAGE) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int row = (tid >> 5) + 16 * k, seg = tid & 31;
        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + 2 * ROWS * 1024 + row * 528 + seg * 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs_p, (int)(unsigned)((row * N + c0 + seg * 8) * 2), (int)(unsigned)(row0 * N * 2), 0);
      }
    }
  };
  for (int i = 0; i < my_tiles; i += 2) {
    one_tile(i, ra);
    if (i + 1 < my_tiles) one_tile(i + 1, rb);
  }
}

// (wave_max / wave_sum / wave_min_i: kl_scan2_helpers.h)

// ---------------------------------------------------------------- gradient into the top layer: dH = dlogits . E
// dH[r][:] = dlogits[r][:256] . ET[:][:256]^T as bf16, weight-stationary like proj_ws_kernel: ET (512 x 256 bf16) lies in the
// registers of two column groups (wave w: 16 output columns, K = 256 in 32 VGPRs), dlogits rows (512 bytes) stream through
// a double-buffered 32-row tile, fetched through registers two tiles ahead -- one 1 KiB load per wave and tile covers two
// rows.  Chunk c (16 bytes) of tile row r sits at position c ^ (r & 15) of the row, so that the sixteen rows a fragment
// read touches fall on different bank groups.
struct KlDhWs {
  const bf16_t* X;       // [M][256] dlogits
  const bf16_t* ET;      // [512][256]
  bf16_t* dH;            // [M][512]
  int M, n_rg;
};

__global__ __launch_bounds__(1024, 1) void dh_ws_kernel(const KlDhWs a) {
  constexpr int KSTEPS = 8, K = 256, N = 512, NCG = N / 256, ROWS = 32, RB = K * 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, yy = blockIdx.x >> 3;
  const int cg = yy % NCG, rq = yy / NCG, rg = xcd * ((a.n_rg + 7) >> 3) + rq;
  if (rg >= a.n_rg) return;
  const int c0 = cg * 256;
  const int n_tiles_all = a.M / ROWS;
  const int my_tiles = rg < n_tiles_all ? (n_tiles_all - rg + a.n_rg - 1) / a.n_rg : 0;
  if (my_tiles == 0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // tile [2][32 rows][512]

  u32x4 bu[KSTEPS];
  {
    const long wrow = (long)(c0 + 16 * wave + (lane & 15)) * K + (lane >> 4) * 8;
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j) bu[j] = *reinterpret_cast<const u32x4*>(a.ET + wrow + j * 32);
  }
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(a.X, (long)a.M * K * 2);
  const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(a.dH, (long)a.M * N * 2);
  // this wave's load covers rows 2 w and 2 w + 1 of a tile: lane -> (row, chunk)
  const int prow = 2 * wave + (lane >> 5), pchunk = lane & 31;
  const unsigned put_lane = (unsigned)(prow * RB + ((pchunk ^ (prow & 15)) * 16));
  const int frow = lane & 15, kg = lane >> 4;      // fragment row (+ 16 h) and k group of this lane
  const int jr = lane & 3, a4 = (lane >> 2) & 3, q4 = lane >> 4;
  auto fetch = [&](int i, u32x4& r) __attribute__((always_inline)) {
    const long row0 = (long)(rg + (long)i * a.n_rg) * ROWS;
    r = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, lane * 16, (int)(unsigned)((row0 + 2 * wave) * RB), 0));
  };
  auto put = [&](int buf, const u32x4& r) __attribute__((always_inline)) {
    *reinterpret_cast<u32x4*>(smem + buf * ROWS * RB + put_lane) = r;
  };
  u32x4 ra, rb;
  fetch(0, ra);
  put(0, ra);
  if (my_tiles > 1) fetch(1, ra);
  if (my_tiles > 2) fetch(2, rb);
  auto one_tile = [&](int i, u32x4& r1) __attribute__((always_inline)) {
    const int buf = i & 1;
    __syncthreads();
    if (i + 1 < my_tiles) put(buf ^ 1, r1);
    if (i + 3 < my_tiles) fetch(i + 3, r1);
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const unsigned char* tb = smem + buf * ROWS * RB;
#pragma unroll
    for (int q = 0; q < KSTEPS; ++q) {
      u32x4 fr[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) fr[h] = *reinterpret_cast<const u32x4*>(tb + (frow + 16 * h) * RB + (((4 * q + kg) ^ frow) * 16));
      const bf16x8 fb = __builtin_bit_cast(bf16x8, bu[q]);
#pragma unroll
      for (int h = 0; h < 2; ++h) acc[h] = mfma16(__builtin_bit_cast(bf16x8, fr[h]), fb, acc[h]);
    }
    const long row0 = (long)(rg + (long)i * a.n_rg) * ROWS;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 v = acc[h];
      quad_transpose(v, jr);
      const u32x2 pk = u32x2{(unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16), (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16)};
      const unsigned off = (unsigned)(((h * 16 + 4 * q4 + jr) * N + c0 + 16 * wave + 4 * a4) * 2);
      __builtin_amdgcn_raw_buffer_store_b64(pk, rs_o, (int)off, (int)(unsigned)(row0 * N * 2), 0);
    }
  };
  for (int i = 0; i < my_tiles; i += 2) {
    one_tile(i, ra);
    if (i + 1 < my_tiles) one_tile(i + 1, rb);
  }
}

// ---------------------------------------------------------------- output projection + softmax + cross-entropy + its gradient
// Training, V = 256, width 512: logits = X . E^T for 32 rows at a time with E (256 x 512 bf16 = the register files of one
// workgroup) stationary as in proj_ws_kernel, the tile's logits turned through LDS so that each wave then owns two whole
// rows (lane = four consecutive characters, as softmax_ce_v256_kernel), and out go only the bf16 gradient rows
// dlogits = (p - onehot(target)) / count and the per-row (loss, hit) pair.  The logits never reach memory: the GEMM +
// softmax pair wrote 0.8 GB of f32 and read it back at the bench shape.  Same rules as the kernel it replaces
// (rating.py:255-258 through Keras: probabilities clipped to [1e-7, 1 - 1e-7] -- no gradient outside --, padded
// positions count in the mean but carry no target, accuracy = first maximum equals the target).
struct KlLogitsCe {
  const bf16_t* X;       // [M][512] (masked) outputs of the top layer, time-major rows r = t B + b
  const bf16_t* E;       // [256][512]
  const int* tgt;        // [B][T]
  bf16_t* dlogits;       // [M][256]
  float* rowstat;        // [M][2]
  int M, B, T, n_rg, last_only;
  float inv_count;
};

__global__ __launch_bounds__(1024, 1) void logits_ce_ws_kernel(const KlLogitsCe a) {
  constexpr int KSTEPS = 16, W = 512, V = 256, ROWS = 32, LDZ = V + 4, OFF_Z = 2 * ROWS * 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = blockIdx.x;
  const int n_tiles_all = a.M / ROWS;
  const int my_tiles = rg < n_tiles_all ? (n_tiles_all - rg + a.n_rg - 1) / a.n_rg : 0;
  if (my_tiles == 0) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // tile [2][32][1024] | logits [32][260] f32
  float* zl = reinterpret_cast<float*>(smem + OFF_Z);

  u32x4 bu[KSTEPS];
  {
    const long wrow = (long)(16 * wave + (lane & 15)) * W + (lane >> 4) * 8;
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j) bu[j] = *reinterpret_cast<const u32x4*>(a.E + wrow + j * 32);
  }
  const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(a.X, (long)a.M * W * 2);
  const __amdgpu_buffer_rsrc_t rs_dl = make_rsrc(a.dlogits, (long)a.M * V * 2);
  const __amdgpu_buffer_rsrc_t rs_rs = make_rsrc(a.rowstat, (long)a.M * 8);
  const unsigned frag_lane = (unsigned)((lane & 15) * 1024 + (((lane >> 4) ^ (lane & 3)) * 16) + 64 * ((lane >> 2) & 3));
  const unsigned put_lane = (unsigned)(wave * 1024 + ((lane ^ wave) & 63) * 16);
  auto fetch = [&](int i, u32x4 (&r)[2]) __attribute__((always_inline)) {
    const long row0 = (long)(rg + (long)i * a.n_rg) * ROWS;
#pragma unroll
    for (int h = 0; h < 2; ++h)
      r[h] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, lane * 16, (int)(unsigned)((row0 + h * 16 + wave) * W * 2), 0));
  };
  auto put = [&](int buf, const u32x4 (&r)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) *reinterpret_cast<u32x4*>(smem + buf * ROWS * 1024 + h * 16 * 1024 + put_lane) = r[h];
  };
  // (ONE tile ahead in registers -- a second set would not fit beside the weights, the fragments and the softmax's
  //  temporaries; the tile after next is asked for as soon as this one's rows are in LDS)
  u32x4 ra[2];
  fetch(0, ra);
  put(0, ra);
  if (my_tiles > 1) fetch(1, ra);
  auto one_tile = [&](int i, u32x4 (&r1)[2]) __attribute__((always_inline)) {
    const int buf = i & 1;
    __syncthreads();                       // tile i is complete in LDS; every wave has left tile i - 1 and its logits
    if (i + 1 < my_tiles) put(buf ^ 1, r1);
    if (i + 2 < my_tiles) fetch(i + 2, r1);
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    {
      const unsigned char* tb = smem + buf * ROWS * 1024;
      u32x4 fr[2][2];
#pragma unroll
      for (int h = 0; h < 2; ++h) fr[0][h] = *reinterpret_cast<const u32x4*>(tb + frag_lane + h * 16 * 1024);
#pragma unroll
      for (int q = 0; q < KSTEPS; ++q) {
        if (q + 1 < KSTEPS) {
          const int q1 = q + 1;
          const unsigned char* ap = tb + (frag_lane ^ (unsigned)(64 * (q1 >> 2))) + 256 * (q1 & 3);
#pragma unroll
          for (int h = 0; h < 2; ++h) fr[q1 & 1][h] = *reinterpret_cast<const u32x4*>(ap + h * 16 * 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 fb = __builtin_bit_cast(bf16x8, bu[4 * (q & 3) + (q >> 2)]);
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[h] = mfma16(__builtin_bit_cast(bf16x8, fr[q & 1][h]), fb, acc[h]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // accumulator layout (rows 4 (lane >> 4) + r, column 16 wave + (lane & 15)) -> the tile's logits in LDS
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) zl[(h * 16 + 4 * (lane >> 4) + r) * LDZ + 16 * wave + (lane & 15)] = acc[h][r];
    __syncthreads();
    // Waves 0-7 take FOUR rows each in one pass (round 4): 16 lanes per row, 16 characters per lane (64 k + 4 (l & 15) .. + 3, k = 0 .. 3),
    // the reductions are four DPP steps inside the 16-lane rows and serve four rows at once -- with a row per pass and wave-wide
    // reductions every wave spent 150 vector instructions per row (the width-128 kernel's measurements, lstm_scan_w128.hip).
    if (wave < 8) {
      const int rs4 = lane >> 4, cl = lane & 15;
      const int lr = 4 * wave + rs4;
      const long row = (long)(rg + (long)i * a.n_rg) * ROWS + lr;
      float e[16];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 z = *reinterpret_cast<const f32x4*>(zl + lr * LDZ + 64 * k + 4 * cl);
#pragma unroll
        for (int j = 0; j < 4; ++j) e[4 * k + j] = z[j];
      }
      float mloc = e[0];
#pragma unroll
      for (int j = 1; j < 16; ++j) mloc = fmaxf(mloc, e[j]);
      const float mx = row16_max(mloc);
      // the first character that reaches the maximum (Keras' argmax); a lane's characters ascend with j
      int first = 0x7fffffff;
#pragma unroll
      for (int j = 15; j >= 0; --j) first = e[j] == mx ? 64 * (j >> 2) + 4 * cl + (j & 3) : first;
      const int amax = row16_min_i(first);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        e[j] = __expf(e[j] - mx);      // (exp2-based: the arguments are <= 0)
        sum += e[j];
      }
      const float inv = 1.f / row16_sum(sum);
      const int b = (int)(row % a.B), tt = (int)(row / a.B);
      int t = a.tgt[(long)b * a.T + tt];
      bool counts = true;
      if (a.last_only && tt != a.T - 1) { t = -1; counts = false; }
      if (t < -1) { t = -1; counts = false; }      // (a dummy stream added by the caller's padding: no accuracy either)
      float ploc = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        e[j] *= inv;
        ploc = (64 * (j >> 2) + 4 * cl + (j & 3)) == t ? e[j] : ploc;
      }
      const float pt = row16_sum(ploc);              // (one lane of the row holds the target's probability)
      const bool valid = t >= 0;
      const bool active = valid && pt >= 1e-7f && pt <= 1.f - 1e-7f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        unsigned short g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float gk = active ? e[4 * k + j] : 0.f;
          if (active && 64 * k + 4 * cl + j == t) gk -= 1.f;
          g[j] = f2bf(gk * a.inv_count);
        }
        // (the row differs from lane to lane: all of the address in the vector offset)
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{(unsigned)g[0] | ((unsigned)g[1] << 16), (unsigned)g[2] | ((unsigned)g[3] << 16)}, rs_dl,
                                              (int)(unsigned)(row * V * 2 + (64 * k + 4 * cl) * 2), 0, 0);
      }
      if (cl == 0) {
        float l = 0.f;
        if (valid) {
          const float pc = fminf(fmaxf(pt, 1e-7f), 1.f - 1e-7f);
          l = -__logf(pc) * a.inv_count;
        }
        const int tsafe = valid ? t : 0;
        __builtin_amdgcn_raw_buffer_store_b64(u32x2{__builtin_bit_cast(unsigned, l), __builtin_bit_cast(unsigned, (counts && amax == tsafe) ? a.inv_count : 0.f)},
                                              rs_rs, (int)(unsigned)(row * 8), 0, 0);
      }
    }
  };
  for (int i = 0; i < my_tiles; ++i) one_tile(i, ra);
}

// ---------------------------------------------------------------- backward
// LDS map (bytes): tile [2][4*KSTEPS][1024] | zt [16 waves][16][17] f32 | pub [4 gates][16 rows][64 units] bf16 | flags | hand-off words [64]
constexpr int bwd2_lds_bytes(int ksteps) { return 2 * 4 * ksteps * 1024 + 16 * 16 * 17 * 4 + 4 * 16 * 64 * 2 + 16 + 256; }

// One layer, 16-row blocks, NP blocks per workgroup and step (2..4).  Wave = (K quarter = gate kq4, unit group ug):
// dh_rec[16 x 16] = dZ[t+1][16 x W(gate kq4)] . Un[W(gate kq4) x 16 units]; the four gate partials meet in LDS.
// a.G is gate-interleaved ([row][unit][4]); dZ leaves gate-major as before (the GEMMs behind it are unchanged).
// Rolling sentinels (a.sentinel == 2) as in the first generation: the publishing lanes re-arm step t - 2 while they store
// step t.  The re-arming store has completed before the same lanes publish step t - 1: a whole step of NP >= 2 blocks
// lies between them, and the counted tile wait at the top of the block after next covers every store of this one.
template <int NP, bool FLAGS>
__global__ __launch_bounds__(1024, 1) void lstm_scan_bwd_wide2_kernel(const KlScanBwd a) {
  constexpr int KSTEPS = 16;              // (width 512 only, see the forward scan)
  constexpr int W = KSTEPS * 32;
  constexpr int NWG_RB = W / 64;
  constexpr int JW = KSTEPS / 4;          // tile pieces per wave
  constexpr int NPIECE = 4 * KSTEPS;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq4 = wave & 3, ug = wave >> 2;
  const int n_rg = a.n_rg, B = a.B, T = a.T;
  const int xcd = blockIdx.x & 7, yy = blockIdx.x >> 3;
  const int cg = yy % NWG_RB, rq = yy / NWG_RB, rg = xcd * ((n_rg + 7) >> 3) + rq;
  if (rg >= n_rg) return;
  const int u0 = cg * 64;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float (*zt)[16][17] = reinterpret_cast<float (*)[16][17]>(smem + 2 * NPIECE * 1024);
  bf16_t* pub = reinterpret_cast<bf16_t*>(smem + 2 * NPIECE * 1024 + 16 * 16 * 17 * 4);
  int& ok_flag = *reinterpret_cast<int*>(smem + 2 * NPIECE * 1024 + 16 * 16 * 17 * 4 + 4 * 16 * 64 * 2);
  unsigned* const fl_l = reinterpret_cast<unsigned*>(smem + 2 * NPIECE * 1024 + 16 * 16 * 17 * 4 + 4 * 16 * 64 * 2 + 16);
  const unsigned lds_tile = (unsigned)(size_t)(lds_void_t*)smem;

  u32x4 bu[KSTEPS];
  {
    const long wrow = (long)(u0 + ug * 16 + (lane & 15)) * 4 * W + (long)kq4 * W + (lane >> 4) * 8;
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j) bu[j] = *reinterpret_cast<const u32x4*>(a.Un[0] + wrow + j * 32);
  }
  const int er = wave, eu = lane;                  // epilogue thread = (row = wave: scalar, unit of 64)
  const long BW = (long)B * W;
  const bf16_t* Gl = a.G[0];
  const float* Cl = a.C[0];
  bf16_t* dZl = a.dZ[0];
  const bf16_t* dHb = a.dHb;
  const bf16_t* Cb = a.Cb;
  const float* maskl = a.mask[0];
  unsigned* status = a.status;
  // per block of this workgroup (slot 0 = the current one: rotated): running dc, the cell state c_t of the step
  // being processed (the c_{t-1} loaded for step t is the c_t of step t-1), the dropout mask on dH
  float dcr[NP], ccur[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const long row = (long)(rg + p * n_rg) * 16 + er;
    dcr[p] = 0.f;
    ccur[p] = Cl[(long)T * BW + row * W + u0 + eu];
  }
  float dbacc[4] = {0.f, 0.f, 0.f, 0.f};
  const __amdgpu_buffer_rsrc_t rs_own = make_rsrc(dZl, (long)T * BW * 4 * 2);
  const __amdgpu_buffer_rsrc_t rs_null = make_rsrc(dZl, 0);
  // Hand-off by flags (a.flags; NP >= 3): each publishing wave posts "my rows of step t are in memory" as the number
  // epoch - t in its own word, flags[row block][column group * 8 + wave], once its stores have completed -- which it
  // learns for free half a block later, where it waits for its epilogue inputs anyway.  A consumer looks at the 64
  // words of a row block before it requests the tile.  The numbers only grow (the epoch advances by T + 2 per launch,
  // bumped by a one-block kernel in front: a replayed hipGraph must not see last launch's flags as this one's), so
  // nothing is re-armed: no second 4W-wide store per row and step, which is 8 of the scan's 36 bytes per cell.
  unsigned* const flags = FLAGS ? a.flags : nullptr;
  const unsigned epoch = FLAGS ? *a.epoch : 0u;
  const __amdgpu_buffer_rsrc_t rs_fl = make_rsrc(flags, FLAGS ? (long)a.n_rb * 64 * 4 : 0);
  bool alive = true;
  if (tid == 0) ok_flag = 1;
#pragma unroll
  for (int j = 0; j < KSTEPS; ++j) asm volatile("" : "+v"(bu[j]));
#pragma unroll
  for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(ccur[p]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bool local = false;      // XCD-local hand-off (opt-in), as in the forward scan
  if (!FLAGS && a.xcc_slots)
    local = __builtin_amdgcn_readfirstlane(
                xcd_local_group(a.xcc_slots, a.gen, NWG_RB, [&](int j) { return xcd + 8 * (rq * NWG_RB + j); }, &ok_flag + 1, status) ? 1 : 0) != 0;
  SSTAMP_INIT(0);

  int vq = 0, seq_tile[2] = {0, 0}, seq_in = 0;
  // Tile image in LDS: piece (gate quarter q, row r) = the 1 KiB of row r's quarter q, one fully coalesced DMA
  // instruction each, its 16-byte chunks XOR-swizzled at the source (chunk c at position c ^ r: see the forward
  // scan).  Wave w fetches the four quarters of row w: 4 KiB contiguous.
  static_assert(KSTEPS == 16, "one quarter of a tile row = one 1 KiB DMA piece");
  const unsigned dma_lane = (unsigned)((lane ^ wave) * 16);
  const unsigned frag_lane = (unsigned)((lane & 15) * 1024 + (((lane >> 4) ^ (lane & 3)) * 16) + 64 * ((lane >> 2) & 3));
  auto issue_tile = [&](int t, int r0, int buf) __attribute__((always_inline)) {      // tile = dZ[t + 1], rows r0 .. r0 + 16, this wave's row
#pragma unroll
    for (int j = 0; j < JW; ++j) arm16(smem + (buf * NPIECE + j * 16 + wave) * 1024 + lane * 16);      // (see arm16)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < JW; ++j) {
      const int p = j * 16 + wave;                      // (quarter j, row wave)
      const unsigned soff = (unsigned)((((long)(t + 1) * B + r0 + wave) * 4 * W + (long)j * W) * 2);
      if (local) glds16_nt_s(rs_own, dma_lane, soff, lds_tile + (unsigned)((buf * NPIECE + p) * 1024));
      else glds16_sc1_s(rs_own, dma_lane, soff, lds_tile + (unsigned)((buf * NPIECE + p) * 1024));
      ++vq;
    }
    seq_tile[buf] = vq;
  };
  // epilogue inputs of a block (gates 8 bytes, c_{t-1} and dh 4 bytes each): asm loads into registers, issued at
  // the top of the block they belong to and consumed behind its MFMA phase.  They are NOT carried around the loop:
  // a loop-carried asm destination was copied by the compiler at the loop head before its data had landed.
  const unsigned in_lane4 = (unsigned)((u0 + eu) * 4);
  int n = 0;
  for (int t = T - 1; t >= 0; --t) {
#pragma unroll 1
    for (int ip = 0; ip < NP; ++ip, ++n) {
      const int buf = n & 1;
      const int r0 = (rg + ip * n_rg) * 16;
      int t1 = t, ip1 = ip + 1;
      if (ip1 >= NP) { ip1 = 0; t1 = t - 1; }
      int t2 = t1, ip2 = ip1 + 1;
      if (ip2 >= NP) { ip2 = 0; t2 = t1 - 1; }
      const int r1 = (rg + ip1 * n_rg) * 16, r2 = (rg + ip2 * n_rg) * 16;
      SSTAMP(16);
      // Epilogue inputs of this block (gates 8 bytes, c_{t-1} and dh as bf16 or f32, the dropout mask on dH): asm loads of THIS
      // iteration into compiler registers armed with all-ones (which no valid datum is: a 16-bit load zero-extends),
      // consumed behind the MFMA phase.  They are not carried around the loop -- a loop-carried asm destination was copied
      // by the compiler at the loop head before its data had landed -- and not kept in registers "reserved" from the
      // compiler either: hipcc honoured neither amdgpu_num_vgpr nor asm clobbers once a variant needed the registers.
      u32x2 gin = u32x2{0xFFFFFFFFu, 0xFFFFFFFFu};
      unsigned cpin = 0xFFFFFFFFu, dhin = 0xFFFFFFFFu, mkin = maskl ? 0xFFFFFFFFu : 0x3f800000u;      // (no mask: 1.0f)
      {
        const long trow = (long)t * B + r0 + er;
        aload8_glb(gin, Gl + trow * W * 4, in_lane4 * 2);
        if (Cb && t > 0) aload2_glb(cpin, Cb + trow * W, in_lane4 >> 1);
        else aload4_glb(cpin, Cl + trow * W, in_lane4);
        aload2_glb(dhin, dHb + trow * W, in_lane4 >> 1);
        vq += 3;
        if (maskl) {
          aload4_glb(mkin, maskl + (long)(r0 + er) * W, in_lane4);
          ++vq;
        }
        seq_in = vq;
      }
      // (flags: the 64 words of the next block's rows come into LDS -- armed with 0 = "not yet" -- and are looked at behind
      //  the MFMAs; the last wave fetches them, it publishes nothing)
      // (a.pf_mode 2: the tile of the block after next is asked for -- into the buffer this block's MFMAs leave --, a block and
      //  a half before it is needed, and only BEHIND the epilogue: the request costs its wave ~170 cycles per piece, which
      //  then no longer stand between the MFMA phase and the publish)
      const bool pf2 = FLAGS && a.pf_mode == 2;
      const int tF = pf2 ? t2 : t1, rF = pf2 ? r2 : r1;
      if (FLAGS && wave == 15 && tF >= 0 && tF < T - 1) {
        fl_l[lane] = 0u;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        glds4_sc1_s(rs_fl, (unsigned)(lane * 4), (unsigned)((rF >> 4) * 256), lds_tile + (unsigned)(2 * NPIECE * 1024 + 16 * 16 * 17 * 4 + 4 * 16 * 64 * 2 + 16));
        ++vq;
      }
      if (!FLAGS && a.pf_mode == 0 && alive && t1 >= 0 && t1 < T - 1) issue_tile(t1, r1, buf ^ 1);       // (a.pf_mode: see the forward scan)
      if (alive && t < T - 1) {
        wait_vm(vq - seq_tile[buf]);
        SSTAMP(25);
        bool ok = true;
#pragma unroll
        for (int j = 0; j < JW; ++j)
          ok = ok && piece_there(smem + (buf * NPIECE + j * 16 + wave) * 1024 + lane * 16);
        ok = __all(ok);
        if (!ok) {
#ifdef KL_STAMP
          if (blockIdx.x == STAMP_WG && threadIdx.x == 0) stamp_lds[28] += 1;
#endif
          for (unsigned spin = 0; spin < SPIN_LIMIT && !ok; ++spin) {
#pragma unroll
            for (int j = 0; j < JW; ++j) {
              if (spin == 0) break;                 // (first round: only wait until everything issued has landed)
              const int p = j * 16 + wave;
              const unsigned soff = (unsigned)((((long)(t + 1) * B + r0 + wave) * 4 * W + (long)j * W) * 2);
              if (local) glds16_nt_s(rs_own, dma_lane, soff, lds_tile + (unsigned)((buf * NPIECE + p) * 1024));
              else glds16_sc1_s(rs_own, dma_lane, soff, lds_tile + (unsigned)((buf * NPIECE + p) * 1024));
              ++vq;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ok = true;
#pragma unroll
            for (int j = 0; j < JW; ++j)
              ok = ok && piece_there(smem + (buf * NPIECE + j * 16 + wave) * 1024 + lane * 16);
            ok = __all(ok);
            if (!ok) {
              if ((spin & 63) == 63 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
              __builtin_amdgcn_s_sleep(2);
            }
          }
          if (!ok) {
            __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok_flag = 0;
          }
        }
      }
      SSTAMP(17);
      __syncthreads();
      SSTAMP(18);
      alive = __builtin_amdgcn_readfirstlane(ok_flag) != 0;
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t < T - 1) {
        // (k-step order and read-ahead as in the forward scan)
        const unsigned char* tb = smem + (buf * NPIECE + kq4 * 16) * 1024;
        u32x4 fr[2];
        fr[0] = *reinterpret_cast<const u32x4*>(tb + frag_lane);
#pragma unroll
        for (int q = 0; q < KSTEPS; ++q) {
          if (q + 1 < KSTEPS) {
            const int q1 = q + 1;
            fr[q1 & 1] = *reinterpret_cast<const u32x4*>(tb + (frag_lane ^ (unsigned)(64 * (q1 >> 2))) + 256 * (q1 & 3));
          }
          __builtin_amdgcn_sched_barrier(0);
          acc = mfma16(__builtin_bit_cast(bf16x8, fr[q & 1]), __builtin_bit_cast(bf16x8, bu[4 * (q & 3) + (q >> 2)]), acc);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) zt[wave][(lane >> 4) * 4 + r][lane & 15] = acc[r];
      SSTAMP(19);
      __syncthreads();
      SSTAMP(20);
      auto request_next = [&]() __attribute__((always_inline)) {
        if (alive && tF >= 0 && tF < T - 1) {
          // (unsigned distance: a word of this launch is at most T - 1 behind the epoch, one of an earlier launch at least T + 2)
          const unsigned far = (unsigned)(tF + 1);
          bool ready = __all(epoch - lds_peek(lds_tile + (unsigned)(2 * NPIECE * 1024 + 16 * 16 * 17 * 4 + 4 * 16 * 64 * 2 + 16) + (unsigned)(lane * 4)) <= far);
          if (!ready) {      // not posted when the words were fetched, or the fetch itself still on its way: ask memory
#ifdef KL_STAMP
            if (blockIdx.x == STAMP_WG && threadIdx.x == 0) stamp_lds[28] += 1;
#endif
            for (unsigned spin = 0; spin < SPIN_LIMIT && !ready; ++spin) {
              const unsigned now = __hip_atomic_load(flags + (long)(rF >> 4) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ready = __all(epoch - now <= far);
              if (!ready) {
                if ((spin & 63) == 63 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
                __builtin_amdgcn_s_sleep(2);
              }
            }
            if (!ready) {
              __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ok_flag = 0;
            }
          }
          if (ready) issue_tile(tF, rF, pf2 ? buf : buf ^ 1);
        }
      };
      if (FLAGS) {
        // the publishing waves post the block before this one: behind this wait (their epilogue inputs, needed in a moment
        // anyway) nothing of theirs is in flight, so that block's stores are in memory
        if (wave < 8 && n > 0) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          int tp = t, ipp = ip - 1;
          if (ipp < 0) { ipp = NP - 1; tp = t + 1; }
          const int rbp = rg + ipp * n_rg;
          if (lane == 0)
            __builtin_amdgcn_raw_buffer_store_b32(epoch - (unsigned)tp, alive ? rs_fl : rs_null, (rbp * 64 + cg * 8 + wave) * 4, 0, 16);
          ++vq;
        }
        if (!pf2) request_next();
      } else
      if (a.pf_mode == 1 && alive && t1 >= 0 && t1 < T - 1) issue_tile(t1, r1, buf ^ 1);     // (its buffer was last read a block ago)
      if (!FLAGS && a.pf_mode == 2 && alive && t2 >= 0 && t2 < T - 1) issue_tile(t2, r2, buf);         // (every wave has finished this block's MFMAs)
      // ---- epilogue: thread = (row er, unit eu)
      // (loads and stores retire independently, so the count is an estimate: the armed registers are checked)
      wait_vm(vq - seq_in);
      use_regs(gin);
      use_regs(cpin);
      use_regs(dhin);
      use_regs(mkin);
      if (__any(max(max(gin.x, gin.y), max(cpin, max(dhin, mkin))) == 0xFFFFFFFFu)) {
#ifdef KL_STAMP
        if (blockIdx.x == STAMP_WG && threadIdx.x == 0) stamp_lds[29] += 1;
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        use_regs(gin);
        use_regs(cpin);
        use_regs(dhin);
        use_regs(mkin);
        // (nothing is in flight any more: registers that are STILL all-ones never received their data -- e.g. a destination the
        //  compiler moved; tools/audit_async_regs.py looks for that in the ISA, this is the run-time net)
        if (__any(max(max(gin.x, gin.y), max(cpin, max(dhin, mkin))) == 0xFFFFFFFFu)) {
          __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok_flag = 0;
        }
      }
      const float gi = bf2f((bf16_t)(gin.x & 0xffffu)), gf = bf2f((bf16_t)(gin.x >> 16));
      const float gg = bf2f((bf16_t)(gin.y & 0xffffu)), go = bf2f((bf16_t)(gin.y >> 16));
      const float cp = (Cb && t > 0) ? u2f(cpin << 16) : u2f(cpin);
      float dh = u2f(dhin << 16);
      SSTAMP(21);
      const int wz = (eu >> 4) * 4;      // the four K-quarter waves of this unit group
      dh = dh * u2f(mkin) + (zt[wz][er][eu & 15] + zt[wz + 1][er][eu & 15] + zt[wz + 2][er][eu & 15] + zt[wz + 3][er][eu & 15]);
      const float tc = fast_tanh(ccur[0]);
      const float dc = dh * go * (1.f - tc * tc) + dcr[0];
      dcr[0] = dc * gf;
      ccur[0] = cp;
      const float d_o = dh * tc, d_i = dc * gg, d_g = dc * gi, d_f = dc * cp;
      const unsigned z0 = f2bf(d_i * gi * (1.f - gi)), z1 = f2bf(d_f * gf * (1.f - gf));
      const unsigned z2 = f2bf(d_g * (1.f - gg * gg)), z3 = f2bf(d_o * go * (1.f - go));
      if (alive) {
        dbacc[0] += bf2f((bf16_t)z0); dbacc[1] += bf2f((bf16_t)z1); dbacc[2] += bf2f((bf16_t)z2); dbacc[3] += bf2f((bf16_t)z3);
      }
      pub[(0 * 16 + er) * 64 + eu] = (bf16_t)z0;
      pub[(1 * 16 + er) * 64 + eu] = (bf16_t)z1;
      pub[(2 * 16 + er) * 64 + eu] = (bf16_t)z2;
      pub[(3 * 16 + er) * 64 + eu] = (bf16_t)z3;
      if (FLAGS && pf2) request_next();
      SSTAMP(22);
      __syncthreads();
      SSTAMP(23);
      // ---- publish dZ[t] (eight waves, one 16-byte write-through store per lane) and re-arm step t - 2
      if (wave < 8) {
        int stid = tid;
        asm volatile("" : "+v"(stid));
        const int g = stid >> 7, prow = (stid >> 3) & 15, seg = stid & 7;
        const uint4 v = *reinterpret_cast<const uint4*>(pub + (g * 16 + prow) * 64 + seg * 8);
        const unsigned off = (unsigned)((((long)t * B + r0 + prow) * 4 * W + (long)g * W + u0 + seg * 8) * 2);
        const unsigned soff = off - (unsigned)((long)2 * B * 4 * W * 2);
        const uint4 ones = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        if (FLAGS) {
          store16_sc1(alive ? rs_own : rs_null, off, v);
          --vq;      // (one store, not two)
        } else if (local) {      // plain stores: they stay in this XCD's L2, where all their readers are
          store16(alive ? rs_own : rs_null, off, 0u, v);
          store16((alive && t >= 2) ? rs_own : rs_null, soff, 0u, ones);
        } else {
          store16_sc1(alive ? rs_own : rs_null, off, v);
          store16_sc1((alive && t >= 2) ? rs_own : rs_null, soff, ones);      // (a null buffer drops the store, the count stays)
        }
        vq += 2;
      }
      SSTAMP(24);
      if (NP > 1) {
        const float d0 = dcr[0], c0 = ccur[0];
#pragma unroll
        for (int p = 0; p + 1 < NP; ++p) { dcr[p] = dcr[p + 1]; ccur[p] = ccur[p + 1]; }
        dcr[NP - 1] = d0;
        ccur[NP - 1] = c0;
      }
    }
  }
  SSTAMP_FLUSH();
  // db[g*W + u] += sum over this workgroup's rows and all steps (16 partials per column meet in LDS)
  if (a.db) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);     // [4 gates][16 rows][64 units]
#pragma unroll
    for (int g = 0; g < 4; ++g) red[(g * 16 + er) * 64 + eu] = dbacc[g];
    __syncthreads();
    if (tid < 256) {
      const int g = tid >> 6, u = tid & 63;
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += red[(g * 16 + r) * 64 + u];
      atomicAdd(a.db + (long)g * W + u0 + u, sum);
    }
  }
}


// ---------------------------------------------------------------- backward, eight waves with the tile through registers
// (five or more 16-row blocks per workgroup and step; below that the 16-wave kernel above)
// What round 3's stamps of the 16-wave kernel showed: the address unit takes ~16 cycles per vector-memory wave instruction
// whatever its width (~32 for a 1 KiB LDS-DMA piece), so 16 waves x (4 tile pieces + 3..4 two- and four-byte input loads +
// stores) kept it busy for 2400-3000 cycles per block, in bursts; the LDS served the same 1 KiB fragment to four waves (one
// MFMA per read); the epilogue inputs, requested at the top of the block they belong to, arrived 1500-2200 cycles late; and
// the 128 registers of a 1024-thread workgroup left no room for anything in flight.  Here:
//  * 512 threads, 256 registers each: wave = (gate quarter of K, 32 units) -- every 1 KiB fragment read feeds two MFMAs with
//    independent accumulators, the fragment reads per block halve;
//  * an epilogue thread owns TWO neighbouring units of one row: its inputs come as 16 + 4 + 4 bytes (gates, c_{t-1}, dH) --
//    24 wave instructions per block instead of 64 -- into ACCUMULATOR registers a0..a5, requested a whole block ahead between
//    the MFMAs of the block before.  Accumulator registers, because a load in flight around the loop's back-edge must land
//    where the compiler never looks: a compiler-chosen VGPR destination was copied at the loop head before its data had
//    landed.  These kernels are built with a fixed number of accumulator registers (function attribute
//    "amdgpu-agpr-alloc", set on the device IR by tools/build_agpr_tu.sh: HIP has no spelling for it, and without it the
//    compiler halves the register budget as soon as an AGPR is named) and with -amdgpu-mfma-vgpr-form (the MFMA accumulator
//    stays in VGPRs); the six are register variables of the kernel (`register unsigned x asm("a0")`) that every statement
//    touching them takes as operands, so the compiler sees them live from the request to the read-out a block later and
//    neither re-uses nor copies them.  tools/audit_async_regs.py checks on the generated ISA that no compiler instruction
//    names an accumulator register.  They are armed with all-ones before a request and polled at the read-out.
//    (NOT two halves of one register by d16 loads: with SRAM ECC on -- gfx950's default -- a d16 load rewrites the whole register.)
//  * the per-slot state (running dc, c_t: 4 x NP registers, the dropout mask as a bit per cell) stays in registers.
// LDS map (bytes): tile [2][64][1024] | zt [8 waves][584 words] (rows of 36 words: conflict-free partial-tile writes and
// 8-byte epilogue reads) | pub [4 gates][16 rows][64 units] bf16 | flags | hand-off words [2][64]
#ifndef KL_BWD_VAR
#define KL_BWD_VAR 0      /* timing experiments only (tools/gpu_variants.sh): 8 no tile writes to LDS, 16 no tile loads, 32 no MFMAs, 64 no fragment reads */
#endif
constexpr int B3_ZT_ROW = 36, B3_ZT_WAVE = 16 * 36 + 8;
constexpr int bwd3_lds_bytes() { return 2 * 64 * 1024 + 8 * B3_ZT_WAVE * 4 + 4 * 16 * 64 * 2 + 16 + 512; }      // (two slots of hand-off words: the register-tile kernel)

#define KL_B3_DECL register unsigned lb0_ asm("a0"), lb1_ asm("a1"), lb2_ asm("a2"), lb3_ asm("a3"), lb4_ asm("a4"), lb5_ asm("a5")
// (buffer loads: resource + 32-bit scalar row offset + lane offset -- the arrays are below 4 GiB, kl_scan_wide2_phases -- instead
//  of 64-bit base pointers: a third of the scalar instructions of a block went into pointer arithmetic)
#define KL_B3_REQ_G(rs_, soff_, g_off)                                                                                             \
  asm volatile("v_accvgpr_write_b32 a0, -1\n\tv_accvgpr_write_b32 a1, -1\n\tv_accvgpr_write_b32 a2, -1\n\t"                         \
               "v_accvgpr_write_b32 a3, -1\n\tv_accvgpr_write_b32 a4, -1\n\tv_accvgpr_write_b32 a5, -1\n\ts_nop 4\n\t"              \
               "buffer_load_dwordx4 a[0:3], %6, %7, %8 offen"                                                                      \
               : "=a"(lb0_), "=a"(lb1_), "=a"(lb2_), "=a"(lb3_), "=a"(lb4_), "=a"(lb5_) : "v"(g_off), "s"(rs_), "s"(soff_) : "memory")
#define KL_B3_REQ_CP(rs_, soff_, h_off) asm volatile("s_nop 4\n\tbuffer_load_dword a4, %1, %2, %3 offen" : "+a"(lb4_) : "v"(h_off), "s"(rs_), "s"(soff_) : "memory")
#define KL_B3_REQ_DH(rs_, soff_, h_off) asm volatile("s_nop 4\n\tbuffer_load_dword a5, %1, %2, %3 offen" : "+a"(lb5_) : "v"(h_off), "s"(rs_), "s"(soff_) : "memory")
#define KL_B3_READ(g0, g1, g2, g3, cp, dh)                                                                                         \
  asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\t"                            \
               "v_accvgpr_read_b32 %3, a3\n\tv_accvgpr_read_b32 %4, a4\n\tv_accvgpr_read_b32 %5, a5"                               \
               : "=v"(g0), "=v"(g1), "=v"(g2), "=v"(g3), "=v"(cp), "=v"(dh)                                                        \
               : "a"(lb0_), "a"(lb1_), "a"(lb2_), "a"(lb3_), "a"(lb4_), "a"(lb5_) : "memory")

// ---- the dZ tile through REGISTERS instead of LDS-DMA
// 64 LDS-DMA pieces per block cost the CU's address unit ~2000 cycles (~32 per 1 KiB piece; a 1 KiB register load passes in
// 16), and they can only be requested once the target buffer is free -- behind the MFMA phase of the block that last read it
// --, so they went out as one burst and landed ~2000 cycles late (stamps of an 8-wave LDS-DMA form, round 3).  Here every
// wave loads eight 1 KiB pieces of the tile of the block AFTER NEXT into 32 accumulator registers (a8..a39) between the MFMAs
// of the current block -- two blocks ahead, no LDS involved, so nothing has to be free --, and a block later, again between
// MFMAs, writes them into the LDS buffer that has just been released (ds_write_b128 straight from the accumulator registers)
// and re-uses the registers for the next tile.  The register file is the third tile buffer the LDS has no room for.
//  * order inside a wave's queue: the eight tile loads of a block are issued BEFORE its three epilogue-input loads; loads
//    return in order, so when the (armed, polled) inputs of block n have landed, the tile pieces requested in front of them
//    have too -- and then nothing but stores is in flight, they retire in order, and "the block before last has reached
//    memory" is an exact counted wait (s_waitcnt vmcnt(2)) instead of a drain, which at this block length (~1.8 us against
//    a write-through acknowledgement of ~3 us) would stall every block.  All eight waves publish one 16-byte store per lane
//    and post one flag word;
//  * flags only: a tile is requested two blocks before it is used and its flag is posted two blocks after its data, hence
//    five or more blocks per step.  The flag words of the tile to request come into LDS a block earlier (wave 7, right behind
//    its post: older than that block's inputs in the queue, so the argument above covers them; two slots, so that the fetch
//    never overwrites words another wave is still looking at);
//  * XCD-local publishes where the row group's eight workgroups are verified to share an XCD (plain stores that stay in its L2).
#define KL_B4_TILE_CLOBBER "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", \
                           "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39"
// piece k of this wave (registers a[8 + 4k : 11 + 4k]; register names and offsets must be literal): piece k = (gate quarter
// k >> 1, row 2 wave + (k & 1)).  Out to LDS: addr_ = this lane's place in piece 0 of the target buffer, the piece itself is
// an immediate offset.  In from memory: soff_ = the byte offset of row 2 wave of the tile, voff0_ / voff1_ = this lane's
// (swizzled) chunk in an even / odd row (+ one row), the quarter an immediate offset -- no per-piece address arithmetic.
#define KL_B4_PUT(k_, addr_)                                                                                                       \
  do {                                                                                                                             \
    if ((k_) == 0) asm volatile("ds_write_b128 %0, a[8:11] offset:0" :: "v"(addr_) : "memory");                    \
    if ((k_) == 1) asm volatile("ds_write_b128 %0, a[12:15] offset:1024" :: "v"(addr_) : "memory");                    \
    if ((k_) == 2) asm volatile("ds_write_b128 %0, a[16:19] offset:16384" :: "v"(addr_) : "memory");                    \
    if ((k_) == 3) asm volatile("ds_write_b128 %0, a[20:23] offset:17408" :: "v"(addr_) : "memory");                    \
    if ((k_) == 4) asm volatile("ds_write_b128 %0, a[24:27] offset:32768" :: "v"(addr_) : "memory");                    \
    if ((k_) == 5) asm volatile("ds_write_b128 %0, a[28:31] offset:33792" :: "v"(addr_) : "memory");                    \
    if ((k_) == 6) asm volatile("ds_write_b128 %0, a[32:35] offset:49152" :: "v"(addr_) : "memory");                    \
    if ((k_) == 7) asm volatile("ds_write_b128 %0, a[36:39] offset:50176" :: "v"(addr_) : "memory");                    \
  } while (0)
#define KL_B4_GET(k_, voff0_, voff1_, rsrc_, soff_)                                                                                \
  do {                                                                                                                             \
    if ((k_) == 0) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[8:11], %0, %1, %2 offen offset:0 sc1" :: "v"(voff0_), "s"(rsrc_), "s"(soff_) : "memory", KL_B4_TILE_CLOBBER); \
    if ((k_) == 1) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[12:15], %0, %1, %2 offen offset:0 sc1" :: "v"(voff1_), "s"(rsrc_), "s"(soff_) : "memory", KL_B4_TILE_CLOBBER); \
    if ((k_) == 2) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[16:19], %0, %1, %2 offen offset:1024 sc1" :: "v"(voff0_), "s"(rsrc_), "s"(soff_) : "memory", KL_B4_TILE_CLOBBER); \
    if ((k_) == 3) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[20:23], %0, %1, %2 offen offset:1024 sc1" :: "v"(voff1_), "s"(rsrc_), "s"(soff_) : "memory", KL_B4_TILE_CLOBBER); \
    if ((k_) == 4) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[24:27], %0, %1, %2 offen offset:2048 sc1" :: "v"(voff0_), "s"(rsrc_), "s"(soff_) : "memory", KL_B4_TILE_CLOBBER); \
    if ((k_) == 5) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[28:31], %0, %1, %2 offen offset:2048 sc1" :: "v"(voff1_), "s"(rsrc_), "s"(soff_) : "memory", KL_B4_TILE_CLOBBER); \
    if ((k_) == 6) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[32:35], %0, %1, %2 offen offset:3072 sc1" :: "v"(voff0_), "s"(rsrc_), "s"(soff_) : "memory", KL_B4_TILE_CLOBBER); \
    if ((k_) == 7) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 a[36:39], %0, %1, %2 offen offset:3072 sc1" :: "v"(voff1_), "s"(rsrc_), "s"(soff_) : "memory", KL_B4_TILE_CLOBBER); \
  } while (0)

// (KL_NO_REGTILE: tools/build_agpr_tu.sh could not set the register-allocation attribute this kernel needs -- another compiler
// version --: it is left out, kl_scan_bwd_regtile_min_np() then keeps every shape on the 16-wave kernel)
#ifndef KL_NO_REGTILE
template <int NP>
__global__ __launch_bounds__(512, 1) void lstm_scan_bwd_regtile_kernel(const KlScanBwd a) {
  static_assert(NP >= 5, "a tile is requested two blocks ahead of its use and posted two blocks behind its publish");
  constexpr int KSTEPS = 16, W = 512, NWG_RB = W / 64, NPIECE = 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq4 = wave & 3, uh = wave >> 2;
  const int n_rg = a.n_rg, B = a.B, T = a.T;
  const int xcd = blockIdx.x & 7, yy = blockIdx.x >> 3;
  const int cg = yy % NWG_RB, rq = yy / NWG_RB, rg = xcd * ((n_rg + 7) >> 3) + rq;
  if (rg >= n_rg) return;
  const int u0 = cg * 64;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const zt = reinterpret_cast<float*>(smem + 2 * NPIECE * 1024);
  bf16_t* const pub = reinterpret_cast<bf16_t*>(smem + 2 * NPIECE * 1024 + 8 * B3_ZT_WAVE * 4);
  int& lds_word = *reinterpret_cast<int*>(smem + 2 * NPIECE * 1024 + 8 * B3_ZT_WAVE * 4 + 4 * 16 * 64 * 2);
  constexpr int FL_OFF = 2 * NPIECE * 1024 + 8 * B3_ZT_WAVE * 4 + 4 * 16 * 64 * 2 + 16;
  unsigned* const fl_l = reinterpret_cast<unsigned*>(smem + FL_OFF);
  const unsigned lds_tile = (unsigned)(size_t)(lds_void_t*)smem;

  u32x4 bu[2][KSTEPS];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const long wrow = (long)(u0 + uh * 32 + x * 16 + (lane & 15)) * 4 * W + (long)kq4 * W + (lane >> 4) * 8;
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j) bu[x][j] = *reinterpret_cast<const u32x4*>(a.Un[0] + wrow + j * 32);
  }
  const int er = 2 * wave + (lane >> 5), eu = 2 * (lane & 31);
  const long BW = (long)B * W;
  const float* Cl = a.C[0];
  const float* maskl = a.mask[0];
  unsigned* status = a.status;
  // (the dropout keep-masks of this thread's 2 NP cells as one bit each + their common scale: inverted dropout knows two
  //  values, 0 and 1 / keep-probability (rating.py:146-152); a mask with other values is refused through the status word)
  float dcr[NP][2], ccur[NP][2];
  unsigned mbits = 0u;
  float mscale = 1.f;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const long row = (long)(rg + p * n_rg) * 16 + er;
    const float2 c2 = *reinterpret_cast<const float2*>(Cl + (long)T * BW + row * W + u0 + eu);
    float2 m2 = float2{1.f, 1.f};
    if (maskl) m2 = *reinterpret_cast<const float2*>(maskl + row * W + u0 + eu);
    dcr[p][0] = dcr[p][1] = 0.f;
    ccur[p][0] = c2.x; ccur[p][1] = c2.y;
    mbits |= (m2.x != 0.f ? 1u : 0u) << (2 * p) | (m2.y != 0.f ? 2u : 0u) << (2 * p);
    if (m2.x != 0.f) { if (mscale != 1.f && m2.x != mscale) status[0] = 1u; mscale = m2.x; }
    if (m2.y != 0.f) { if (mscale != 1.f && m2.y != mscale) status[0] = 1u; mscale = m2.y; }
  }
  float dbsum = 0.f;
  const __amdgpu_buffer_rsrc_t rs_own = make_rsrc(a.dZ[0], (long)T * BW * 4 * 2);
  const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(a.G[0], (long)T * BW * 4 * 2);
  const __amdgpu_buffer_rsrc_t rs_cb = make_rsrc(a.Cb, (long)(T + 1) * BW * 2);
  const __amdgpu_buffer_rsrc_t rs_dh = make_rsrc(a.dHb, (long)T * BW * 2);
  unsigned* const flags = a.flags;
  const unsigned epoch = *a.epoch;
  const __amdgpu_buffer_rsrc_t rs_fl = make_rsrc(flags, (long)a.n_rb * 64 * 4);
  if (tid < 128) fl_l[tid] = 0u;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j) asm volatile("" : "+v"(bu[x][j]));
#pragma unroll
  for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(ccur[p][0]), "+v"(ccur[p][1]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // XCD-local hand-off: the eight workgroups that exchange a row group's tiles are dealt to ONE XCD by the grid mapping; where
  // that is verified (posted XCC ids, kl_scan_common.h) the publishes are PLAIN stores that stay in that XCD's L2, where the
  // partners' L1-bypassing loads find them.  Not verified: write-through stores.
  bool local = false;
  if (a.xcc_slots)
    local = __builtin_amdgcn_readfirstlane(
                xcd_local_group(a.xcc_slots, a.gen, NWG_RB, [&](int j) { return xcd + 8 * (rq * NWG_RB + j); }, &lds_word, status) ? 1 : 0) != 0;
  SSTAMP_INIT(0);

  // lane parts of the addresses (everything else is wave-uniform and lives in scalar registers)
  const unsigned src_lane0 = (unsigned)(((lane ^ (2 * wave)) & 63) * 16);                          // tile pieces of row 2 wave ...
  const unsigned src_lane1r = (unsigned)(((lane ^ (2 * wave + 1)) & 63) * 16 + 4 * W * 2);         // ... and of row 2 wave + 1
  const unsigned put_lane = lds_tile + (unsigned)(2 * wave * 1024 + lane * 16);
  const unsigned frag_lane = (unsigned)((lane & 15) * 1024 + (((lane >> 4) ^ (lane & 3)) * 16) + 64 * ((lane >> 2) & 3));
  const unsigned in_g = (unsigned)((u0 + eu) * 8 + (lane >> 5) * W * 8), in_h = (unsigned)((u0 + eu) * 2 + (lane >> 5) * W * 2);
  const unsigned pub_lane = (unsigned)(((((tid >> 3) & 15) * 4 * W) + (tid >> 7) * W + u0 + (tid & 7) * 8) * 2);
  const bf16_t* const pub_rd = pub + ((tid >> 7) * 16 + ((tid >> 3) & 15)) * 64 + (tid & 7) * 8;
  const bf16_t* const db_col = pub + ((tid & 255) >> 6) * 1024 + (tid >> 8) * 512 + (tid & 63);
  KL_B3_DECL;
  // requests of a block's epilogue inputs: trow = first row of the block + 2 wave (time-major row index)
#define KL_B4_REQUEST(k_, trow_)                                                                                                    \
  do {                                                                                                                             \
    if (k_ == 0) KL_B3_REQ_G(rs_g, (unsigned)(trow_) * (unsigned)(W * 8), in_g);                                                   \
    if (k_ == 1) KL_B3_REQ_CP(rs_cb, (unsigned)(trow_) * (unsigned)(W * 2), in_h);                                                 \
    if (k_ == 2) KL_B3_REQ_DH(rs_dh, (unsigned)(trow_) * (unsigned)(W * 2), in_h);                                                 \
  } while (0)
  {
    const int trow = (T - 1) * B + rg * 16 + 2 * wave;
    KL_B4_REQUEST(0, trow); KL_B4_REQUEST(1, trow); KL_B4_REQUEST(2, trow);
  }

  // positions of this block and the three after it (time step, first row); a position past the end keeps t = -1
  int t0 = T - 1, ip0 = 0, t1, ip1, t2, ip2, t3, ip3;
  auto next_pos = [&](int tt, int ii, int& tn, int& in) __attribute__((always_inline)) {
    in = ii + 1; tn = tt;
    if (in >= NP) { in = 0; tn = tt - 1; }
    if (tt < 0) tn = -1;
  };
  next_pos(t0, ip0, t1, ip1); next_pos(t1, ip1, t2, ip2); next_pos(t2, ip2, t3, ip3);
  bool loaded = false;      // the landing registers hold the tile of the NEXT block (requested during the block before this one)
  for (int n = 0; n < T * NP; ++n) {
    const int t = t0, buf = n & 1;
    const int r0 = (rg + ip0 * n_rg) * 16, r1 = (rg + ip1 * n_rg) * 16, r2 = (rg + ip2 * n_rg) * 16;
    SSTAMP(16);
    SSTAMP(25);
    SSTAMP(17);
    __syncthreads();
    SSTAMP(18);
    // ---- this block's epilogue inputs out of their landing registers; with them every load this wave has issued so far has
    // landed (they were the last ones requested during the block before, and loads return in order): the tile pieces of the
    // next block in a8..a39 too.  Polled, not waited for by count: a count of the operations issued since would also wait
    // for OLDER stores (the post of the block before, a write-through store that takes longer than a block to be acknowledged)
    unsigned gin[4], cpin, dhin;
    KL_B3_READ(gin[0], gin[1], gin[2], gin[3], cpin, dhin);
    if (__any(max(max(max(gin[0], gin[1]), max(gin[2], gin[3])), max(cpin, dhin)) == 0xFFFFFFFFu)) {
      bool got = false;
      for (unsigned spin = 0; spin < SPIN_LIMIT; ++spin) {
#ifdef KL_STAMP
        if (blockIdx.x == STAMP_WG && threadIdx.x == 0) stamp_lds[29] += 1;
#endif
        __builtin_amdgcn_s_sleep(1);
        KL_B3_READ(gin[0], gin[1], gin[2], gin[3], cpin, dhin);
        if (!__any(max(max(max(gin[0], gin[1]), max(gin[2], gin[3])), max(cpin, dhin)) == 0xFFFFFFFFu)) { got = true; break; }
        if ((spin & 255) == 255 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
      }
      if (!got) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (n > 1) {
      // Post the block before last: only stores are in flight now, they retire in order, and the two youngest -- the last
      // block's publish and the post before this one (none yet in block 2) -- are all that may still be on their way
      if (n > 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      int tp = t, ipp = ip0 - 2;
      if (ipp < 0) { ipp += NP; tp = t + 1; }
      if (lane == 0)
        __builtin_amdgcn_raw_buffer_store_b32(epoch - (unsigned)tp, rs_fl, 0, ((rg + ipp * n_rg) * 64 + cg * 8 + wave) * 4, 16);
    }
    // the tile to request now (block n + 2: dZ[t2 + 1], rows r2) is there if its 64 flag words, fetched a block ago, say so
    bool ready = t2 >= 0 && t2 < T - 1;
    if (ready) {
      const unsigned far = (unsigned)(t2 + 1);      // (a word of this launch is at most T - 1 behind the epoch, one of an earlier launch at least T + 2)
      ready = __all(epoch - lds_peek(lds_tile + (unsigned)(FL_OFF + (n & 1) * 256) + (unsigned)(lane * 4)) <= far);
      if (!ready) {      // not posted when the words were fetched: ask memory (rare; this wait drains the wave's queue)
#ifdef KL_STAMP
        if (blockIdx.x == STAMP_WG && threadIdx.x == 0) stamp_lds[28] += 1;
#endif
        for (unsigned spin = 0; spin < SPIN_LIMIT && !ready; ++spin) {
          const unsigned now = __hip_atomic_load(flags + (long)(r2 >> 4) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ready = __all(epoch - now <= far);
          if (!ready) {
            if ((spin & 63) == 63 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            __builtin_amdgcn_s_sleep(2);
          }
        }
        if (!ready) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the scan goes on without the tile: results invalid, flagged)
      }
    }
    // the flag words of the tile the NEXT block will request (block n + 3), a block ahead of their use, into the other slot
    // (last read during the block before this one)
    if (wave == 7 && t3 >= 0 && t3 < T - 1) {
      fl_l[((n + 1) & 1) * 64 + lane] = 0u;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      glds4_sc1_s(rs_fl, (unsigned)(lane * 4), (unsigned)((rg + ip3 * n_rg) * 256), lds_tile + (unsigned)(FL_OFF + ((n + 1) & 1) * 256));
    }
    const bool put = loaded;                 // write the next block's tile into the buffer the block before this one released
    const unsigned put_addr = put_lane + (unsigned)((buf ^ 1) * NPIECE * 1024);
    const unsigned get_soff = (unsigned)(((t2 + 1) * B + r2 + 2 * wave) * (4 * W * 2));
    const int trow_n = (t1 >= 0 ? t1 * B + r1 : t * B + r0) + 2 * wave;      // (the very last block asks for its own rows again)
    f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
    // one slot per pair of MFMAs: tile piece k out to LDS and its registers re-used for the tile after next, then the inputs
#define KL_B4_SLOT(q_, put_, get_)                                                                                                  \
  do {                                                                                                                             \
    const int kk_ = (q_) < 11 ? ((q_) & 1 ? -1 : (q_) >> 1) : ((q_) == 11 ? 6 : ((q_) == 12 ? 7 : -1));                            \
    if (kk_ >= 0) {                                                                                                                \
      if ((put_) && !(KL_BWD_VAR & 8)) KL_B4_PUT(kk_, put_addr);                   /* (KL_BWD_VAR: timing experiments only) */      \
      if ((get_) && !(KL_BWD_VAR & 16)) KL_B4_GET(kk_, src_lane0, src_lane1r, rs_own, get_soff);                                   \
    }                                                                                                                              \
    if ((q_) == 13) KL_B4_REQUEST(0, trow_n);                                                                                      \
    if ((q_) == 14) KL_B4_REQUEST(1, trow_n);                                                                                      \
    if ((q_) == 15) KL_B4_REQUEST(2, trow_n);                                                                                      \
  } while (0)
#define KL_B4_MFMA_PHASE(put_, get_)                                                                                                \
  do {                                                                                                                             \
    const unsigned char* tb = smem + (buf * NPIECE + kq4 * 16) * 1024;                                                             \
    u32x4 fr[3];      /* (fragments two k-steps ahead: a third one in flight did not fit the 216 VGPRs next to 40 accumulator registers) */ \
    fr[0] = *reinterpret_cast<const u32x4*>(tb + frag_lane);                                                                       \
    fr[1] = *reinterpret_cast<const u32x4*>(tb + frag_lane + 256);                                                                 \
    _Pragma("unroll") for (int q = 0; q < KSTEPS; ++q) {                                                                           \
      if (q + 2 < KSTEPS && !(KL_BWD_VAR & 64))                                                                                    \
        fr[(q + 2) % 3] = *reinterpret_cast<const u32x4*>(tb + (frag_lane ^ (unsigned)(64 * ((q + 2) >> 2))) + 256 * ((q + 2) & 3)); \
      __builtin_amdgcn_sched_barrier(0);                                                                                           \
      const int j = 4 * (q & 3) + (q >> 2);      /* k-steps in the order j = 4 (q & 3) + (q >> 2): one lane address per group of four */ \
      if (KL_BWD_VAR & 32) {                                                                                                       \
        asm volatile("" :: "v"(fr[q % 3]), "v"(bu[0][j]), "v"(bu[1][j]));                                                          \
      } else {                                                                                                                     \
        acc0 = mfma16(__builtin_bit_cast(bf16x8, fr[q % 3]), __builtin_bit_cast(bf16x8, bu[0][j]), acc0);                          \
        acc1 = mfma16(__builtin_bit_cast(bf16x8, fr[q % 3]), __builtin_bit_cast(bf16x8, bu[1][j]), acc1);                          \
      }                                                                                                                            \
      __builtin_amdgcn_sched_barrier(0);                                                                                           \
      KL_B4_SLOT(q, put_, get_);                                                                                                   \
      __builtin_amdgcn_sched_barrier(0);                                                                                           \
    }                                                                                                                              \
  } while (0)
    if (t < T - 1 && put && ready) {
      KL_B4_MFMA_PHASE(true, true);            // (the steady state: straight-line)
    } else if (t < T - 1) {
      KL_B4_MFMA_PHASE(put, ready);
    } else {
#pragma unroll
      for (int q = 0; q < KSTEPS; ++q) KL_B4_SLOT(q, put, ready);
    }
    loaded = ready;
    {
      float* zw = zt + wave * B3_ZT_WAVE + ((lane >> 4) * 4) * B3_ZT_ROW + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) { zw[r * B3_ZT_ROW] = acc0[r]; zw[r * B3_ZT_ROW + 16] = acc1[r]; }
    }
    SSTAMP(19);
    __syncthreads();
    SSTAMP(20);
    // ---- epilogue: thread = (row er, units eu and eu + 1)
    {
      const float* zr = zt + (eu >> 5) * 4 * B3_ZT_WAVE + er * B3_ZT_ROW + (eu & 31);
      const float2 p0 = *reinterpret_cast<const float2*>(zr), p1 = *reinterpret_cast<const float2*>(zr + B3_ZT_WAVE);
      const float2 p2 = *reinterpret_cast<const float2*>(zr + 2 * B3_ZT_WAVE), p3 = *reinterpret_cast<const float2*>(zr + 3 * B3_ZT_WAVE);
      const float rec[2] = {(p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y)};
      SSTAMP(21);
      float dzv[2][4];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const unsigned g01 = gin[2 * c], g23 = gin[2 * c + 1];
        const float gi = u2f(g01 << 16), gf = u2f(g01 & 0xffff0000u), gg = u2f(g23 << 16), go = u2f(g23 & 0xffff0000u);
        const float cp = c ? u2f(cpin & 0xffff0000u) : u2f(cpin << 16);
        const float dhi = c ? u2f(dhin & 0xffff0000u) : u2f(dhin << 16);
        const float dh = dhi * (((mbits >> c) & 1u) ? mscale : 0.f) + rec[c];
        const float tc = fast_tanh(ccur[0][c]);
        const float dc = dh * go * (1.f - tc * tc) + dcr[0][c];
        dcr[0][c] = dc * gf;
        ccur[0][c] = cp;
        const float d_o = dh * tc, d_i = dc * gg, d_g = dc * gi, d_f = dc * cp;
        dzv[c][0] = d_i * gi * (1.f - gi);
        dzv[c][1] = d_f * gf * (1.f - gf);
        dzv[c][2] = d_g * (1.f - gg * gg);
        dzv[c][3] = d_o * go * (1.f - go);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<unsigned*>(pub + (g * 16 + er) * 64 + eu) = (unsigned)f2bf(dzv[0][g]) | ((unsigned)f2bf(dzv[1][g]) << 16);
    }
    SSTAMP(22);
    __syncthreads();
    SSTAMP(23);
    // ---- publish dZ[t] (all eight waves, one 16-byte store per lane) and the bias gradient: column (gate, unit) = tid & 255 of
    // the staged tile, rows 8 (tid >> 8) .. + 8 (what was stored: the bf16 values)
    {
      const uint4 v = *reinterpret_cast<const uint4*>(pub_rd);
      const unsigned soff = (unsigned)((t * B + r0) * (4 * W * 2));
      if (local) __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs_own, (int)pub_lane, (int)soff, 0);
      else __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, rs_own, (int)pub_lane, (int)soff, 16);
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) sum += bf2f(db_col[r * 64]);
      dbsum += sum;
    }
    SSTAMP(24);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float d0 = dcr[0][c], c0 = ccur[0][c];
#pragma unroll
      for (int p = 0; p + 1 < NP; ++p) { dcr[p][c] = dcr[p + 1][c]; ccur[p][c] = ccur[p + 1][c]; }
      dcr[NP - 1][c] = d0; ccur[NP - 1][c] = c0;
    }
    mbits = (mbits >> 2) | ((mbits & 3u) << (2 * (NP - 1)));
    t0 = t1; ip0 = ip1; t1 = t2; ip1 = ip2; t2 = t3; ip2 = ip3;
    next_pos(t2, ip2, t3, ip3);
  }
  SSTAMP_FLUSH();
  if (a.db) atomicAdd(a.db + (long)((tid & 255) >> 6) * W + u0 + (tid & 63), dbsum);
}
#endif      // KL_NO_REGTILE

}  // namespace

// ---------------------------------------------------------------- helpers of the gate-interleaved layouts
namespace {

// out[r][u*4 + g] = in[r][g*W + u] (+ bias[g*W + u]); f32 rows of 4W columns
__global__ void permute_gate_cols_f32_kernel(const float* __restrict__ in, const float* __restrict__ bias, float* __restrict__ out, long rows, int W) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one output quad (row, unit)
  if (i >= rows * W) return;
  const long r = i / W;
  const int u = (int)(i % W);
  const float* p = in + r * 4 * W + u;
  float4 v = float4{p[0], p[W], p[2 * W], p[3 * W]};
  if (bias) { v.x += bias[u]; v.y += bias[W + u]; v.z += bias[2 * W + u]; v.w += bias[3 * W + u]; }
  *reinterpret_cast<float4*>(out + i * 4) = v;
}

// out[(u*4 + g)][k] = in[(g*W + u)][k]; bf16 rows of K elements (16-byte pieces)
__global__ void permute_gate_rows_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int W, int K) {
  const int pieces = K / 8;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)4 * W * pieces) return;
  const int orow = (int)(i / pieces), pc = (int)(i % pieces);
  const int u = orow >> 2, g = orow & 3;
  *reinterpret_cast<uint4*>(out + (long)orow * K + pc * 8) = *reinterpret_cast<const uint4*>(in + ((long)g * W + u) * K + pc * 8);
}

// ids_tm[t][b] = (idx[b][t] * 4W * 4, ctx[b][t][0] * 4W * 4): table-row byte offsets, time-major; block T repeats block T-1
__global__ void ids_tm_kernel(const int* __restrict__ idx, const int* __restrict__ ctx, int n_ctx, int B, int T, unsigned row_bytes,
                              int V, int ctx_vocab, int* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)(T + 1) * B) return;
  int t = (int)(i / B);
  const int b = (int)(i % B);
  if (t >= T) t = T - 1;
  int id = idx[(long)b * T + t];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);
  int c = n_ctx > 0 ? ctx[((long)b * T + t) * n_ctx] : 0;
  c = c < 0 ? 0 : (c >= ctx_vocab ? ctx_vocab - 1 : c);
  out[i * 2] = (int)((unsigned)id * row_bytes);
  out[i * 2 + 1] = (int)((unsigned)c * row_bytes);
}

// Gate inputs of layer 0 as P rows for the second-generation forward scan when there are SEVERAL context variables (its
// table mode adds one context row to the character row; rating.py:118-122 allows any number): P0[t * B + b][u][gate] =
// EKp[idx] + sum over n of CtxKp_n[ctx_n], rows time-major, gate-interleaved, bf16 -- what proj_ws_kernel leaves for the
// layers above.  16 bytes in per table and cell, 8 bytes out.
struct KlCtxTabs { const float* t[8]; };
__global__ void p_gather_il_kernel(const float* __restrict__ EKp, const KlCtxTabs tabs, int n_ctx, const int* __restrict__ idx,
                                   const int* __restrict__ ctx, int B, int T, int W, int V, int ctx_vocab, bf16_t* __restrict__ out) {
  const long total = (long)T * B * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / W;
    const int u = (int)(i - row * W);
    const int t = (int)(row / B), b = (int)(row - (long)t * B);
    const long src = (long)b * T + t;
    int id = idx[src];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    f32x4 v = *reinterpret_cast<const f32x4*>(EKp + ((long)id * W + u) * 4);
    for (int n = 0; n < n_ctx; ++n) {
      int c = ctx[src * n_ctx + n];
      c = c < 0 ? 0 : (c >= ctx_vocab ? ctx_vocab - 1 : c);
      v += *reinterpret_cast<const f32x4*>(tabs.t[n] + ((long)c * W + u) * 4);
    }
    uint2 o;
    o.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
    o.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(out + i * 4) = o;
  }
}

}  // namespace

int kl_launch_p_gather_il(const float* EKp, const float* const* CtxKp, int n_ctx, const int* idx, const int* ctx, int B, int T, int W,
                          int V, int ctx_vocab, bf16_t* out, hipStream_t stream) {
  if (n_ctx < 0 || n_ctx > 8 || !EKp || !idx || !out || (n_ctx > 0 && (!ctx || !CtxKp))) return KL_ERR_ARG;
  KlCtxTabs tabs;
  memset(&tabs, 0, sizeof(tabs));
  for (int n = 0; n < n_ctx; ++n) tabs.t[n] = CtxKp[n];
  long g = ((long)T * B * W + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(p_gather_il_kernel, dim3((unsigned)g), dim3(256), 0, stream, EKp, tabs, n_ctx, idx, ctx, B, T, W, V, ctx_vocab, out);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

int kl_launch_permute_gate_cols_f32(const float* in, const float* bias, float* out, long rows, int W, hipStream_t stream) {
  const long n = rows * W;
  hipLaunchKernelGGL(permute_gate_cols_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, bias, out, rows, W);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

int kl_launch_permute_gate_rows_bf16(const bf16_t* in, bf16_t* out, int W, int K, hipStream_t stream) {
  if (K & 7) return KL_ERR_SHAPE;
  const long n = (long)4 * W * (K / 8);
  hipLaunchKernelGGL(permute_gate_rows_bf16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, in, out, W, K);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

int kl_launch_ids_tm(const int* idx, const int* ctx, int n_ctx, int B, int T, int W, int V, int ctx_vocab, int* out, hipStream_t stream) {
  const long n = (long)(T + 1) * B;
  hipLaunchKernelGGL(ids_tm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, idx, ctx, n_ctx, B, T,
                     (unsigned)(4 * W * 4), V, ctx_vocab, out);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

// ---------------------------------------------------------------- launchers
// Grid plan of the second-generation wide scans: every workgroup serves exactly NP phases of `rows` rows per step.
// Returns NP (2..4) or 0 = not applicable (the first generation takes the shape).
int kl_scan_wide2_phases(int B, int T, int W, int rows, int max_np) {
  if (W != 512 || B < 1 || T < 1 || (B % rows)) return 0;      // (one tile row = one 1 KiB DMA piece: width 512)
  int cus = 256;
  {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      cus = prop.multiProcessorCount > 256 ? 256 : prop.multiProcessorCount;
  }
  const int n_ph = B / rows, col_groups = W / 64;
  int g = cus / col_groups;
  if (g < 1) return 0;
  if (g > n_ph) g = n_ph;
  if (n_ph % g) return 0;
  const int np = n_ph / g;
  if (np < 2 || np > max_np) return 0;
  if ((long)T * B * 4 * W * 2 > 0xfffffff0L) return 0;      // unsigned 32-bit buffer offsets (gate rows / dZ)
  return np;
}

int kl_launch_scan_fwd_wide2(KlScanFwdWide a, int rows, hipStream_t stream) {
  const int W = a.W;
  // (16-row phases: up to five per workgroup and step -- 2560 streams; 32-row phases: up to four)
  const int np = kl_scan_wide2_phases(a.B, a.T, W, rows, rows == 16 ? 5 : 4);
  if (!np || (rows != 16 && rows != 32)) return KL_ERR_SHAPE;
  if (!a.sentinel || a.HT || a.HdT) return KL_ERR_SHAPE;
  const bool tab = a.P == nullptr;
  if (tab && (!a.EK || !a.ids_tm || a.n_ctx > 1)) return KL_ERR_ARG;
  a.n_rb = a.B / rows;                 // (phases)
  a.n_rg = a.n_rb / np;
  const int col_groups = W / 64;
  dim3 grid(8 * col_groups * ((a.n_rg + 7) / 8)), block(1024);
  const int nb = rows / 16;
  const size_t lds = (size_t)fwd2_lds_bytes(W / 32, nb, tab);
#define KL_F2_CASE(KS, NB_, NP_, TAB_)                                                                                      \
  do {                                                                                                                      \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_scan_fwd_wide2_kernel<NB_, NP_, TAB_>),                \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return KL_ERR_LAUNCH;     \
    hipLaunchKernelGGL((lstm_scan_fwd_wide2_kernel<NB_, NP_, TAB_>), grid, block, lds, stream, a);                      \
  } while (0)
#define KL_F2_NP(KS, NB_, TAB_)                                                    \
  do {                                                                             \
    if (np == 2) KL_F2_CASE(KS, NB_, 2, TAB_);                                     \
    else if (np == 3) KL_F2_CASE(KS, NB_, 3, TAB_);                                \
    else if (np == 4) KL_F2_CASE(KS, NB_, 4, TAB_);                                \
    else if (NB_ == 1) KL_F2_CASE(KS, 1, 5, TAB_);                                 \
    else return KL_ERR_SHAPE;                                                      \
  } while (0)
#define KL_F2_TAB(KS, NB_)                          \
  do {                                              \
    if (tab) KL_F2_NP(KS, NB_, true);               \
    else KL_F2_NP(KS, NB_, false);                  \
  } while (0)
  if (nb == 2) KL_F2_TAB(16, 2);
  else KL_F2_TAB(16, 1);
#undef KL_F2_TAB
#undef KL_F2_NP
#undef KL_F2_CASE
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}


// gate inputs of a layer above the first for all rows at once (proj_ws_kernel); KL_ERR_SHAPE = not applicable
int kl_launch_proj_ws(const bf16_t* X, const bf16_t* KTp, const float* bp, bf16_t* P, long M, int W, unsigned* status,
                      hipStream_t stream) {
  if (W != 512 || M < 32 * 32 || (M % 32) != 0 || M * 2048L * 2 > 0xfffffff0L) return KL_ERR_SHAPE;
  KlProjWs a;
  a.X = X; a.KTp = KTp; a.bp = bp; a.P = P; a.M = (int)M; a.n_rg = 32; a.status = status;
  const size_t lds = (size_t)2 * 32 * 1024;
  // (eight waves of 32 columns: half the LDS reads per MFMA; KL_PROJ_WS8 = 0: the 16-wave form)
  static const int eight = [] { const char* e = getenv("KL_PROJ_WS8"); return e ? atoi(e) : 1; }();      // (0: the 16-wave form; 2: with staged stores -- measured equal, 23.73 ms per step both, 23.91 with 16 waves)
  if (eight == 2) {
    static KlLdsGrant grant;
    const size_t lds8 = lds + 32 * 528;
    if (kl_grant_lds(grant, reinterpret_cast<const void*>(&proj_ws8_kernel<true>), lds8)) return KL_ERR_LAUNCH;
    hipLaunchKernelGGL(proj_ws8_kernel<true>, dim3(8 * 8 * ((a.n_rg + 7) / 8)), dim3(512), lds8, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
  }
  if (eight == 1) {
    static KlLdsGrant grant;
    if (kl_grant_lds(grant, reinterpret_cast<const void*>(&proj_ws8_kernel<false>), lds)) return KL_ERR_LAUNCH;
    hipLaunchKernelGGL(proj_ws8_kernel<false>, dim3(8 * 8 * ((a.n_rg + 7) / 8)), dim3(512), lds, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
  }
  static KlLdsGrant grant16;
  if (kl_grant_lds(grant16, reinterpret_cast<const void*>(&proj_ws_kernel), lds)) return KL_ERR_LAUNCH;
  hipLaunchKernelGGL(proj_ws_kernel, dim3(8 * 8 * ((a.n_rg + 7) / 8)), dim3(1024), lds, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

// training only: logits, softmax, cross-entropy and its gradient in one pass (logits_ce_ws_kernel); KL_ERR_SHAPE = not applicable.
// The caller still reduces rowstat into the loss accumulators (kl_launch_rowstat_reduce).
int kl_launch_logits_ce_ws(const bf16_t* X, const bf16_t* E, const int* tgt, bf16_t* dlogits, float* rowstat, int B, int T, int W,
                           int V, long ld_dl, float inv_count, int last_only, hipStream_t stream) {
  const long M = (long)B * T;
  if (W != 512 || V != 256 || ld_dl != 256 || M < 32 * 256 || (M % 32) != 0 || M * 512L * 2 > 0xfffffff0L) return KL_ERR_SHAPE;
  KlLogitsCe a;
  a.X = X; a.E = E; a.tgt = tgt; a.dlogits = dlogits; a.rowstat = rowstat;
  a.M = (int)M; a.B = B; a.T = T; a.n_rg = 256; a.last_only = last_only; a.inv_count = inv_count;
  const size_t lds = (size_t)2 * 32 * 1024 + 32 * (256 + 4) * 4;
  static KlLdsGrant grant;
  if (kl_grant_lds(grant, reinterpret_cast<const void*>(&logits_ce_ws_kernel), lds)) return KL_ERR_LAUNCH;
  hipLaunchKernelGGL(logits_ce_ws_kernel, dim3(a.n_rg), dim3(1024), lds, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

// In front of a flag-mode backward scan: the epoch moves on by `step` (> the scan's T + 1, so that no word a former
// launch left can pass for one of this launch); near the end of the 32-bit range everything starts over.
__global__ void scan_epoch_kernel(unsigned* flags, int n_flags, unsigned* epoch, unsigned step) {
  __shared__ unsigned old;
  if (threadIdx.x == 0) old = *epoch;
  __syncthreads();
  const bool wrap = old > 0xF0000000u;
  if (wrap)
    for (int i = threadIdx.x; i < n_flags; i += blockDim.x) flags[i] = 0u;
  __syncthreads();
  if (threadIdx.x == 0) *epoch = (wrap ? 0u : old) + step;
}
int kl_launch_scan_epoch(unsigned* flags, int n_flags, unsigned* epoch, unsigned step, hipStream_t stream) {
  hipLaunchKernelGGL(scan_epoch_kernel, dim3(1), dim3(1024), 0, stream, flags, n_flags, epoch, step);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

// dH = dlogits . E as bf16 for a training window (dh_ws_kernel: V padded to 256, width 512); KL_ERR_SHAPE = not applicable
int kl_launch_dh_ws(const bf16_t* dlogits, const bf16_t* ET, bf16_t* dH, long M, int W, int Vp, hipStream_t stream) {
  if (W != 512 || Vp != 256 || M < 32 * 128 || (M % 32) != 0 || M * 512L * 2 > 0xfffffff0L) return KL_ERR_SHAPE;
  KlDhWs a;
  a.X = dlogits; a.ET = ET; a.dH = dH; a.M = (int)M; a.n_rg = 128;
  const size_t lds = (size_t)2 * 32 * 512;
  hipLaunchKernelGGL(dh_ws_kernel, dim3(8 * 2 * ((a.n_rg + 7) / 8)), dim3(1024), lds, stream, a);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

int kl_launch_scan_bwd_wide2(KlScanBwd a, hipStream_t stream) {
  const int W = a.W;
  const int np = kl_scan_wide2_phases(a.B, a.T, W, 16, 6);
  if (!np || a.L != 1 || a.dZT || a.T < 3) return KL_ERR_SHAPE;
  if (a.flags ? (np < 3 || !a.epoch) : a.sentinel != 2) return KL_ERR_SHAPE;      // (flags: see the kernel; two blocks per step leave them no time)
  a.n_rb = a.B / 16;
  a.n_rg = a.n_rb / np;
  const int col_groups = W / 64;
  dim3 grid(8 * col_groups * ((a.n_rg + 7) / 8)), block(1024);
  const size_t lds = (size_t)bwd2_lds_bytes(W / 32);
#define KL_B2_CASE1(KS, NP_, FL_)                                                                                          \
  do {                                                                                                                      \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_scan_bwd_wide2_kernel<NP_, FL_>),                      \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return KL_ERR_LAUNCH;     \
    hipLaunchKernelGGL((lstm_scan_bwd_wide2_kernel<NP_, FL_>), grid, block, lds, stream, a);                            \
  } while (0)
#define KL_B2_CASE(KS, NP_)                   \
  do {                                        \
    if (a.flags) KL_B2_CASE1(KS, NP_, true);  \
    else KL_B2_CASE1(KS, NP_, false);         \
  } while (0)
#define KL_B2_NP(KS)                                                                             \
  do {                                                                                          \
    switch (np) {                                                                               \
      case 2: KL_B2_CASE(KS, 2); break;                                                         \
      case 3: KL_B2_CASE(KS, 3); break;                                                         \
      case 4: KL_B2_CASE(KS, 4); break;                                                         \
      case 5: KL_B2_CASE(KS, 5); break;                                                         \
      default: KL_B2_CASE(KS, 6); break;                                                        \
    }                                                                                           \
  } while (0)
  KL_B2_NP(16);
#undef KL_B2_NP
#undef KL_B2_CASE
#undef KL_B2_CASE1
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

// register-landing tiles (lstm_scan_bwd_regtile_kernel): flags only, from this many blocks per workgroup and step
#ifdef KL_NO_REGTILE
int kl_scan_bwd_regtile_min_np() { return 1 << 30; }
int kl_launch_scan_bwd_regtile(KlScanBwd, hipStream_t) { return KL_ERR_SHAPE; }
#else
int kl_scan_bwd_regtile_min_np() { return 5; }
int kl_launch_scan_bwd_regtile(KlScanBwd a, hipStream_t stream) {
  const int W = a.W;
  const int np = kl_scan_wide2_phases(a.B, a.T, W, 16, 6);
  if (np < kl_scan_bwd_regtile_min_np() || W != 512 || a.L != 1 || a.dZT || a.T < 3 || !a.flags || !a.epoch) return KL_ERR_SHAPE;
  if (!a.Cb || !a.dHb) return KL_ERR_ARG;
  a.n_rb = a.B / 16;
  a.n_rg = a.n_rb / np;
  dim3 grid(8 * (W / 64) * ((a.n_rg + 7) / 8)), block(512);
  const size_t lds = (size_t)bwd3_lds_bytes();
#define KL_B4_CASE(NP_)                                                                                                     \
  do {                                                                                                                      \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_scan_bwd_regtile_kernel<NP_>),                             \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return KL_ERR_LAUNCH;     \
    hipLaunchKernelGGL((lstm_scan_bwd_regtile_kernel<NP_>), grid, block, lds, stream, a);                                   \
  } while (0)
  if (np == 5) KL_B4_CASE(5);
  else KL_B4_CASE(6);
#undef KL_B4_CASE
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}
#endif      // KL_NO_REGTILE

#ifdef KL_STAMP
extern "C" int kl_test_scan2_stamps(unsigned long long* out, int reset) {
  if (reset) {
    unsigned long long z[32] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(kl_scan2_stamps), z, sizeof(z)) == hipSuccess ? 0 : KL_ERR_LAUNCH;
  }
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(kl_scan2_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : KL_ERR_LAUNCH;
}
#endif
