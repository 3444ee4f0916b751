// Persistent LSTM scans for gfx950: the whole T-step recurrence of every layer in
// ONE launch, weights resident in registers.
//
// Why: a launch per time step re-streams the layer's 4W x W weight slice from
// beyond L2 on every step and pays launch ramp + kernarg fetch each time (measured
// 10-12 us per step at W=512).  Here every workgroup keeps its slice of the
// recurrent (and, above layer 0, input) kernel in VGPRs for all T steps and the
// only per-step traffic is the 16 x K state tile it contracts with.
//
// Decomposition: workgroup (layer l, unit group ug of 16 hidden units, row group
// rg); rows (independent stateful streams) come in blocks of 16 = one MFMA row
// tile.  Inside a workgroup the contraction dimension K is split over the four
// waves (forward: each wave holds all four gate tiles for its K quarter; backward:
// the K quarter IS one gate's column range), every wave loads exactly the fragment
// bytes it multiplies -- straight into registers, nothing staged -- and the four
// partial tiles meet in LDS where the gate math runs.
//
// Streams never interact, so synchronisation is per (layer, row block, step): the
// W/16 workgroups that produce h_l[t] for a row block each add 1 to a counter once
// their slice is published; consumers (same layer at t+1, layer l+1 at t) poll it.
// There is no grid-wide barrier anywhere.
//
// Hand-off protocol (MI355X_MICROARCH.md, valid form "ONE lane of each storing
// workgroup ... agent-scope atomic add"): payload stores are write-through (sc1),
// every storing wave drains vmcnt(0), workgroup barrier, one lane adds to the
// counter; the consumer's lane 0 polls the counter with sc1 loads, workgroup
// barrier, then EVERY load of handed-off bytes is an sc1 buffer load.  Counters are
// zeroed by a memset node before each launch.  All spins are bounded: on timeout
// (or if another workgroup has raised the abort word) the kernel drains and the
// host sees a non-zero status word instead of a hang.
// Residency: at most 512 workgroups of 256 threads (<= 256 VGPRs, 18 KiB LDS), i.e.
// two per CU on a full chip; a grid that oversubscribes the chip (tried: 1024)
// times out cleanly through the bounded spins.
//
// The above describes the fused thin scans (lstm_scan_fwd_kernel / lstm_scan_bwd_kernel, few row blocks).
// For many streams the layer-sequential WIDE scans further down take over: one layer per launch,
// 1024-thread workgroups of 64 hidden units that fetch the state tile once and share it through LDS,
// hand-off by data sentinels instead of counters (re-armed inside the backward scan), the next row
// block's tile prefetched by LDS-DMA, optional XCD-local publishes -- see the comments at those kernels
// and DESIGN.md section 4.
#include <stdlib.h>
#include <string.h>

#include "kl_common.h"
#include "kl_kernels.h"

namespace {

#include "kl_scan_common.h"

// ---------------------------------------------------------------- forward scan
// block = 256 threads; wave w contracts over K quarter w (KQ = W/128 k-steps of 32
// per operand) for all 4 gates x 16 units.  MAXRB = row blocks served per step.
// IN = false: single-layer launches whose input side arrives precomputed in P1 (the layer-sequential path
// for width 1024, where U alone takes 128 of the 256 registers): no input-weight registers at all.
template <int KSTEPS, int MAXRB, bool IN = true>
__global__ __launch_bounds__(256, 2) void lstm_scan_fwd_kernel(const KlScanFwd a) {
  // K is split over KW waves (four; fewer for widths below 128, where the other waves contribute zero tiles)
  constexpr int KW = KSTEPS < 4 ? KSTEPS : 4;
  constexpr int KQ = KSTEPS / KW;
  static_assert(KSTEPS % KW == 0, "K split");
  constexpr int W = KSTEPS * 32;
  constexpr int NUG = W / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool kactive = wave < KW;
  const int kw = kactive ? wave : 0;       // (idle waves read wave 0's slice and discard the product)
  int id = blockIdx.x;
  const int n_rg = a.n_rg, n_rb = a.n_rb, B = a.B, T = a.T;
  const int l = id / (NUG * n_rg);
  id -= l * NUG * n_rg;
  const int ug = id / n_rg, rg = id % n_rg;
  const int u0 = ug * 16;
  const bool has_in = IN && l > 0;

  __shared__ float zt[4][4][16][17];   // [wave][gate][row][unit] partial tiles
  __shared__ int ok_flag;

  // ---- resident weights as B fragments: gate g, this wave's K quarter
  const int kq = (lane >> 4) * 8;
  uint4 bu[4][KQ], bk[IN ? 4 : 1][IN ? KQ : 1];
  {
    const bf16_t* UT = a.UT[l];
    const bf16_t* KT = a.KT[l];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const long wrow = ((long)g * W + u0 + (lane & 15)) * W + (kw * KQ) * 32 + kq;
#pragma unroll
      for (int j = 0; j < KQ; ++j) {
        bu[g][j] = *reinterpret_cast<const uint4*>(UT + wrow + j * 32);
        if (IN) bk[g][j] = has_in ? *reinterpret_cast<const uint4*>(KT + wrow + j * 32) : uint4{0, 0, 0, 0};
      }
    }
  }
  // epilogue thread = (row er, unit eu)
  const int er = tid >> 4, eu = tid & 15;
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if (has_in) {
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = a.bias[l][(long)g * W + u0 + eu];
  }
  float* Cl = a.C[l];
  bf16_t* Hl = a.H[l];
  bf16_t* Gl = a.G[l];
  bf16_t* Hdl = a.Hd[l];
  const float* maskl = a.mask[l];
  const float* P1 = a.P1;
  unsigned* status = a.status;
  float c_reg[MAXRB];
#pragma unroll
  for (int i = 0; i < MAXRB; ++i) {
    const int rb = rg + i * n_rg;
    const int row = min(rb * 16 + er, B - 1);
    c_reg[i] = (rb < n_rb) ? Cl[(long)row * W + u0 + eu] : 0.f;
  }
  const long BW = (long)B * W;
  const bf16_t* Hin = has_in ? (a.Hd[l - 1] ? a.Hd[l - 1] : a.H[l - 1] + BW) : Hl;   // input rows of step t at block t
  const __amdgpu_buffer_rsrc_t rs_h = make_rsrc(Hl, (long)(T + 1) * BW * 2);
  const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(Hin, (long)T * BW * 2);
  unsigned* cnt_own = a.counters + (long)l * n_rb * T;
  unsigned* cnt_in = a.counters + (long)(has_in ? l - 1 : 0) * n_rb * T;
  bool alive = true;
  SSTAMP_INIT(gridDim.x - 1);

  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int i = 0; i < MAXRB; ++i) {
      const int rb = rg + i * n_rg;
      if (rb >= n_rb) continue;
      const int r0 = rb * 16;
      const int erow = min(r0 + er, B - 1);
      SSTAMP(0);
      // gate inputs that do not depend on the hand-off: issue first
      float zin[4];
      if (!has_in) {
        const float* p = P1 + ((long)t * B + erow) * 4 * W + u0 + eu;
#pragma unroll
        for (int g = 0; g < 4; ++g) zin[g] = p[(long)g * W];
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) zin[g] = bias4[g];
      }
      float mk = 1.f;
      if (maskl) mk = maskl[(long)erow * W + u0 + eu];
      // ---- wait for h_{l-1}[t] and h_l[t-1] of this row block
      if (tid == 0) {
        bool ok = alive;
        if (ok && has_in) ok = poll_counter(cnt_in + (long)rb * T + t, NUG, status);
        if (ok && t > 0) ok = poll_counter(cnt_own + (long)rb * T + (t - 1), NUG, status);
        ok_flag = ok ? 1 : 0;
      }
      SSTAMP(1);
      __syncthreads();
      SSTAMP(2);
      alive = ok_flag != 0;
      // ---- this wave's fragments of the 16 x K state tile, write-through reads
      const int arow = min(r0 + (lane & 15), B - 1);
      const unsigned abase = (unsigned)((((long)t * B + arow) * W + (kw * KQ) * 32 + kq) * 2);
      uint4 ah[KQ], ax[IN ? KQ : 1];
      if (alive) {
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
          if (IN) ax[j] = has_in ? load16_sc1(rs_in, abase + j * 64) : uint4{0, 0, 0, 0};
          ah[j] = load16_sc1(rs_h, abase + j * 64);
        }
      } else {
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
          if (IN) ax[j] = uint4{0, 0, 0, 0};
          ah[j] = uint4{0, 0, 0, 0};
        }
      }
      SSTAMP(3);
      f32x4 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (IN && has_in) {
#pragma unroll
        for (int j = 0; j < (IN ? KQ : 1); ++j) {
          frag16 fa;
          fa.u = ax[j];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            frag16 fb;
            fb.u = bk[g][j];
            acc[g] = mfma16(fa.v, fb.v, acc[g]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < KQ; ++j) {
        frag16 fa;
        fa.u = ah[j];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          frag16 fb;
          fb.u = bu[g][j];
          acc[g] = mfma16(fa.v, fb.v, acc[g]);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) zt[wave][g][(lane >> 4) * 4 + r][lane & 15] = kactive ? acc[g][r] : 0.f;
      SSTAMP(5);
      __syncthreads();
      SSTAMP(6);
      // ---- gates for (row er, unit eu)
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) z[g] = zin[g] + zt[0][g][er][eu] + zt[1][g][er][eu] + zt[2][g][er][eu] + zt[3][g][er][eu];
      const float gi = fast_sigmoid(z[0]), gf = fast_sigmoid(z[1]), gg = fast_tanh(z[2]), go = fast_sigmoid(z[3]);
      const float c = gf * c_reg[i] + gi * gg;
      c_reg[i] = c;
      const float h = go * fast_tanh(c);
      const bool row_ok = (r0 + er) < B;
      SSTAMP(7);
      // publish h (and its masked copy) write-through, two units per 4-byte store
      const unsigned hb = f2bf(h), hdb = f2bf(h * mk);
      const unsigned hb_n = __shfl_xor(hb, 1), hdb_n = __shfl_xor(hdb, 1);
      const long orow = (long)t * B + r0 + er;
      if (row_ok && alive && (eu & 1) == 0) {
        __hip_atomic_store(reinterpret_cast<unsigned*>(Hl + (orow + B) * W + u0 + eu), hb | (hb_n << 16), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        if (Hdl)
          __hip_atomic_store(reinterpret_cast<unsigned*>(Hdl + orow * W + u0 + eu), hdb | (hdb_n << 16), __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_AGENT);
      }
      SSTAMP(8);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      SSTAMP(9);
      __syncthreads();   // also orders the zt reads above before the next iteration's writes
      if (tid == 0) __hip_atomic_fetch_add(cnt_own + (long)rb * T + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // what only later launches read (cell state, gate activations) is stored after the
      // publish: plain stores, off the hand-off chain
      if (row_ok && alive) {
        Cl[(orow + B) * W + u0 + eu] = c;
        if (Gl) {
          bf16_t* gp = Gl + orow * 4 * W + u0 + eu;
          gp[0] = f2bf(gi);
          gp[W] = f2bf(gf);
          gp[2 * W] = f2bf(gg);
          gp[3 * W] = f2bf(go);
        }
      }
      SSTAMP(10);
    }
  }
  SSTAMP_FLUSH();
}

// ---------------------------------------------------------------- forward scan, split precision (rating)
// The inference twin of lstm_scan_fwd_kernel for rate / rate2 / test windows: every
// operand is a bf16 (hi, lo) pair and every contraction three MFMAs (hi.hi + lo.hi +
// hi.lo), i.e. ~f32 accuracy with f32 accumulation.  The state tile is exchanged as two
// bf16 planes (Xhi, Xlo); the f32 outputs the logits need are stored off the hand-off
// chain.  Weights take 4 x the registers of the training kernel (U, K) x (hi, lo):
// one 256-thread workgroup per CU.
template <int KSTEPS, int MAXRB>
__global__ __launch_bounds__(256, 1) void lstm_scan_fwd_split_kernel(const KlScanFwdSplit a) {
  // K is split over KW waves (four; fewer for widths below 128, where the other waves contribute zero tiles)
  constexpr int KW = KSTEPS < 4 ? KSTEPS : 4;
  constexpr int KQ = KSTEPS / KW;
  static_assert(KSTEPS % KW == 0, "K split");
  constexpr int W = KSTEPS * 32;
  constexpr int NUG = W / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int id = blockIdx.x;
  const int n_rg = a.n_rg, n_rb = a.n_rb, B = a.B, T = a.T;
  const int l = id / (NUG * n_rg);
  id -= l * NUG * n_rg;
  const int ug = id / n_rg, rg = id % n_rg;
  const int u0 = ug * 16;
  const bool has_in = l > 0;
  const bool kactive = wave < KW;
  const int kw = kactive ? wave : 0;       // (idle waves read wave 0's slice and discard the product)

  __shared__ float zt[4][4][16][17];
  __shared__ int ok_flag;

  const int kq = (lane >> 4) * 8;
  uint4 bu[2][4][KQ], bk[2][4][KQ];     // [hi/lo][gate][k-step]
  {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const long wrow = ((long)g * W + u0 + (lane & 15)) * W + (kw * KQ) * 32 + kq;
#pragma unroll
      for (int j = 0; j < KQ; ++j) {
        bu[0][g][j] = *reinterpret_cast<const uint4*>(a.UT_hi[l] + wrow + j * 32);
        bu[1][g][j] = *reinterpret_cast<const uint4*>(a.UT_lo[l] + wrow + j * 32);
        bk[0][g][j] = has_in ? *reinterpret_cast<const uint4*>(a.KT_hi[l] + wrow + j * 32) : uint4{0, 0, 0, 0};
        bk[1][g][j] = has_in ? *reinterpret_cast<const uint4*>(a.KT_lo[l] + wrow + j * 32) : uint4{0, 0, 0, 0};
      }
    }
  }
  const int er = tid >> 4, eu = tid & 15;
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if (has_in) {
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = a.bias[l][(long)g * W + u0 + eu];
  }
  float* Cl = a.C[l];
  float* Hf = a.Hf[l];
  bf16_t* Xhi = a.Xhi[l];
  bf16_t* Xlo = a.Xlo[l];
  const float* P1 = a.P1;
  unsigned* status = a.status;
  float c_reg[MAXRB];
#pragma unroll
  for (int i = 0; i < MAXRB; ++i) {
    const int rb = rg + i * n_rg;
    const int row = min(rb * 16 + er, B - 1);
    c_reg[i] = (rb < n_rb) ? Cl[(long)row * W + u0 + eu] : 0.f;
  }
  const long BW = (long)B * W;
  const __amdgpu_buffer_rsrc_t rs_hi = make_rsrc(Xhi, (long)(T + 1) * BW * 2);
  const __amdgpu_buffer_rsrc_t rs_lo = make_rsrc(Xlo, (long)(T + 1) * BW * 2);
  // input rows of step t = outputs of the layer below at block t + 1
  const __amdgpu_buffer_rsrc_t rs_ihi = make_rsrc(has_in ? a.Xhi[l - 1] + BW : Xhi, (long)T * BW * 2);
  const __amdgpu_buffer_rsrc_t rs_ilo = make_rsrc(has_in ? a.Xlo[l - 1] + BW : Xlo, (long)T * BW * 2);
  unsigned* cnt_own = a.counters + (long)l * n_rb * T;
  unsigned* cnt_in = a.counters + (long)(has_in ? l - 1 : 0) * n_rb * T;
  bool alive = true;

  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int i = 0; i < MAXRB; ++i) {
      const int rb = rg + i * n_rg;
      if (rb >= n_rb) continue;
      const int r0 = rb * 16;
      const int erow = min(r0 + er, B - 1);
      float zin[4];
      if (!has_in) {
        const float* p = P1 + ((long)t * B + erow) * 4 * W + u0 + eu;
#pragma unroll
        for (int g = 0; g < 4; ++g) zin[g] = p[(long)g * W];
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) zin[g] = bias4[g];
      }
      if (tid == 0) {
        bool ok = alive;
        if (ok && has_in) ok = poll_counter(cnt_in + (long)rb * T + t, NUG, status);
        if (ok && t > 0) ok = poll_counter(cnt_own + (long)rb * T + (t - 1), NUG, status);
        ok_flag = ok ? 1 : 0;
      }
      __syncthreads();
      alive = ok_flag != 0;
      const int arow = min(r0 + (lane & 15), B - 1);
      const unsigned abase = (unsigned)((((long)t * B + arow) * W + (kw * KQ) * 32 + kq) * 2);
      f32x4 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (alive) {
        uint4 ahi[KQ], alo[KQ];
        if (has_in) {
#pragma unroll
          for (int j = 0; j < KQ; ++j) {
            ahi[j] = load16_sc1(rs_ihi, abase + j * 64);
            alo[j] = load16_sc1(rs_ilo, abase + j * 64);
          }
#pragma unroll
          for (int j = 0; j < KQ; ++j) {
            frag16 fh, fl;
            fh.u = ahi[j];
            fl.u = alo[j];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              frag16 wh, wl;
              wh.u = bk[0][g][j];
              wl.u = bk[1][g][j];
              acc[g] = mfma16(fl.v, wh.v, acc[g]);
              acc[g] = mfma16(fh.v, wl.v, acc[g]);
              acc[g] = mfma16(fh.v, wh.v, acc[g]);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
          ahi[j] = load16_sc1(rs_hi, abase + j * 64);
          alo[j] = load16_sc1(rs_lo, abase + j * 64);
        }
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
          frag16 fh, fl;
          fh.u = ahi[j];
          fl.u = alo[j];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            frag16 wh, wl;
            wh.u = bu[0][g][j];
            wl.u = bu[1][g][j];
            acc[g] = mfma16(fl.v, wh.v, acc[g]);
            acc[g] = mfma16(fh.v, wl.v, acc[g]);
            acc[g] = mfma16(fh.v, wh.v, acc[g]);
          }
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) zt[wave][g][(lane >> 4) * 4 + r][lane & 15] = kactive ? acc[g][r] : 0.f;
      __syncthreads();
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) z[g] = zin[g] + zt[0][g][er][eu] + zt[1][g][er][eu] + zt[2][g][er][eu] + zt[3][g][er][eu];
      const float gi = fast_sigmoid(z[0]), gf = fast_sigmoid(z[1]), gg = fast_tanh(z[2]), go = fast_sigmoid(z[3]);
      const float c = gf * c_reg[i] + gi * gg;
      c_reg[i] = c;
      const float h = go * fast_tanh(c);
      const bool row_ok = (r0 + er) < B;
      const unsigned hh = f2bf(h);
      const unsigned hl = f2bf(h - bf2f((bf16_t)hh));
      const unsigned hh_n = __shfl_xor(hh, 1), hl_n = __shfl_xor(hl, 1);
      const long orow = (long)t * B + r0 + er;
      if (row_ok && alive && (eu & 1) == 0) {
        __hip_atomic_store(reinterpret_cast<unsigned*>(Xhi + (orow + B) * W + u0 + eu), hh | (hh_n << 16), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned*>(Xlo + (orow + B) * W + u0 + eu), hl | (hl_n << 16), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(cnt_own + (long)rb * T + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (row_ok && alive) {
        Hf[(orow + B) * W + u0 + eu] = h;
        if (t == T - 1) Cl[(orow + B) * W + u0 + eu] = c;
      }
    }
  }
}

// The same scan with the hand-off carried by the DATA (the rating window at the reference's batching, 1 x 256, is one
// latency chain of T + L - 1 steps: profiles/r02_rating_window_B1_kernel_stats.csv had 4.25 us per step, of which the
// counter protocol -- drain the write-through stores, barrier, atomic add, poll, barrier, only then load -- is most).
// Blocks 1..T of the (hi, lo) planes start out as 0xFFFF halfwords; every wave loads the fragments it is going to
// multiply and repeats the loads until none of them shows the sentinel, so a step costs ONE store -> load trip.
// h = o * tanh(c) lies inside (-1, 1) and its bf16 remainder is smaller still: bit 14 of every valid halfword (the top
// exponent bit) is 0 and the check is an OR over the loaded dwords.  Block 0 (the carried-in state, any value) is not
// polled.  One workgroup barrier per step: the partial tiles alternate between two LDS buffers.
// IN = false: one layer per launch whose input side arrives precomputed in P1 (width 1024: the recurrent weights of 16
// units as (hi, lo) planes alone are 256 registers per lane; the caller runs the layers one after the other and forms
// P1 = X . K^T + b for all steps by one split-precision GEMM in between).
template <int KSTEPS, int MAXRB, bool IN = true>
__global__ __launch_bounds__(256, 1) void lstm_scan_fwd_split_sent_kernel(const KlScanFwdSplit a) {
  constexpr int KW = KSTEPS < 4 ? KSTEPS : 4;
  constexpr int KQ = KSTEPS / KW;
  static_assert(KSTEPS % KW == 0, "K split");
  constexpr int W = KSTEPS * 32;
  constexpr int NUG = W / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int id = blockIdx.x;
  const int n_rg = a.n_rg, n_rb = a.n_rb, B = a.B, T = a.T;
  const int l = id / (NUG * n_rg);
  id -= l * NUG * n_rg;
  const int ug = id / n_rg, rg = id % n_rg;
  const int u0 = ug * 16;
  const bool has_in = IN && l > 0;
  const bool kactive = wave < KW;
  const int kw = kactive ? wave : 0;

  __shared__ float zt[2][4][4][16][17];

  const int kq = (lane >> 4) * 8;
  uint4 bu[2][4][KQ], bk[2][4][IN ? KQ : 1];     // [hi/lo][gate][k-step]
  {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const long wrow = ((long)g * W + u0 + (lane & 15)) * W + (kw * KQ) * 32 + kq;
#pragma unroll
      for (int j = 0; j < KQ; ++j) {
        bu[0][g][j] = *reinterpret_cast<const uint4*>(a.UT_hi[l] + wrow + j * 32);
        bu[1][g][j] = *reinterpret_cast<const uint4*>(a.UT_lo[l] + wrow + j * 32);
        if (IN) {
          bk[0][g][j] = has_in ? *reinterpret_cast<const uint4*>(a.KT_hi[l] + wrow + j * 32) : uint4{0, 0, 0, 0};
          bk[1][g][j] = has_in ? *reinterpret_cast<const uint4*>(a.KT_lo[l] + wrow + j * 32) : uint4{0, 0, 0, 0};
        }
      }
    }
  }
  const int er = tid >> 4, eu = tid & 15;
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if (has_in) {
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = a.bias[l][(long)g * W + u0 + eu];
  }
  float* Cl = a.C[l];
  float* Hf = a.Hf[l];
  bf16_t* Xhi = a.Xhi[l];
  bf16_t* Xlo = a.Xlo[l];
  const float* P1 = a.P1;
  unsigned* status = a.status;
  float c_reg[MAXRB];
#pragma unroll
  for (int i = 0; i < MAXRB; ++i) {
    const int rb = rg + i * n_rg;
    const int row = min(rb * 16 + er, B - 1);
    c_reg[i] = (rb < n_rb) ? Cl[(long)row * W + u0 + eu] : 0.f;
  }
  const long BW = (long)B * W;
  const __amdgpu_buffer_rsrc_t rs_hi = make_rsrc(Xhi, (long)(T + 1) * BW * 2);
  const __amdgpu_buffer_rsrc_t rs_lo = make_rsrc(Xlo, (long)(T + 1) * BW * 2);
  // input rows of step t = outputs of the layer below at block t + 1 (always a polled block)
  const __amdgpu_buffer_rsrc_t rs_ihi = make_rsrc(has_in ? a.Xhi[l - 1] + BW : Xhi, (long)T * BW * 2);
  const __amdgpu_buffer_rsrc_t rs_ilo = make_rsrc(has_in ? a.Xlo[l - 1] + BW : Xlo, (long)T * BW * 2);
  bool alive = true;                 // (wave-uniform: false once this wave or any other has given up)
  int par = 0;

  // the fragments of one operand, loaded until none of them is a sentinel (`poll` false: loaded once)
  auto fetch = [&](const __amdgpu_buffer_rsrc_t& rh, const __amdgpu_buffer_rsrc_t& rl, unsigned abase, bool poll,
                   uint4 (&fh)[KQ], uint4 (&fl)[KQ]) {
    for (unsigned spin = 0;; ++spin) {
      unsigned any = 0;
#pragma unroll
      for (int j = 0; j < KQ; ++j) {
        fh[j] = load16_sc1(rh, abase + j * 64);
        fl[j] = load16_sc1(rl, abase + j * 64);
      }
#pragma unroll
      for (int j = 0; j < KQ; ++j)
        any |= fh[j].x | fh[j].y | fh[j].z | fh[j].w | fl[j].x | fl[j].y | fl[j].z | fl[j].w;
      if (!poll || !alive || __all((any & 0x40004000u) == 0)) break;
      if ((spin & 63) == 63 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) alive = false;
      if (spin >= SPIN_LIMIT) {
        __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        alive = false;
      }
    }
  };

  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int i = 0; i < MAXRB; ++i) {
      const int rb = rg + i * n_rg;
      if (rb >= n_rb) continue;
      const int r0 = rb * 16;
      const int erow = min(r0 + er, B - 1);
      float zin[4];
      if (!has_in) {
        const float* p = P1 + ((long)t * B + erow) * 4 * W + u0 + eu;
#pragma unroll
        for (int g = 0; g < 4; ++g) zin[g] = p[(long)g * W];
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) zin[g] = bias4[g];
      }
      const int arow = min(r0 + (lane & 15), B - 1);
      const unsigned abase = (unsigned)((((long)t * B + arow) * W + (kw * KQ) * 32 + kq) * 2);
      f32x4 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      uint4 ahi[KQ], alo[KQ];
      if (IN && has_in) {      // (the layer below runs ahead: normally there at the first look)
        fetch(rs_ihi, rs_ilo, abase, true, ahi, alo);
#pragma unroll
        for (int j = 0; j < (IN ? KQ : 1); ++j) {
          frag16 fh, fl;
          fh.u = ahi[j];
          fl.u = alo[j];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            frag16 wh, wl;
            wh.u = bk[0][g][j];
            wl.u = bk[1][g][j];
            acc[g] = mfma16(fl.v, wh.v, acc[g]);
            acc[g] = mfma16(fh.v, wl.v, acc[g]);
            acc[g] = mfma16(fh.v, wh.v, acc[g]);
          }
        }
      }
      fetch(rs_hi, rs_lo, abase, t > 0, ahi, alo);
#pragma unroll
      for (int j = 0; j < KQ; ++j) {
        frag16 fh, fl;
        fh.u = ahi[j];
        fl.u = alo[j];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          frag16 wh, wl;
          wh.u = bu[0][g][j];
          wl.u = bu[1][g][j];
          acc[g] = mfma16(fl.v, wh.v, acc[g]);
          acc[g] = mfma16(fh.v, wl.v, acc[g]);
          acc[g] = mfma16(fh.v, wh.v, acc[g]);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) zt[par][wave][g][(lane >> 4) * 4 + r][lane & 15] = kactive ? acc[g][r] : 0.f;
      __syncthreads();
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
        z[g] = zin[g] + zt[par][0][g][er][eu] + zt[par][1][g][er][eu] + zt[par][2][g][er][eu] + zt[par][3][g][er][eu];
      par ^= 1;
      const float gi = fast_sigmoid(z[0]), gf = fast_sigmoid(z[1]), gg = fast_tanh(z[2]), go = fast_sigmoid(z[3]);
      const float c = gf * c_reg[i] + gi * gg;
      c_reg[i] = c;
      float h = go * fast_tanh(c);
      // (a non-finite or out-of-range h -- broken weights -- must not look like a sentinel to the consumers: they would
      //  spin until the time-out; the f32 output keeps the value as computed)
      const float hx = (h > -1.f && h < 1.f) ? h : (h >= 1.f ? 1.f : (h <= -1.f ? -1.f : 0.f));
      const bool row_ok = (r0 + er) < B;
      const unsigned hh = f2bf(hx);
      const unsigned hl = f2bf(hx - bf2f((bf16_t)hh));
      const unsigned hh_n = __shfl_xor(hh, 1), hl_n = __shfl_xor(hl, 1);
      const long orow = (long)t * B + r0 + er;
      if (row_ok && (eu & 1) == 0) {
        __hip_atomic_store(reinterpret_cast<unsigned*>(Xhi + (orow + B) * W + u0 + eu), hh | (hh_n << 16), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned*>(Xlo + (orow + B) * W + u0 + eu), hl | (hl_n << 16), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
      if (row_ok) {
        Hf[(orow + B) * W + u0 + eu] = h;
        if (t == T - 1) Cl[(orow + B) * W + u0 + eu] = c;
      }
    }
  }
}

// EIGHT units per workgroup (width 1024, few streams): with 16 units the (hi, lo) planes of U AND K are 512 registers per
// lane, which is why the kernel above runs width 1024 layer by layer (IN = false) and the rating window of the cfg5
// topology is a chain of depth x T steps.  Here a workgroup's 32 columns are 4 gates x 8 units -- MFMA column tile c
// holds gates 2c, 2c + 1 -- so U and K fit (256 registers), W/8 = 128 workgroups make a layer and TWO layers run as a
// wavefront per launch on the 256 CUs (a.l0 = the launch's first layer; its input planes, if any, are complete).
template <int KSTEPS, int MAXRB>
__global__ __launch_bounds__(256, 1) void lstm_scan_fwd_split_sent8_kernel(const KlScanFwdSplit a) {
  constexpr int KW = KSTEPS < 4 ? KSTEPS : 4;
  constexpr int KQ = KSTEPS / KW;
  static_assert(KSTEPS % KW == 0, "K split");
  constexpr int W = KSTEPS * 32;
  constexpr int NUG = W / 8;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int id = blockIdx.x;
  const int n_rg = a.n_rg, n_rb = a.n_rb, B = a.B, T = a.T;
  const int lrel = id / (NUG * n_rg);
  id -= lrel * NUG * n_rg;
  const int l = a.l0 + lrel;
  const int ug = id / n_rg, rg = id % n_rg;
  const int u0 = ug * 8;
  const bool has_in = l > 0;
  const bool kactive = wave < KW;
  const int kw = kactive ? wave : 0;

  __shared__ float zt[2][4][2][16][17];      // [buffer][wave][column tile][row][gate-of-pair * 8 + unit]

  const int kq = (lane >> 4) * 8;
  uint4 bu[2][2][KQ], bk[2][2][KQ];     // [hi/lo][column tile][k-step]
  {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const long wrow = ((long)(2 * g + ((lane & 15) >> 3)) * W + u0 + (lane & 7)) * W + (kw * KQ) * 32 + kq;
#pragma unroll
      for (int j = 0; j < KQ; ++j) {
        bu[0][g][j] = *reinterpret_cast<const uint4*>(a.UT_hi[l] + wrow + j * 32);
        bu[1][g][j] = *reinterpret_cast<const uint4*>(a.UT_lo[l] + wrow + j * 32);
        bk[0][g][j] = has_in ? *reinterpret_cast<const uint4*>(a.KT_hi[l] + wrow + j * 32) : uint4{0, 0, 0, 0};
        bk[1][g][j] = has_in ? *reinterpret_cast<const uint4*>(a.KT_lo[l] + wrow + j * 32) : uint4{0, 0, 0, 0};
      }
    }
  }
  const int er = (tid >> 3) & 15, eu = tid & 7;      // epilogue threads: tid < 128 = (row, unit)
  const bool ethread = tid < 128;
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  if (has_in) {
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[g] = a.bias[l][(long)g * W + u0 + eu];
  }
  float* Cl = a.C[l];
  float* Hf = a.Hf[l];
  bf16_t* Xhi = a.Xhi[l];
  bf16_t* Xlo = a.Xlo[l];
  const float* P1 = a.P1;
  unsigned* status = a.status;
  float c_reg[MAXRB];
#pragma unroll
  for (int i = 0; i < MAXRB; ++i) {
    const int rb = rg + i * n_rg;
    const int row = min(rb * 16 + er, B - 1);
    c_reg[i] = (rb < n_rb) ? Cl[(long)row * W + u0 + eu] : 0.f;
  }
  const long BW = (long)B * W;
  const __amdgpu_buffer_rsrc_t rs_hi = make_rsrc(Xhi, (long)(T + 1) * BW * 2);
  const __amdgpu_buffer_rsrc_t rs_lo = make_rsrc(Xlo, (long)(T + 1) * BW * 2);
  // input rows of step t = outputs of the layer below at block t + 1 (always a polled block)
  const __amdgpu_buffer_rsrc_t rs_ihi = make_rsrc(has_in ? a.Xhi[l - 1] + BW : Xhi, (long)T * BW * 2);
  const __amdgpu_buffer_rsrc_t rs_ilo = make_rsrc(has_in ? a.Xlo[l - 1] + BW : Xlo, (long)T * BW * 2);
  bool alive = true;                 // (wave-uniform: false once this wave or any other has given up)
  int par = 0;

  // the fragments of one operand, loaded until none of them is a sentinel (`poll` false: loaded once)
  auto fetch = [&](const __amdgpu_buffer_rsrc_t& rh, const __amdgpu_buffer_rsrc_t& rl, unsigned abase, bool poll,
                   uint4 (&fh)[KQ], uint4 (&fl)[KQ]) {
    for (unsigned spin = 0;; ++spin) {
      unsigned any = 0;
#pragma unroll
      for (int j = 0; j < KQ; ++j) {
        fh[j] = load16_sc1(rh, abase + j * 64);
        fl[j] = load16_sc1(rl, abase + j * 64);
      }
#pragma unroll
      for (int j = 0; j < KQ; ++j)
        any |= fh[j].x | fh[j].y | fh[j].z | fh[j].w | fl[j].x | fl[j].y | fl[j].z | fl[j].w;
      if (!poll || !alive || __all((any & 0x40004000u) == 0)) break;
      if ((spin & 63) == 63 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) alive = false;
      if (spin >= SPIN_LIMIT) {
        __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        alive = false;
      }
    }
  };

  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int i = 0; i < MAXRB; ++i) {
      const int rb = rg + i * n_rg;
      if (rb >= n_rb) continue;
      const int r0 = rb * 16;
      const int erow = min(r0 + er, B - 1);
      float zin[4];
      if (!has_in) {
        const float* p = P1 + ((long)t * B + erow) * 4 * W + u0 + eu;
#pragma unroll
        for (int g = 0; g < 4; ++g) zin[g] = p[(long)g * W];
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) zin[g] = bias4[g];
      }
      const int arow = min(r0 + (lane & 15), B - 1);
      const unsigned abase = (unsigned)((((long)t * B + arow) * W + (kw * KQ) * 32 + kq) * 2);
      f32x4 acc[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
      uint4 ahi[KQ], alo[KQ];
      if (has_in) {      // (the layer below runs ahead: normally there at the first look)
        fetch(rs_ihi, rs_ilo, abase, true, ahi, alo);
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
          frag16 fh, fl;
          fh.u = ahi[j];
          fl.u = alo[j];
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            frag16 wh, wl;
            wh.u = bk[0][g][j];
            wl.u = bk[1][g][j];
            acc[g] = mfma16(fl.v, wh.v, acc[g]);
            acc[g] = mfma16(fh.v, wl.v, acc[g]);
            acc[g] = mfma16(fh.v, wh.v, acc[g]);
          }
        }
      }
      fetch(rs_hi, rs_lo, abase, t > 0, ahi, alo);
#pragma unroll
      for (int j = 0; j < KQ; ++j) {
        frag16 fh, fl;
        fh.u = ahi[j];
        fl.u = alo[j];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          frag16 wh, wl;
          wh.u = bu[0][g][j];
          wl.u = bu[1][g][j];
          acc[g] = mfma16(fl.v, wh.v, acc[g]);
          acc[g] = mfma16(fh.v, wl.v, acc[g]);
          acc[g] = mfma16(fh.v, wh.v, acc[g]);
        }
      }
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) zt[par][wave][g][(lane >> 4) * 4 + r][lane & 15] = kactive ? acc[g][r] : 0.f;
      __syncthreads();
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ct = g >> 1, cc = (g & 1) * 8 + eu;      // gate g sits in column tile g / 2, columns (g % 2) * 8 + unit
        z[g] = zin[g] + zt[par][0][ct][er][cc] + zt[par][1][ct][er][cc] + zt[par][2][ct][er][cc] + zt[par][3][ct][er][cc];
      }
      par ^= 1;
      const float gi = fast_sigmoid(z[0]), gf = fast_sigmoid(z[1]), gg = fast_tanh(z[2]), go = fast_sigmoid(z[3]);
      const float c = gf * c_reg[i] + gi * gg;
      c_reg[i] = c;
      float h = go * fast_tanh(c);
      // (a non-finite or out-of-range h -- broken weights -- must not look like a sentinel to the consumers: they would
      //  spin until the time-out; the f32 output keeps the value as computed)
      const float hx = (h > -1.f && h < 1.f) ? h : (h >= 1.f ? 1.f : (h <= -1.f ? -1.f : 0.f));
      const bool row_ok = (r0 + er) < B;
      const unsigned hh = f2bf(hx);
      const unsigned hl = f2bf(hx - bf2f((bf16_t)hh));
      const unsigned hh_n = __shfl_xor(hh, 1), hl_n = __shfl_xor(hl, 1);
      const long orow = (long)t * B + r0 + er;
      if (ethread && row_ok && (eu & 1) == 0) {
        __hip_atomic_store(reinterpret_cast<unsigned*>(Xhi + (orow + B) * W + u0 + eu), hh | (hh_n << 16), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned*>(Xlo + (orow + B) * W + u0 + eu), hl | (hl_n << 16), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
      if (ethread && row_ok) {
        Hf[(orow + B) * W + u0 + eu] = h;
        if (t == T - 1) Cl[(orow + B) * W + u0 + eu] = c;
      }
    }
  }
}


// ---------------------------------------------------------------- backward scan
// block = 256 threads; output tile = dh of 16 rows x 16 units; wave w contracts
// over gate w's K range (W of the 4W columns) of dZ_l[t+1] . U_l^T and, below the
// top layer, dZ_{l+1}[t] . K_{l+1}^T; the four partial tiles meet in LDS and the
// gate derivatives run on the reduced tile.  dc lives in registers for all T steps.
// UP = false: single-layer launches (the from-above term arrives in dH from the big GEMM): no Kn registers.
template <int KSTEPS, int MAXRB, bool UP = true>
__global__ __launch_bounds__(256, 2) void lstm_scan_bwd_kernel(const KlScanBwd a) {
  constexpr int W = KSTEPS * 32;
  constexpr int NUG = W / 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int id = blockIdx.x;
  const int n_rg = a.n_rg, n_rb = a.n_rb, B = a.B, T = a.T;
  const int l = id / (NUG * n_rg);
  id -= l * NUG * n_rg;
  const int ug = id / n_rg, rg = id % n_rg;
  const int u0 = ug * 16;
  const bool has_up = UP && l < a.L - 1;

  __shared__ float zt[4][16][17];
  __shared__ int ok_flag;

  const long wrow = (long)(u0 + (lane & 15)) * 4 * W + (long)wave * W;
  const int kq = (lane >> 4) * 8;
  uint4 bu[KSTEPS], bk[UP ? KSTEPS : 1];
#pragma unroll
  for (int j = 0; j < KSTEPS; ++j) {
    bu[j] = *reinterpret_cast<const uint4*>(a.Un[l] + wrow + j * 32 + kq);
    if (UP) bk[j] = has_up ? *reinterpret_cast<const uint4*>(a.Kn[l + 1] + wrow + j * 32 + kq) : uint4{0, 0, 0, 0};
  }
  const int er = tid >> 4, eu = tid & 15;
  float dc_reg[MAXRB];
#pragma unroll
  for (int i = 0; i < MAXRB; ++i) dc_reg[i] = 0.f;
  const long BW = (long)B * W;
  const bf16_t* Gl = a.G[l];
  const float* Cl = a.C[l];
  bf16_t* dZl = a.dZ[l];
  const float* dH = a.dH;
  const float* maskl = a.mask[l];
  unsigned* status = a.status;
  const __amdgpu_buffer_rsrc_t rs_own = make_rsrc(dZl, (long)T * BW * 4 * 2);
  const __amdgpu_buffer_rsrc_t rs_up = make_rsrc(has_up ? a.dZ[l + 1] : dZl, (long)T * BW * 4 * 2);
  unsigned* cnt_own = a.counters + (long)l * n_rb * T;
  unsigned* cnt_up = a.counters + (long)(has_up ? l + 1 : l) * n_rb * T;
  bool alive = true;
  SSTAMP_INIT(0);

  for (int t = T - 1; t >= 0; --t) {
#pragma unroll
    for (int i = 0; i < MAXRB; ++i) {
      const int rb = rg + i * n_rg;
      if (rb >= n_rb) continue;
      const int r0 = rb * 16;
      const int erow = min(r0 + er, B - 1);
      SSTAMP(16);
      // epilogue operands (written by earlier launches: plain loads), issued before the wait
      const bf16_t* gp = Gl + ((long)t * B + erow) * 4 * W + u0 + eu;
      const bf16_t g0 = gp[0], g1 = gp[W], g2 = gp[2 * W], g3 = gp[3 * W];
      const float c = Cl[((long)(t + 1) * B + erow) * W + u0 + eu];
      const float cp = Cl[((long)t * B + erow) * W + u0 + eu];
      float dh = 0.f;
      if (!has_up) dh = dH[((long)t * B + erow) * W + u0 + eu];
      float mk = 1.f;
      if (maskl) mk = maskl[(long)erow * W + u0 + eu];
      if (!has_up) dh *= mk;
      float omask[4] = {1.f, 1.f, 1.f, 1.f};
      if (has_up && maskl) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int mr = min(r0 + (lane >> 4) * 4 + r, B - 1);
          omask[r] = maskl[(long)mr * W + u0 + (lane & 15)];
        }
      }
      if (tid == 0) {
        bool ok = alive;
        if (ok && has_up) ok = poll_counter(cnt_up + (long)rb * T + t, NUG, status);
        if (ok && t < T - 1) ok = poll_counter(cnt_own + (long)rb * T + (t + 1), NUG, status);
        ok_flag = ok ? 1 : 0;
      }
      SSTAMP(17);
      __syncthreads();
      SSTAMP(18);
      alive = ok_flag != 0;
      const int arow = min(r0 + (lane & 15), B - 1);
      // (two partial sums per contraction: with one wave per SIMD -- few streams -- a single accumulator makes every MFMA
      //  wait out the one before it, ~0.5 us per contraction and step at width 512)
      f32x4 acc_up = f32x4{0.f, 0.f, 0.f, 0.f}, acc = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 acc_up2 = f32x4{0.f, 0.f, 0.f, 0.f}, acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (UP && alive && has_up) {
        uint4 av[UP ? KSTEPS : 1];
        const unsigned base = (unsigned)((((long)t * B + arow) * 4 * W + (long)wave * W + kq) * 2);
#pragma unroll
        for (int j = 0; j < (UP ? KSTEPS : 1); ++j) av[j] = load16_sc1(rs_up, base + j * 64);
#pragma unroll
        for (int j = 0; j < (UP ? KSTEPS : 1); ++j) {
          frag16 fa, fb;
          fa.u = av[j];
          fb.u = bk[j];
          if (j & 1) acc_up2 = mfma16(fa.v, fb.v, acc_up2);
          else acc_up = mfma16(fa.v, fb.v, acc_up);
        }
      }
      if (alive && t < T - 1) {
        // (at most 16 fragments in flight: width 1024 would otherwise need 128 registers for them)
        constexpr int CH = KSTEPS < 16 ? KSTEPS : 16;
        const unsigned base = (unsigned)((((long)(t + 1) * B + arow) * 4 * W + (long)wave * W + kq) * 2);
#pragma unroll
        for (int c0 = 0; c0 < KSTEPS; c0 += CH) {
          uint4 av[CH];
#pragma unroll
          for (int j = 0; j < CH; ++j) av[j] = load16_sc1(rs_own, base + (c0 + j) * 64);
#pragma unroll
          for (int j = 0; j < CH; ++j) {
            frag16 fa, fb;
            fa.u = av[j];
            fb.u = bu[c0 + j];
            if (j & 1) acc2 = mfma16(fa.v, fb.v, acc2);
            else acc = mfma16(fa.v, fb.v, acc);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) zt[wave][(lane >> 4) * 4 + r][lane & 15] = (acc[r] + acc2[r]) + (acc_up[r] + acc_up2[r]) * omask[r];
      SSTAMP(19);
      __syncthreads();
      SSTAMP(20);
      dh += zt[0][er][eu] + zt[1][er][eu] + zt[2][er][eu] + zt[3][er][eu];
      const float gi = bf2f(g0), gf = bf2f(g1), gg = bf2f(g2), go = bf2f(g3);
      const float tc = fast_tanh(c);
      const float dc = dh * go * (1.f - tc * tc) + dc_reg[i];
      dc_reg[i] = dc * gf;
      const float d_o = dh * tc, d_i = dc * gg, d_g = dc * gi, d_f = dc * cp;
      const unsigned z0 = f2bf(d_i * gi * (1.f - gi)), z1 = f2bf(d_f * gf * (1.f - gf));
      const unsigned z2 = f2bf(d_g * (1.f - gg * gg)), z3 = f2bf(d_o * go * (1.f - go));
      const unsigned n0 = __shfl_xor(z0, 1), n1 = __shfl_xor(z1, 1), n2 = __shfl_xor(z2, 1), n3 = __shfl_xor(z3, 1);
      SSTAMP(21);
      if ((r0 + er) < B && alive && (eu & 1) == 0) {
        unsigned* zp = reinterpret_cast<unsigned*>(dZl + ((long)t * B + r0 + er) * 4 * W + u0 + eu);
        __hip_atomic_store(zp, z0 | (n0 << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(zp + W / 2, z1 | (n1 << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(zp + W, z2 | (n2 << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(zp + 3 * W / 2, z3 | (n3 << 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      SSTAMP(22);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      SSTAMP(23);
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(cnt_own + (long)rb * T + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      SSTAMP(24);
    }
  }
  SSTAMP_FLUSH();
}

// dynamic LDS of the wide kernels up to (excluding) the per-row-block state slots [MAXRB][1024] f32
#define KL_BWD_WIDE_LDS(KS) (4 * (KS) * 1024 + 16 * 16 * 17 * 4 + 2 * 4 * 64 * 16 * 2 + 16)
#define KL_FWD_WIDE_LDS0(KS) ((KS) * 1024 + 16 * 4 * 16 * 17 * 4 + 2 * 64 * 16 * 2 + 16 * 64 * 2 + 16)
#define KL_FWD_WIDE_LDS(KS) (KL_FWD_WIDE_LDS0(KS) + 16 * 64 * 4 + 4 * 16 * 64 * 2 + 16 * 64 * 2 + 4 * 64 * 4)

// ---------------------------------------------------------------- backward scan, one layer, wide workgroups
// For many row blocks the thin kernel above is bound by fabric traffic: W/16 workgroups
// per row block each pull the same 16 x 4W dZ tile with write-through reads (stamps:
// 46 % of a step).  Here a workgroup is 16 waves = 4 K-quarters x 4 unit groups = 64
// hidden units, the tile is fetched ONCE per workgroup (each wave 1/16 of it) into LDS
// and shared, so only W/64 workgroups read it.  One layer per launch (recurrent
// contraction only; the from-above term arrives in dH from the big GEMM).  On request (a.dZT: the
// TN weight-gradient GEMMs, KL_GEMM_AN=0) the step's dZ tile is also written transposed ([4W][T*B]);
// by default the weight-gradient GEMMs read the row-major dZ K-major and nothing transposed is
// written: these scattered 16-byte stores cost 15-21 % of a step.
// SENT: data-sentinel hand-off instead of counters (a.sentinel 1: dZ pre-filled with 0xFFFF halfwords
// by the caller; 2: only the first two steps pre-filled, the publishing lanes re-arm step t-2): no
// poll, no drain, no atomics; with several row blocks per workgroup the next block's tile is
// prefetched by LDS-DMA behind the epilogue (see lstm_scan_fwd_wide_kernel).
template <int KSTEPS, int MAXRB, bool SENT>
__global__ __launch_bounds__(1024, 1) void lstm_scan_bwd_wide_kernel(const KlScanBwd a) {
  constexpr int W = KSTEPS * 32;
  constexpr int NWG_RB = W / 64;          // producers per (row block, step)
  constexpr int JW = KSTEPS / 4;          // k-steps of a quarter that one wave fetches
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq4 = wave & 3, ug = wave >> 2;
  const int n_rg = a.n_rg, n_rb = a.n_rb, B = a.B, T = a.T;
  // placement: XCD x = blockIdx % 8 hosts the row groups x R .. x R + R - 1 (R = ceil(n_rg / 8)), each
  // with all its column groups: partners share an L2, and so do the row blocks whose 32-byte pieces
  // make up one line of the transposed copies (4 adjacent blocks = 64 rows = 128 bytes)
  const int xcd = blockIdx.x & 7, yy = blockIdx.x >> 3;
  const int cg = yy % NWG_RB, rq = yy / NWG_RB, rg = xcd * ((n_rg + 7) >> 3) + rq;
  if (rg >= n_rg) return;
  const int u0 = cg * 64;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* a_tile = smem;                                           // [4 quarters][KSTEPS][1 KiB]
  float (*zt)[16][17] = reinterpret_cast<float (*)[16][17]>(smem + 4 * KSTEPS * 1024);   // [16 waves][16][17]
  bf16_t* tr = reinterpret_cast<bf16_t*>(smem + 4 * KSTEPS * 1024 + 16 * 16 * 17 * 4);   // [4 gates][64 units][16 rows]
  // (all LDS in the dynamic region: a static object in front would shift its 16-byte alignment)
  bf16_t* pub = tr + 4 * 64 * 16;                                         // [4 gates][16 rows][64 units]
  int& ok_flag = *reinterpret_cast<int*>(smem + 4 * KSTEPS * 1024 + 16 * 16 * 17 * 4 + 2 * 4 * 64 * 16 * 2);

  const int kq = (lane >> 4) * 8;
  uint4 bu[KSTEPS];
  {
    const long wrow = (long)(u0 + ug * 16 + (lane & 15)) * 4 * W + (long)kq4 * W;
#pragma unroll
    for (int j = 0; j < KSTEPS; ++j) bu[j] = *reinterpret_cast<const uint4*>(a.Un[0] + wrow + j * 32 + kq);
  }
  const int er = wave, eu = lane;                  // epilogue thread = (row = wave: scalar, unit of 64)
  // running dc of this thread's (row, unit) per row block: a register for one block, else LDS
  // slots (the row-block loop stays rolled: unrolled it spilled 7-47 VGPRs at the 128 cap)
  float dc_one = 0.f;
  float* dc_slot = reinterpret_cast<float*>(smem + KL_BWD_WIDE_LDS(KSTEPS)) + tid;
  if (MAXRB > 1) {
    for (int i = 0; i < MAXRB; ++i) dc_slot[i * 1024] = 0.f;
  }
  float dbacc[4] = {0.f, 0.f, 0.f, 0.f};   // bias gradient: this thread's (unit, gate) summed over its rows and steps
  const long BW = (long)B * W;
  const bf16_t* Gl = a.G[0];
  const float* Cl = a.C[0];
  bf16_t* dZl = a.dZ[0];
  const bool has_dzt = a.dZT != nullptr;
  const __amdgpu_buffer_rsrc_t rs_dzt = make_rsrc(a.dZT, has_dzt ? (long)4 * W * a.ldt * 2 : 0);
  const float* dH = a.dH;
  const float* maskl = a.mask[0];
  unsigned* status = a.status;
  const __amdgpu_buffer_rsrc_t rs_own = make_rsrc(dZl, (long)T * BW * 4 * 2);
  unsigned* cnt_own = a.counters;
  bool alive = true;
  int pend = -1;       // publishing waves: counter index of stores issued but not yet signalled
  // (deferring needs a second row block whose publish releases the first one's signal: a workgroup
  // with a single block would wait for its own deferred signal)
  const bool defer = rg + n_rg < n_rb;
  constexpr bool PREF = SENT && MAXRB > 1;
  const bool pref_ok = (B & 15) == 0;          // whole tiles only: the counted wait assumes every store is issued
  const int n_raw = maskl ? 8 : 7;             // epilogue inputs loaded at the top of a block, behind the DMA
  const unsigned lds_a = (unsigned)(size_t)(lds_void_t*)a_tile;
  // vector memory operations of a wave between its DMA and the next block's wait: the publish (waves 0-7)
  // Rolling sentinels (a.sentinel == 2): instead of a pre-fill of all of dZ by the caller (1 GiB per layer at
  // B = 1024), the publishing lanes write the sentinel over their piece of step t - 2 together with the data
  // of step t (the caller pre-fills only steps T-1 and T-2).  Safe: a consumer touches step t - 2 only after it
  // has seen this workgroup's data of step t - 1, which was stored after the wait for this block's epilogue
  // inputs (vmcnt(0): the sentinel store of step t had completed by then).
  const bool roll = SENT && a.sentinel == 2;
  const __amdgpu_buffer_rsrc_t rs_null = make_rsrc(dZl, 0);
  const int pf_after = wave < 8 ? (roll ? 2 : 1) : 0;
  int pf_issued = 0;
  if (tid == 0) ok_flag = 1;
  __syncthreads();
  bool local = false;
  if (SENT && a.xcc_slots)
    local = xcd_local_group(a.xcc_slots, a.gen, NWG_RB, [&](int j) { return xcd + 8 * (rq * NWG_RB + j); }, &ok_flag + 1, status);
  SSTAMP_INIT(0);

  // the transposed copy (waves 8..15): behind the publish, or -- several row blocks per workgroup -- one
  // phase late, at the start of the next block's MFMA phase (see LATE in lstm_scan_fwd_wide_kernel)
  constexpr bool LATE = SENT && MAXRB > 1;
  int t_prev = 0, r0_prev = 0;
  bool have_prev = false;
  auto store_transposed = [&](int t, int r0) {
    int stid = tid;
    if (MAXRB > 1 || SENT) asm volatile("" : "+v"(stid));
    // 256 columns (gate, unit) x 16 rows: two 16-byte stores per column (the waves that do not
    // publish); buffer store = lane offset + scalar step offset (B % 8 == 0 is a launch condition)
    const int col = (stid - 512) >> 1, half = stid & 1;
    const int g = col >> 6, u = col & 63;
    if (r0 + half * 8 < B)
      store16(rs_dzt, (unsigned)(((long)(g * W + u0 + u) * a.ldt + half * 8) * 2), (unsigned)(t * B + r0) * 2u,
              *reinterpret_cast<const uint4*>(tr + col * 16 + half * 8));
  };
  for (int t = T - 1; t >= 0; --t) {
#pragma unroll 1
    for (int i = 0; i < MAXRB; ++i) {
      const int rb = rg + i * n_rg;
      if (rb >= n_rb) continue;
      const int r0 = rb * 16;
      const int erow = min(r0 + er, B - 1);          // wave-uniform: scalar row bases, one lane offset per load
      // (raw loads only; they are consumed in the epilogue behind a compiler fence, or their waits --
      // in-order behind the previous step's write-through stores -- would be scheduled right here)
      const bf16_t* gp = Gl + ((long)t * B + erow) * 4 * W + u0;
      unsigned g0 = gp[eu], g1 = gp[W + eu], g2 = gp[2 * W + eu], g3 = gp[3 * W + eu];
      float c = (Cl + ((long)(t + 1) * B + erow) * W + u0)[eu];
      float cp = (Cl + ((long)t * B + erow) * W + u0)[eu];
      float dh = (dH + ((long)t * B + erow) * W + u0)[eu];
      float mkv = maskl ? (maskl + (long)erow * W + u0)[eu] : 1.f;
      SSTAMP(16);
      if (!SENT) {
        if (tid == 0) {
          bool ok = alive;
          if (ok && t < T - 1) ok = poll_counter(cnt_own + (long)rb * T + (t + 1), 8 * NWG_RB, status);
          ok_flag = ok ? 1 : 0;
        }
        SSTAMP(17);
        __syncthreads();
        SSTAMP(18);
        alive = ok_flag != 0;
      }
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
      if (t < T - 1) {
        // fetch 1/16 of the 16 x 4W tile of dZ[t+1]: quarter kq4, k-steps ug*JW .. +JW
        const int arow = min(r0 + (lane & 15), B - 1);
        const unsigned base = (unsigned)((((long)(t + 1) * B + arow) * 4 * W + (long)kq4 * W + kq) * 2);
        unsigned char* frag = a_tile + ((kq4 * KSTEPS) + ug * JW) * 1024 + lane * 16;
        if (!SENT) {
          uint4 av[JW];
#pragma unroll
          for (int j = 0; j < JW; ++j) av[j] = alive ? load16_sc1(rs_own, base + (ug * JW + j) * 64) : uint4{0, 0, 0, 0};
#pragma unroll
          for (int j = 0; j < JW; ++j) *reinterpret_cast<uint4*>(frag + j * 1024) = av[j];
        } else {
          // sentinel hand-off: every wave brings its own granules into LDS (DMA: no registers) and
          // re-fetches them until none carries the sentinel; with the prefetch the first round is
          // already in flight
          bool ok = false;
          if (alive) {
            // Without a prefetch in flight the wave first probes ONE of its fragments (1/4 of the
            // traffic per round: 256 workgroups spinning on whole 64 KiB tiles slow the publishes down)
            // and fetches the rest once that one is there.
            bool issued = PREF && pf_issued;
            int lo = issued ? 0 : JW - 1;          // fragments [lo, JW) are fetched and checked this round
            for (unsigned spin = 0; spin < SPIN_LIMIT; ++spin) {
              if (issued) {
                wait_vm(pf_after + n_raw);
              } else {
#pragma unroll
                for (int j = 0; j < JW; ++j)
                  if (j >= lo) {
                    if (local) glds16_nt(rs_own, base + (ug * JW + j) * 64, lds_a + ((kq4 * KSTEPS) + ug * JW + j) * 1024);
                    else glds16_sc1(rs_own, base + (ug * JW + j) * 64, lds_a + ((kq4 * KSTEPS) + ug * JW + j) * 1024);
                  }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              }
              bool all_ok = true;
#pragma unroll
              for (int j = 0; j < JW; ++j)
                if (j >= lo) all_ok = all_ok && granule_valid(*reinterpret_cast<const uint4*>(frag + j * 1024));
              if (__all(all_ok)) {
                if (lo == 0) { ok = true; break; }
                lo = 0;                            // probe passed: now everything (the probe fragment again: cheap, and keeps the loop simple)
                issued = false;
                continue;
              }
#ifdef KL_STAMP
              if (issued && blockIdx.x == STAMP_WG && threadIdx.x == 0) stamp_lds[28] += 1;   // a prefetch that came too early
#endif
              issued = false;
              if (lo == 0 && JW > 1) lo = JW - 1;  // back to probing
              if ((spin & 63) == 63 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
              __builtin_amdgcn_s_sleep(2);
            }
            if (!ok) {
              __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ok_flag = 0;
            }
          }
          if (!ok) {
#pragma unroll
            for (int j = 0; j < JW; ++j) *reinterpret_cast<uint4*>(frag + j * 1024) = uint4{0, 0, 0, 0};
          }
        }
        SSTAMP(19);
        __syncthreads();
        SSTAMP(20);
        if (SENT) alive = ok_flag != 0;
        if (LATE && have_prev && wave >= 8) store_transposed(t_prev, r0_prev);
#pragma unroll
        for (int j = 0; j < KSTEPS; ++j) {
          frag16 fa, fb;
          fa.u = *reinterpret_cast<const uint4*>(a_tile + (kq4 * KSTEPS + j) * 1024 + lane * 16);
          fb.u = bu[j];
          acc = mfma16(fa.v, fb.v, acc);
        }
      }
      else if (LATE && have_prev && wave >= 8) store_transposed(t_prev, r0_prev);     // (first step: nothing to fetch)
#pragma unroll
      for (int r = 0; r < 4; ++r) zt[wave][(lane >> 4) * 4 + r][lane & 15] = acc[r];
      SSTAMP(21);
      __syncthreads();
      SSTAMP(22);
      const int wz = (eu >> 4) * 4;      // the four K-quarter waves of this unit group
      asm volatile("" : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(c), "+v"(cp), "+v"(dh), "+v"(mkv));
      SSTAMP(27);
      if (PREF) {
        // the tile buffer is free again (the MFMAs above consumed it): fetch the next block's tile
        pf_issued = 0;
        int ni = i + 1, nt = t;
        if (ni >= MAXRB || rg + ni * n_rg >= n_rb) { ni = 0; nt = t - 1; }
        if (pref_ok && alive && nt >= 0 && nt < T - 1) {
          const int nr0 = (rg + ni * n_rg) * 16;
          const unsigned nbase = (unsigned)((((long)(nt + 1) * B + nr0 + (lane & 15)) * 4 * W + (long)kq4 * W + kq) * 2);
#pragma unroll
          for (int j = 0; j < JW; ++j)
            if (local) glds16_nt(rs_own, nbase + (ug * JW + j) * 64, lds_a + ((kq4 * KSTEPS) + ug * JW + j) * 1024);
            else glds16_sc1(rs_own, nbase + (ug * JW + j) * 64, lds_a + ((kq4 * KSTEPS) + ug * JW + j) * 1024);
          pf_issued = 1;
        }
      }
      dh = dh * mkv + (zt[wz][er][eu & 15] + zt[wz + 1][er][eu & 15] + zt[wz + 2][er][eu & 15] + zt[wz + 3][er][eu & 15]);
      const float gi = bf2f((bf16_t)g0), gf = bf2f((bf16_t)g1), gg = bf2f((bf16_t)g2), go = bf2f((bf16_t)g3);
      const float tc = fast_tanh(c);
      const float dc = dh * go * (1.f - tc * tc) + (MAXRB > 1 ? dc_slot[i * 1024] : dc_one);
      if (MAXRB > 1) dc_slot[i * 1024] = dc * gf;
      else dc_one = dc * gf;
      const float d_o = dh * tc, d_i = dc * gg, d_g = dc * gi, d_f = dc * cp;
      const unsigned z0 = f2bf(d_i * gi * (1.f - gi)), z1 = f2bf(d_f * gf * (1.f - gf));
      const unsigned z2 = f2bf(d_g * (1.f - gg * gg)), z3 = f2bf(d_o * go * (1.f - go));
      const bool row_ok = (r0 + er) < B;
      if (row_ok && alive) {
        dbacc[0] += bf2f((bf16_t)z0); dbacc[1] += bf2f((bf16_t)z1); dbacc[2] += bf2f((bf16_t)z2); dbacc[3] += bf2f((bf16_t)z3);
      }
      pub[(0 * 16 + er) * 64 + eu] = (bf16_t)z0;
      pub[(1 * 16 + er) * 64 + eu] = (bf16_t)z1;
      pub[(2 * 16 + er) * 64 + eu] = (bf16_t)z2;
      pub[(3 * 16 + er) * 64 + eu] = (bf16_t)z3;
      if (has_dzt) {   // stage the tile transposed for the off-chain copy below
        tr[(0 * 64 + eu) * 16 + er] = row_ok ? (bf16_t)z0 : (bf16_t)0;
        tr[(1 * 64 + eu) * 16 + er] = row_ok ? (bf16_t)z1 : (bf16_t)0;
        tr[(2 * 64 + eu) * 16 + er] = row_ok ? (bf16_t)z2 : (bf16_t)0;
        tr[(3 * 64 + eu) * 16 + er] = row_ok ? (bf16_t)z3 : (bf16_t)0;
      }
      SSTAMP(23);
      __syncthreads();
      SSTAMP(24);
      // publish dZ[t]: eight waves, one 16-byte write-through store per lane; each storing
      // wave drains its own stores and then counts itself in (8 arrivals per workgroup)
      // (lane offsets re-derived every step from an opaque copy of the thread id: as hoisted loop
      // invariants they cost registers the 128 cap does not have)
      int stid = tid;
      if (MAXRB > 1 || SENT) asm volatile("" : "+v"(stid));
      if (wave < 8) {
        // (several row blocks per workgroup: the previous block's drain + signal happen here, its
        // write-through latency hidden behind this block's step)
        if (!SENT && MAXRB > 1 && pend >= 0) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          signal_wave(cnt_own + pend, lane);
        }
        const int g = stid >> 7, prow = (stid >> 3) & 15, seg = stid & 7;
        if (alive && r0 + prow < B) {
          const uint4 v = *reinterpret_cast<const uint4*>(pub + (g * 16 + prow) * 64 + seg * 8);
          const unsigned off = (unsigned)((((long)t * B + r0 + prow) * 4 * W + (long)g * W + u0 + seg * 8) * 2);
          if (SENT && local) store16(rs_own, off, 0u, v);      // stays in this XCD's L2, where all its readers are
          else store16_sc1(rs_own, off, v);
          if (roll) {    // (steps T-1, T-2 have no t + 2: a null buffer drops the store, the count stays)
            const unsigned soff = off - (unsigned)((long)2 * B * 4 * W * 2);
            const uint4 ones = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if (local) store16(t >= 2 ? rs_own : rs_null, soff, 0u, ones);
            else store16_sc1(t >= 2 ? rs_own : rs_null, soff, ones);
          }
        }
        if (SENT) {
          // the data is its own signal: nothing to drain, nothing to count
        } else if (MAXRB > 1 && defer) {
          pend = rb * T + t;           // drained and signalled at the next publish
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          SSTAMP(25);
          signal_wave(cnt_own + (long)rb * T + t, lane);
        }
      }
      SSTAMP(26);
      if (!LATE && has_dzt && alive && wave >= 8) store_transposed(t, r0);
      if (LATE) {
        t_prev = t;
        r0_prev = r0;
        have_prev = has_dzt && alive;
      }
    }
  }
  if (!SENT && MAXRB > 1 && tid < 512 && pend >= 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    signal_wave(cnt_own + pend, lane);
  }
  if (LATE && have_prev && wave >= 8) store_transposed(t_prev, r0_prev);     // the last block's
  SSTAMP_FLUSH();
  // db[g*W + u] += sum over this workgroup's rows and all steps (16 partials per column meet in LDS)
  if (a.db) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(a_tile);     // [4 gates][16 rows][64 units]
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

// ---------------------------------------------------------------- forward scan, one layer, wide workgroups
// Counterpart of lstm_scan_bwd_wide_kernel for B >= 512 streams: 16 waves = 4 K-quarters x
// 4 unit groups = 64 hidden units per workgroup, the 16 x W state tile is fetched once
// per workgroup (one fragment per wave) and shared through LDS.  Only the recurrent
// contraction is carried; the input side arrives either as the precomputed P rows of the
// big GEMM (layers >= 1) or -- layer 0 -- straight from the look-up tables
// EK[idx] + sum_n CtxK_n[ctx_n] + b (no P1 buffer at all).  On request (a.HT / a.HdT: the TN
// weight-gradient GEMMs) the outputs are also written transposed ([W][(T+1)B]); by default they are not
// (the weight-gradient GEMMs read the row-major outputs K-major).
template <int KSTEPS, int MAXRB, bool SENT>
__global__ __launch_bounds__(1024, 1) void lstm_scan_fwd_wide_kernel(const KlScanFwdWide a) {
  // Register budget: 1024 threads leave 128 VGPRs per lane and the resident weights take 64, so
  // everything else is kept cheap: the hand-off flavour is a template parameter, row-wise values
  // are wave-uniform (an epilogue thread's row IS its wave: scalar registers), and every per-thread
  // global access is a buffer operation = a 32-bit lane offset + a scalar offset for the step.
  constexpr int W = KSTEPS * 32;
  constexpr int KQ = KSTEPS / 4;
  constexpr int NWG_RB = W / 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq4 = wave & 3, ug = wave >> 2;
  const int n_rg = a.n_rg, n_rb = a.n_rb, B = a.B, T = a.T;
  // placement: XCD x = blockIdx % 8 hosts the row groups x R .. x R + R - 1 (R = ceil(n_rg / 8)), each
  // with all its column groups: partners share an L2, and so do the row blocks whose 32-byte pieces
  // make up one line of the transposed copies (4 adjacent blocks = 64 rows = 128 bytes)
  const int xcd = blockIdx.x & 7, yy = blockIdx.x >> 3;
  const int cg = yy % NWG_RB, rq = yy / NWG_RB, rg = xcd * ((n_rg + 7) >> 3) + rq;
  if (rg >= n_rg) return;
  const int u0 = cg * 64;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* a_tile = smem;                                                       // [KSTEPS][1 KiB]
  float (*zt)[4][16][17] = reinterpret_cast<float (*)[4][16][17]>(smem + KSTEPS * 1024);   // [16 waves][4 gates][16][17]
  bf16_t* tr = reinterpret_cast<bf16_t*>(smem + KSTEPS * 1024 + 16 * 4 * 16 * 17 * 4);   // [2][64 units][16 rows]
  bf16_t* pub = tr + 2 * 64 * 16;                                                       // [16 rows][64 units]
  int& ok_flag = *reinterpret_cast<int*>(smem + KSTEPS * 1024 + 16 * 4 * 16 * 17 * 4 + 2 * 64 * 16 * 2 + 16 * 64 * 2);
  float* st_c = reinterpret_cast<float*>(smem + KL_FWD_WIDE_LDS0(KSTEPS));              // [16 rows][64 units] staged c
  bf16_t* st_g = reinterpret_cast<bf16_t*>(st_c + 16 * 64);                             // [4 gates][16 rows][64 units]
  bf16_t* st_hd = st_g + 4 * 16 * 64;                                                   // [16 rows][64 units]

  const int kq = (lane >> 4) * 8;
  uint4 bu[4][KQ];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const long wrow = ((long)g * W + u0 + ug * 16 + (lane & 15)) * W + (kq4 * KQ) * 32 + kq;
#pragma unroll
    for (int j = 0; j < KQ; ++j) bu[g][j] = *reinterpret_cast<const uint4*>(a.UT + wrow + j * 32);
  }
  const int er = wave, eu = lane;           // epilogue thread = (row = wave, unit = lane)
  const float* maskl = a.mask;
  const float* P = a.P;
  unsigned* status = a.status;
  // layer-0 bias of this workgroup's 4 x 64 gate columns: LDS, not registers (zero in P mode)
  float* bias_l = reinterpret_cast<float*>(st_hd + 16 * 64);                            // [4 gates][64 units]
  if (tid < 256) bias_l[tid] = P ? 0.f : a.bias[(long)(tid >> 6) * W + u0 + (tid & 63)];
  // cell state of this thread's (row, unit) per row block: a register for one block, else LDS slots
  float c_one = 0.f;
  float* c_slot = reinterpret_cast<float*>(smem + KL_FWD_WIDE_LDS(KSTEPS)) + tid;
  for (int i = 0; i < MAXRB; ++i) {
    const int rb = rg + i * n_rg;
    const int row = min(rb * 16 + er, B - 1);
    const float c0 = (rb < n_rb) ? a.C[(long)row * W + u0 + eu] : 0.f;
    if (MAXRB > 1) c_slot[i * 1024] = c0;
    else c_one = c0;
  }
  const long BW = (long)B * W;
  const __amdgpu_buffer_rsrc_t rs_h = make_rsrc(a.H, (long)(T + 1) * BW * 2);
  const __amdgpu_buffer_rsrc_t rs_c = make_rsrc(a.C, (long)(T + 1) * BW * 4);
  const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(a.G, (long)T * BW * 4 * 2);
  const __amdgpu_buffer_rsrc_t rs_hd = make_rsrc(a.Hd, a.Hd ? (long)T * BW * 2 : 0);          // zero records: stores dropped
  const __amdgpu_buffer_rsrc_t rs_ht = make_rsrc(a.HT, a.HT ? (long)W * a.ldt * 2 : 0);
  const __amdgpu_buffer_rsrc_t rs_hdt = make_rsrc(a.HdT, a.HdT ? (long)W * a.ldt_d * 2 : 0);
  unsigned* cnt_own = a.counters;
  bool alive = true;
  if (tid == 0) ok_flag = 1;
  __syncthreads();
  bool local = false;
  if (SENT && a.xcc_slots)
    local = xcd_local_group(a.xcc_slots, a.gen, NWG_RB, [&](int j) { return xcd + 8 * (rq * NWG_RB + j); }, &ok_flag + 1, status);
  // Tile prefetch (sentinel hand-off, several row blocks per workgroup): the next block's tile was
  // published a whole block ago, so its fetch is issued as an LDS-DMA right after this block's MFMA
  // phase has released the tile buffer and lands behind the epilogue; the next block only reads its own
  // granules back and checks them (falling back to the spin when a producer was late).
  constexpr bool PREF = SENT && MAXRB > 1;
  const bool pref_ok = (B & 15) == 0;          // whole tiles only: the counted wait below assumes every store is issued
  const unsigned lds_a = (unsigned)(size_t)(lds_void_t*)a_tile;
  // vector memory operations a wave issues between the DMA and the next block's wait: the publish
  const int pf_after = wave < 2 ? 1 : 0;       // (the other stores leave late, in front of the DMA: see LATE)
  int pf_issued = 0;
  int pend = -1;       // counter hand-off, publishing waves: counter index of stores issued but not yet signalled
  // (deferring needs a second row block whose publish releases the first one's signal: a workgroup
  // with a single block would wait for its own deferred signal)
  const bool defer = rg + n_rg < n_rb;
  SSTAMP_INIT(0);

  // layer 0: the table row ids of a step are fetched (scalar loads, the row is wave-uniform) at the
  // end of the step before, where their latency hides behind the hand-off
  int id_cur = 0, c0_cur = 0;
  if (!P) {
    const long src0 = (long)min(rg * 16 + er, B - 1) * T;
    id_cur = sload_i32(a.idx + src0);
    if (a.n_ctx > 0) c0_cur = sload_i32(a.ctx + src0 * a.n_ctx);
  }
  // What only later launches read (waves 2..15).  With one row block per workgroup it follows the publish;
  // with several (LATE) a block's staging buffers are stored one phase late, at the start of the next
  // block's MFMA phase when that block's tile has arrived, instead of sitting in the memory pipeline in
  // front of the next tile's loads (same box, B = 1024 / 2048: 2-3 % of a training step; at B = 512 the
  // late stores cost more under the epilogue's wait for its inputs than they save).
  constexpr bool LATE = SENT && MAXRB > 1;
  unsigned trow_prev = 0;
  int r0_prev = 0;
  bool have_prev = false;
  auto offchain = [&](unsigned trow, int r0) {
    int stid = tid;
    if (MAXRB > 1) asm volatile("" : "+v"(stid));
        // off the hand-off chain (buffer stores: lane offset + scalar step offset; a null buffer has
      // zero records and drops the store)
      const int q = stid - 128;
      if (q < 512) {                       // gate activations: [gate][row] x 8 pieces of 8 units
        const int g = q >> 7, prow = (q >> 3) & 15, seg = q & 7;
        if (r0 + prow < B)
          store16(rs_g, (unsigned)((prow * 4 * W + g * W + seg * 8) * 2), (trow * 4 * W + u0) * 2u,
                  *reinterpret_cast<const uint4*>(st_g + (g * 16 + prow) * 64 + seg * 8));
      } else if (q < 768) {                // cell state: [row] x 16 pieces of 4 units
        const int prow = (q - 512) >> 4, seg = q & 15;
        if (r0 + prow < B)
          store16(rs_c, (unsigned)((prow * W + seg * 4) * 4), ((trow + B) * W + u0) * 4u,
                  *reinterpret_cast<const uint4*>(st_c + prow * 64 + seg * 4));
      } else {                             // masked outputs: [row] x 8 pieces of 8 units
        const int prow = (q - 768) >> 3, seg = q & 7;
        if (r0 + prow < B)
          store16(rs_hd, (unsigned)((prow * W + seg * 8) * 2), (trow * W + u0) * 2u,
                  *reinterpret_cast<const uint4*>(st_hd + prow * 64 + seg * 8));
      }
      if (q < 256) {   // transposed copies: 64 units x 16 rows, two 16-byte stores per unit (x2 buffers)
        const int which = q >> 7, unit = (q >> 1) & 63, half = q & 1;
        if (r0 + half * 8 < B && (which ? a.HdT != nullptr : a.HT != nullptr)) {
          const uint4 v = *reinterpret_cast<const uint4*>(tr + (which * 64 + unit) * 16 + half * 8);
          if (which) store16(rs_hdt, (unsigned)(((long)(u0 + unit) * a.ldt_d + half * 8) * 2), trow * 2u, v);
          else store16(rs_ht, (unsigned)(((long)(u0 + unit) * a.ldt + half * 8) * 2), (trow + B) * 2u, v);
        }
      }
  };
  for (int t = 0; t < T; ++t) {
#pragma unroll 1
    for (int i = 0; i < MAXRB; ++i) {
      const int rb = rg + i * n_rg;
      if (rb >= n_rb) continue;
      const int r0 = rb * 16;
      const int erow = min(r0 + er, B - 1);          // wave-uniform
      SSTAMP(0);
      if (!SENT) {
        if (tid == 0) {
          bool ok = alive;
          if (ok && t > 0) ok = poll_counter(cnt_own + (long)rb * T + (t - 1), 2 * NWG_RB, status);
          ok_flag = ok ? 1 : 0;
        }
        SSTAMP(1);
        __syncthreads();
        SSTAMP(2);
        alive = ok_flag != 0;
        if (wave < KSTEPS) {   // one fragment of the 16 x W tile of h[t-1] per wave
          const int arow = min(r0 + (lane & 15), B - 1);
          const uint4 v = alive ? load16_sc1(rs_h, (unsigned)((((long)t * B + arow) * W + wave * 32 + kq) * 2)) : uint4{0, 0, 0, 0};
          *reinterpret_cast<uint4*>(a_tile + wave * 1024 + lane * 16) = v;
        }
      } else if (wave < KSTEPS) {
        // sentinel hand-off: every wave fetches its own fragment and re-fetches it until all 64
        // granules are there -- no counter, no poll lane, no barrier in front of the loads
        const int arow = min(r0 + (lane & 15), B - 1);
        const unsigned off = (unsigned)((((long)t * B + arow) * W + wave * 32 + kq) * 2);
        uint4 v = uint4{0, 0, 0, 0};
        bool in_lds = false;
        if (alive) {
          bool ok = false;
          if (PREF && pf_issued) {
            wait_vm(pf_after);
            v = *reinterpret_cast<const uint4*>(a_tile + wave * 1024 + lane * 16);
            ok = __all(granule_valid(v));
            in_lds = ok;
#ifdef KL_STAMP
            if (!ok && blockIdx.x == STAMP_WG && threadIdx.x == 0) stamp_lds[12] += 1;   // a prefetch that came too early
#endif
          }
          if (!ok) for (unsigned spin = 0; spin < SPIN_LIMIT; ++spin) {
            v = local ? load16_nt(rs_h, off) : load16_sc1(rs_h, off);
            if (__all(t == 0 || granule_valid(v))) { ok = true; break; }
            if ((spin & 63) == 63 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
            __builtin_amdgcn_s_sleep(2);
          }
          if (!ok) {
            __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok_flag = 0;
            v = uint4{0, 0, 0, 0};
          }
        }
        if (!in_lds) *reinterpret_cast<uint4*>(a_tile + wave * 1024 + lane * 16) = v;
      }
      // gate inputs that do not depend on the hand-off, issued behind the tile fetch: they are
      // consumed two barriers later, and their wait must not sit in front of the tile's (vmcnt is
      // in order: a wait here would also wait out the previous step's write-through stores).
      // The row is wave-uniform, so the row bases are scalar and each load takes one lane offset.
      // (raw loads only: the sums are formed in the epilogue, behind a compiler fence, or the waits
      // for these loads would be scheduled right here)
      float za[4], zb[4] = {0.f, 0.f, 0.f, 0.f};
      if (P) {
        const float* p = P + ((long)t * B + erow) * 4 * W + u0;
#pragma unroll
        for (int g = 0; g < 4; ++g) za[g] = p[g * W + eu];
      } else {
        const float* e = a.EK + (long)id_cur * 4 * W + u0;
#pragma unroll
        for (int g = 0; g < 4; ++g) za[g] = e[g * W + eu];
        if (a.n_ctx > 0) {
          const float* q = a.CtxK[0] + (long)c0_cur * 4 * W + u0;
#pragma unroll
          for (int g = 0; g < 4; ++g) zb[g] = q[g * W + eu];
        }
        for (int n = 1; n < a.n_ctx; ++n) {     // further context variables (rare): summed right away
          const float* q = a.CtxK[n] + (long)sload_i32(a.ctx + ((long)erow * T + t) * a.n_ctx + n) * 4 * W + u0;
#pragma unroll
          for (int g = 0; g < 4; ++g) zb[g] += q[g * W + eu];
        }
      }
      float mk = 1.f;
      if (maskl) mk = maskl[(long)erow * W + u0 + eu];
      SSTAMP(3);
      __syncthreads();
      SSTAMP(4);
      if (SENT) alive = ok_flag != 0;
      if (LATE && have_prev && wave >= 2) offchain(trow_prev, r0_prev);
      f32x4 acc[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < KQ; ++j) {
        frag16 fa;
        fa.u = *reinterpret_cast<const uint4*>(a_tile + (kq4 * KQ + j) * 1024 + lane * 16);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          frag16 fb;
          fb.u = bu[g][j];
          acc[g] = mfma16(fa.v, fb.v, acc[g]);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 4; ++r) zt[wave][g][(lane >> 4) * 4 + r][lane & 15] = acc[g][r];
      // next (step, row block) this workgroup visits
      int ni = i + 1, nt = t;
      if (ni >= MAXRB || rg + ni * n_rg >= n_rb) { ni = 0; nt = t + 1; }
      const int nr0 = (rg + ni * n_rg) * 16;
      if (MAXRB > 1 && !P && nt < T) {   // its table row ids (scalar loads; their wait hides among the waves arriving at the barrier)
        const long nsrc = (long)min(nr0 + er, B - 1) * T + nt;
        id_cur = sload_i32(a.idx + nsrc);
        if (a.n_ctx > 0) c0_cur = sload_i32(a.ctx + nsrc * a.n_ctx);
      }
      SSTAMP(5);
      __syncthreads();
      SSTAMP(6);
      if (PREF) {
        // the gate-input loads are waited for here (issued two phases ago), on every path: a wait further
        // down would also wait for the DMA (vmcnt is in order)
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(za[g]), "+v"(zb[g]));
        asm volatile("" : "+v"(mk));
        SSTAMP(11);
        pf_issued = 0;
        if (pref_ok && alive && nt < T && wave < KSTEPS) {
          const unsigned noff = (unsigned)((((long)nt * B + nr0 + (lane & 15)) * W + wave * 32 + kq) * 2);
          if (local) glds16_nt(rs_h, noff, lds_a + wave * 1024);
          else glds16_sc1(rs_h, noff, lds_a + wave * 1024);
          pf_issued = 1;
        }
      }
      const int wz = (eu >> 4) * 4, ue = eu & 15;
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        asm volatile("" : "+v"(za[g]), "+v"(zb[g]));     // first use of the gate-input loads: here, not earlier
        z[g] = (za[g] + zb[g] + bias_l[g * 64 + eu]) + zt[wz][g][er][ue] + zt[wz + 1][g][er][ue] + zt[wz + 2][g][er][ue] + zt[wz + 3][g][er][ue];
      }
      const float gi = fast_sigmoid(z[0]), gf = fast_sigmoid(z[1]), gg = fast_tanh(z[2]), go = fast_sigmoid(z[3]);
      const float c = gf * (MAXRB > 1 ? c_slot[i * 1024] : c_one) + gi * gg;
      if (MAXRB > 1) c_slot[i * 1024] = c;
      else c_one = c;
      const float h = go * fast_tanh(c);
      const bool row_ok = (r0 + er) < B;
      const unsigned hb = f2bf(h), hdb = f2bf(h * mk);
      // everything leaves through LDS as whole 16-byte pieces: the publish (waves 0, 1) and what
      // only later launches read (waves 2..15)
      pub[er * 64 + eu] = (bf16_t)hb;
      if (a.HT) tr[eu * 16 + er] = row_ok ? (bf16_t)hb : (bf16_t)0;
      if (a.HdT) tr[(64 + eu) * 16 + er] = row_ok ? (bf16_t)hdb : (bf16_t)0;
      st_c[er * 64 + eu] = c;
      st_hd[er * 64 + eu] = (bf16_t)hdb;
      st_g[(0 * 16 + er) * 64 + eu] = f2bf(gi);
      st_g[(1 * 16 + er) * 64 + eu] = f2bf(gf);
      st_g[(2 * 16 + er) * 64 + eu] = f2bf(gg);
      st_g[(3 * 16 + er) * 64 + eu] = f2bf(go);
      SSTAMP(7);
      __syncthreads();
      SSTAMP(8);
      const unsigned trow = (unsigned)(t * B + r0);      // first time-major row of this tile (uniform)
      // (lane offsets of the stores are re-derived every step from an opaque copy of the thread id:
      // hoisted out of the loop as invariants they were spilled to scratch at the 128-register cap)
      int stid = tid;
      if (MAXRB > 1) asm volatile("" : "+v"(stid));
      if (wave < 2) {
        // publish h[t]: two waves, one 16-byte write-through store per lane
        if (!SENT && MAXRB > 1 && pend >= 0) {
          // counters, several row blocks per workgroup: the previous block's drain + signal happen
          // here, its write-through latency hidden behind this block's step
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          signal_wave(cnt_own + pend, lane);
        }
        const int prow = stid >> 3, seg = stid & 7;
        if (alive && r0 + prow < B) {
          const uint4 v = *reinterpret_cast<const uint4*>(pub + prow * 64 + seg * 8);
          const unsigned off = (unsigned)((prow * W + seg * 8) * 2) + ((trow + B) * W + u0) * 2u;
          if (SENT && local) store16(rs_h, off, 0u, v);        // stays in this XCD's L2, where all its readers are
          else store16_sc1(rs_h, off, v);
        }
        if (SENT) {
          // the data is its own signal: nothing to drain, nothing to count
        } else if (MAXRB > 1 && defer) {
          pend = rb * T + t;
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          SSTAMP(9);
          signal_wave(cnt_own + (long)rb * T + t, lane);
        }
        SSTAMP(10);
      } else if (!LATE && alive) {
        offchain(trow, r0);
      }
      if (LATE) {
        trow_prev = trow;
        r0_prev = r0;
        have_prev = alive;
      }
      if (MAXRB == 1 && !P && nt < T) {   // one block per workgroup: the ids of the next step, behind the stores
        const long nsrc = (long)min(nr0 + er, B - 1) * T + nt;
        id_cur = sload_i32(a.idx + nsrc);
        if (a.n_ctx > 0) c0_cur = sload_i32(a.ctx + nsrc * a.n_ctx);
      }
    }
  }
  if (!SENT && MAXRB > 1 && wave < 2 && pend >= 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    signal_wave(cnt_own + pend, lane);
  }
  if (LATE && have_prev && wave >= 2) offchain(trow_prev, r0_prev);     // the last block's
  SSTAMP_FLUSH();
}

// Compute units of the current device (256 on MI355X).  The scans are sized for co-residency of
// every workgroup, so their grids follow this number instead of assuming it.
int scan_cus() {
  static int v = 0;
  if (!v) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      v = prop.multiProcessorCount;
    else
      v = 256;
    if (v > 256) v = 256;      // (the row-group arithmetic below was laid out for at most 256)
  }
  return v;
}

// Upper bound of co-resident thin-scan workgroups (KL_SCAN_WGS overrides; default two per CU).
int scan_max_wgs() {
  static int v = 0;
  if (!v) {
    const char* e = getenv("KL_SCAN_WGS");
    v = e ? atoi(e) : 2 * scan_cus();
    if (v < 64) v = 64;
    if (v > 512) v = 512;
  }
  return v;
}

// grid plan shared by both scans; false = this shape must use the launch-per-step path
bool plan_scan(int W, int L, int B, int T, int* n_rb, int* n_rg, int* per_wg) {
  if (W != 512 && W != 256 && W != 128 && W != 64 && !(W == 1024 && L == 1)) return false;      // (width 1024: one layer per launch)
  if (L < 1 || L > KL_SCAN_MAXL || B < 1 || T < 1) return false;
  const int col_tasks = L * (W / 16);
  if (col_tasks > 256) return false;
  *n_rb = (B + 15) / 16;
  int g = scan_max_wgs() / col_tasks;
  if (g > *n_rb) g = *n_rb;
  if (g < 1) return false;
  *n_rg = g;
  *per_wg = (*n_rb + g - 1) / g;
  return *per_wg <= ((W == 512 || W == 1024) ? 8 : 4);      // (the dispatch tables: 8 row blocks per workgroup at widths 512 and 1024, else 4)
}

}  // namespace

#define KL_SCAN_CASE(KERNEL, KS, RB) hipLaunchKernelGGL((KERNEL<KS, RB>), grid, block, 0, stream, a)
#define KL_SCAN_DISPATCH(KERNEL)                                                                      \
  do {                                                                                                \
    if (W == 512) {                                                                                   \
      if (per_wg == 1) KL_SCAN_CASE(KERNEL, 16, 1);                                                   \
      else if (per_wg == 2) KL_SCAN_CASE(KERNEL, 16, 2);                                              \
      else if (per_wg <= 4) KL_SCAN_CASE(KERNEL, 16, 4);                                              \
      else KL_SCAN_CASE(KERNEL, 16, 8);                                                               \
    } else if (W == 256) {                                                                            \
      if (per_wg == 1) KL_SCAN_CASE(KERNEL, 8, 1);                                                    \
      else if (per_wg == 2) KL_SCAN_CASE(KERNEL, 8, 2);                                               \
      else KL_SCAN_CASE(KERNEL, 8, 4);                                                                \
    } else if (W == 128) {                                                                            \
      if (per_wg == 1) KL_SCAN_CASE(KERNEL, 4, 1);                                                    \
      else if (per_wg == 2) KL_SCAN_CASE(KERNEL, 4, 2);                                               \
      else KL_SCAN_CASE(KERNEL, 4, 4);                                                                \
    } else {                                                                                          \
      if (per_wg == 1) KL_SCAN_CASE(KERNEL, 2, 1);                                                    \
      else if (per_wg == 2) KL_SCAN_CASE(KERNEL, 2, 2);                                               \
      else KL_SCAN_CASE(KERNEL, 2, 4);                                                                \
    }                                                                                                 \
  } while (0)

// Persistent forward scan; KL_ERR_SHAPE = use the launch-per-step path instead.
int kl_launch_scan_fwd(KlScanFwd a, hipStream_t stream) {
  const int W = a.W;
  int per_wg = 0;
  if (!plan_scan(W, a.L, a.B, a.T, &a.n_rb, &a.n_rg, &per_wg)) return KL_ERR_SHAPE;
  if ((long)(a.T + 1) * a.B * W * 2 > 0xfffffff0L) return KL_ERR_SHAPE;      // unsigned 32-bit buffer offsets into H
  // (the caller has zeroed a.counters -- L*ceil(B/16)*T words -- and a.status, write-through)
  dim3 grid(a.L * (W / 16) * a.n_rg), block(256);
  if (W == 1024) {      // one layer per launch, input side precomputed in P1 (U alone takes 128 of the 256 registers)
#define KL_SCAN_CASE3(KERNEL, KS, RB) hipLaunchKernelGGL((KERNEL<KS, RB, false>), grid, block, 0, stream, a)
    if (per_wg == 1) KL_SCAN_CASE3(lstm_scan_fwd_kernel, 32, 1);
    else if (per_wg == 2) KL_SCAN_CASE3(lstm_scan_fwd_kernel, 32, 2);
    else if (per_wg <= 4) KL_SCAN_CASE3(lstm_scan_fwd_kernel, 32, 4);
    else KL_SCAN_CASE3(lstm_scan_fwd_kernel, 32, 8);
    return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
  }
  KL_SCAN_DISPATCH(lstm_scan_fwd_kernel);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

int kl_launch_scan_bwd(KlScanBwd a, hipStream_t stream) {
  const int W = a.W;
  int per_wg = 0;
  if (!plan_scan(W, a.L, a.B, a.T, &a.n_rb, &a.n_rg, &per_wg)) return KL_ERR_SHAPE;
  if ((long)a.T * a.B * 4 * W * 2 > 0xfffffff0L) return KL_ERR_SHAPE;        // unsigned 32-bit buffer offsets into dZ
  // (the caller has zeroed a.counters -- L*ceil(B/16)*T words -- and a.status, write-through)
  dim3 grid(a.L * (W / 16) * a.n_rg), block(256);
  if (W == 1024) {      // one layer per launch (the caller's layer-sequential path)
    if (per_wg == 1) KL_SCAN_CASE3(lstm_scan_bwd_kernel, 32, 1);
    else if (per_wg == 2) KL_SCAN_CASE3(lstm_scan_bwd_kernel, 32, 2);
    else if (per_wg <= 4) KL_SCAN_CASE3(lstm_scan_bwd_kernel, 32, 4);
    else KL_SCAN_CASE3(lstm_scan_bwd_kernel, 32, 8);
#undef KL_SCAN_CASE3
    return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
  }
  KL_SCAN_DISPATCH(lstm_scan_bwd_kernel);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

#ifdef KL_STAMP
extern "C" int kl_test_scan_stamps(unsigned long long* out, int reset) {
  if (reset) {
    unsigned long long z[32] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(kl_scan_stamps), z, sizeof(z)) == hipSuccess ? 0 : KL_ERR_LAUNCH;
  }
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(kl_scan_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : KL_ERR_LAUNCH;
}
#endif

// One-layer backward scan with 64-unit workgroups (a.L must be 1).  KL_ERR_SHAPE = not applicable.
// shapes the wide backward scan serves (the launcher applies the same test)
bool kl_scan_bwd_wide_applicable(int B, int T, int W) {
  if ((W != 512 && W != 256) || B < 1 || T < 1) return false;
  const int n_rb = (B + 15) / 16, col_groups = W / 64;
  int g = scan_cus() / col_groups;
  if (g < 1) return false;
  if (g > n_rb) g = n_rb;
  if ((n_rb + g - 1) / g > 8) return false;
  return (long)T * B * 4 * W * 2 <= 0xfffffff0L;      // unsigned 32-bit buffer offsets
}

// row blocks each workgroup of a wide scan serves per step (both wide scans use the same grid plan)
int kl_scan_wide_blocks_per_wg(int B, int W) {
  const int n_rb = (B + 15) / 16, col_groups = W / 64;
  int g = col_groups > 0 ? scan_cus() / col_groups : 0;
  if (g < 1) return 0;
  if (g > n_rb) g = n_rb;
  return (n_rb + g - 1) / g;
}

int kl_launch_scan_bwd_wide(KlScanBwd a, hipStream_t stream) {
  const int W = a.W;
  if (a.L != 1 || !kl_scan_bwd_wide_applicable(a.B, a.T, W)) return KL_ERR_SHAPE;
  if (a.dZT && ((a.ldt & 7) || (a.B & 7))) return KL_ERR_SHAPE;
  a.n_rb = (a.B + 15) / 16;
  const int col_groups = W / 64;
  int g = scan_cus() / col_groups;     // one 1024-thread workgroup per CU
  if (g > a.n_rb) g = a.n_rb;
  a.n_rg = g;
  const int per_wg = (a.n_rb + g - 1) / g;
  if (per_wg > 8) return KL_ERR_SHAPE;
  dim3 grid(8 * col_groups * ((g + 7) / 8)), block(1024);     // 8 XCDs x column groups x row groups per XCD (surplus ones exit)
  const size_t lds = (size_t)KL_BWD_WIDE_LDS(W / 32) + (per_wg > 4 ? 8 : (per_wg > 1 ? 4 : 0)) * 1024 * sizeof(float);
#define KL_WIDE_CASE2(KS, RB, S)                                                                                     \
  do {                                                                                                               \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_scan_bwd_wide_kernel<KS, RB, S>),                  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return KL_ERR_LAUNCH; \
    hipLaunchKernelGGL((lstm_scan_bwd_wide_kernel<KS, RB, S>), grid, block, lds, stream, a);                         \
  } while (0)
#define KL_WIDE_CASE(KS, RB)                          \
  do {                                                \
    if (a.sentinel) KL_WIDE_CASE2(KS, RB, true);      \
    else KL_WIDE_CASE2(KS, RB, false);                \
  } while (0)
  if (W == 512) { if (per_wg == 1) KL_WIDE_CASE(16, 1); else if (per_wg == 2) KL_WIDE_CASE(16, 2); else if (per_wg <= 4) KL_WIDE_CASE(16, 4); else KL_WIDE_CASE(16, 8); }
  else { if (per_wg == 1) KL_WIDE_CASE(8, 1); else if (per_wg == 2) KL_WIDE_CASE(8, 2); else if (per_wg <= 4) KL_WIDE_CASE(8, 4); else KL_WIDE_CASE(8, 8); }
#undef KL_WIDE_CASE2
#undef KL_WIDE_CASE
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

// shapes the wide forward scan serves (the launcher applies the same test)
bool kl_scan_fwd_wide_applicable(int B, int T, int W) {
  if ((W != 512 && W != 256) || B < 1 || T < 1 || (B & 7)) return false;
  const int n_rb = (B + 15) / 16, col_groups = W / 64;
  int g = scan_cus() / col_groups;
  if (g < 1) return false;
  if (g > n_rb) g = n_rb;
  if ((n_rb + g - 1) / g > 8) return false;
  return (long)T * B * 4 * W * 2 <= 0xfffffff0L;      // unsigned 32-bit buffer offsets (the gate rows are the largest)
}

// One-layer forward scan with 64-unit workgroups.  KL_ERR_SHAPE = not applicable.
int kl_launch_scan_fwd_wide(KlScanFwdWide a, hipStream_t stream) {
  const int W = a.W;
  if (!kl_scan_fwd_wide_applicable(a.B, a.T, W)) return KL_ERR_SHAPE;
  if ((a.HT && (a.ldt & 7)) || (a.HdT && (a.ldt_d & 7))) return KL_ERR_SHAPE;
  if (!a.P && (!a.EK || !a.idx || !a.bias || a.n_ctx > 8)) return KL_ERR_ARG;
  a.n_rb = (a.B + 15) / 16;
  const int col_groups = W / 64;
  int g = scan_cus() / col_groups;
  if (g < 1) return KL_ERR_SHAPE;
  if (g > a.n_rb) g = a.n_rb;
  a.n_rg = g;
  const int per_wg = (a.n_rb + g - 1) / g;
  if (per_wg > 8) return KL_ERR_SHAPE;
  dim3 grid(8 * col_groups * ((g + 7) / 8)), block(1024);     // 8 XCDs x column groups x row groups per XCD (surplus ones exit)
  const size_t lds = (size_t)KL_FWD_WIDE_LDS(W / 32) + (per_wg > 4 ? 8 : (per_wg > 1 ? 4 : 0)) * 1024 * sizeof(float);
  if ((long)a.T * a.B * 4 * W * 2 > 0xfffffff0L) return KL_ERR_SHAPE;   // unsigned 32-bit buffer offsets (the gate rows are the largest)
#define KL_WIDE_CASE2(KS, RB, S)                                                                                     \
  do {                                                                                                               \
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_scan_fwd_wide_kernel<KS, RB, S>),                  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return KL_ERR_LAUNCH; \
    hipLaunchKernelGGL((lstm_scan_fwd_wide_kernel<KS, RB, S>), grid, block, lds, stream, a);                         \
  } while (0)
#define KL_WIDE_CASE(KS, RB)                          \
  do {                                                \
    if (a.sentinel) KL_WIDE_CASE2(KS, RB, true);      \
    else KL_WIDE_CASE2(KS, RB, false);                \
  } while (0)
  if (W == 512) { if (per_wg == 1) KL_WIDE_CASE(16, 1); else if (per_wg == 2) KL_WIDE_CASE(16, 2); else if (per_wg <= 4) KL_WIDE_CASE(16, 4); else KL_WIDE_CASE(16, 8); }
  else { if (per_wg == 1) KL_WIDE_CASE(8, 1); else if (per_wg == 2) KL_WIDE_CASE(8, 2); else if (per_wg <= 4) KL_WIDE_CASE(8, 4); else KL_WIDE_CASE(8, 8); }
#undef KL_WIDE_CASE2
#undef KL_WIDE_CASE
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}

// Split-precision inference scan (all layers fused).  KL_ERR_SHAPE = not applicable.
int kl_launch_scan_fwd_split(KlScanFwdSplit a, hipStream_t stream) {
  const int W = a.W;
  if (W == 1024 && a.units8) {
    // a.L (one or two) layers from a.l0 as a wavefront, eight units per workgroup, one workgroup per CU
    if (a.L < 1 || a.L > 2 || !a.sentinel || a.B < 1 || a.T < 1) return KL_ERR_SHAPE;
    if ((long)(a.T + 1) * a.B * W * 2 > 0x7fffffffL) return KL_ERR_SHAPE;
    a.n_rb = (a.B + 15) / 16;
    const int col_tasks = a.L * (W / 8);
    int g = scan_cus() / col_tasks;
    if (g < 1) return KL_ERR_SHAPE;
    if (g > a.n_rb) g = a.n_rb;
    a.n_rg = g;
    const int per_wg = (a.n_rb + g - 1) / g;
    if (per_wg > 2) return KL_ERR_SHAPE;
    dim3 grid(col_tasks * g), block(256);
    if (per_wg == 1) hipLaunchKernelGGL((lstm_scan_fwd_split_sent8_kernel<32, 1>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((lstm_scan_fwd_split_sent8_kernel<32, 2>), grid, block, 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
  }
  if (W == 1024) {
    // one layer per launch, the input side precomputed in P1, data sentinels only (the caller's layer-sequential path)
    if (a.L != 1 || !a.sentinel || a.B < 1 || a.T < 1) return KL_ERR_SHAPE;
    if ((long)(a.T + 1) * a.B * W * 2 > 0x7fffffffL) return KL_ERR_SHAPE;
    a.n_rb = (a.B + 15) / 16;
    int g = scan_cus() / (W / 16);
    if (g < 1) return KL_ERR_SHAPE;
    if (g > a.n_rb) g = a.n_rb;
    a.n_rg = g;
    const int per_wg = (a.n_rb + g - 1) / g;
    if (per_wg > 4) return KL_ERR_SHAPE;
    dim3 grid((W / 16) * g), block(256);
    if (per_wg == 1) hipLaunchKernelGGL((lstm_scan_fwd_split_sent_kernel<32, 1, false>), grid, block, 0, stream, a);
    else if (per_wg == 2) hipLaunchKernelGGL((lstm_scan_fwd_split_sent_kernel<32, 2, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((lstm_scan_fwd_split_sent_kernel<32, 4, false>), grid, block, 0, stream, a);
    return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
  }
  if (W != 512 && W != 256 && W != 128 && W != 64) return KL_ERR_SHAPE;
  if (a.L < 1 || a.L > KL_SCAN_MAXL || a.B < 1 || a.T < 1) return KL_ERR_SHAPE;
  const int col_tasks = a.L * (W / 16);
  if (col_tasks > 256) return KL_ERR_SHAPE;
  if ((long)(a.T + 1) * a.B * W * 2 > 0x7fffffffL) return KL_ERR_SHAPE;
  a.n_rb = (a.B + 15) / 16;
  int g = scan_cus() / col_tasks;        // one workgroup per CU (the weights fill the register file)
  if (g < 1) return KL_ERR_SHAPE;
  if (g > a.n_rb) g = a.n_rb;
  a.n_rg = g;
  const int per_wg = (a.n_rb + g - 1) / g;
  if (per_wg > 4) return KL_ERR_SHAPE;
  dim3 grid(col_tasks * g), block(256);
  if (a.sentinel) KL_SCAN_DISPATCH(lstm_scan_fwd_split_sent_kernel);
  else KL_SCAN_DISPATCH(lstm_scan_fwd_split_kernel);
  return hipGetLastError() == hipSuccess ? 0 : KL_ERR_LAUNCH;
}
