// Device helpers shared by the persistent LSTM scans (lstm_scan.hip, lstm_scan2.hip): buffer
// loads / stores with cache policies, LDS-DMA, counted waits, the sentinel test, hand-off
// counters, diagnostic cycle stamps.  Included inside each file's anonymous namespace.
#pragma once


typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr unsigned SPIN_LIMIT = 1u << 20;

// diagnostic build only (-DKL_STAMP): cycle shares of the forward scan's step phases
#ifdef KL_STAMP
#ifndef KL_STAMP_TID
#define KL_STAMP_TID 0      /* the stamped thread of the stamped workgroup */
#endif
#ifndef KL_STAMP_ARRAY
#define KL_STAMP_ARRAY kl_scan_stamps
#endif
__device__ unsigned long long KL_STAMP_ARRAY[32];   // [0,16) forward scan, [16,32) backward scan
#define SSTAMP(i)                                                                     \
  do {                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                \
    if (blockIdx.x == STAMP_WG && threadIdx.x == KL_STAMP_TID) {                      \
      const unsigned long long now_ = clock64();                                      \
      stamp_lds[i] += now_ - last_;     /* LDS: no global round trip inside the step */ \
      last_ = now_;                                                                   \
    }                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                \
  } while (0)
#define SSTAMP_INIT(wg)                                                               \
  const int STAMP_WG = (wg);                                                          \
  __shared__ unsigned long long stamp_lds[32];                                        \
  if (threadIdx.x < 32) stamp_lds[threadIdx.x] = 0;                                   \
  __syncthreads();                                                                    \
  unsigned long long last_ = clock64();
#define SSTAMP_FLUSH()                                                                \
  do {                                                                                \
    __syncthreads();                                                                  \
    if (blockIdx.x == STAMP_WG && threadIdx.x < 32 && stamp_lds[threadIdx.x])         \
      atomicAdd(&KL_STAMP_ARRAY[threadIdx.x], stamp_lds[threadIdx.x]);                \
  } while (0)
#else
#define SSTAMP(i)
#define SSTAMP_INIT(wg)
#define SSTAMP_FLUSH()
#endif

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long bytes) {
  // (num_records is an unsigned 32-bit byte count: buffers up to 4 GiB - 1)
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(unsigned)(bytes > 0xffffffffL ? 0xffffffffL : bytes), 0x00020000);
}
__device__ __forceinline__ uint4 load16_nt(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 2);
  return uint4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ uint4 load16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 16);
  return uint4{v.x, v.y, v.z, v.w};
}
// wave-uniform 4-byte load through the scalar cache (data written by an earlier launch only)
__device__ __forceinline__ int sload_i32(const int* p) {
  int v;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}
// plain 16-byte buffer store: lane offset + wave-uniform (scalar) offset
__device__ __forceinline__ void store16(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned uniform_off, uint4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, r, (int)lane_off, (int)uniform_off, 0);
}
__device__ __forceinline__ void store16_sc1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, uint4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(u32x4{v.x, v.y, v.z, v.w}, r, (int)byte_off, 0, 16);
}

// 16 bytes per lane straight into LDS (write-through-coherent read): lane l of the wave lands at
// lds_addr + 16 l.  Inline asm: the compiler must not know about the load, or it would wait for it
// at the next LDS access; the waits are counted by hand (wait_vm).  M0 is compiler-reserved.
__device__ __forceinline__ void glds16_sc1(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen sc1 lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(lds_addr)
      : "memory");
}
// the same as a streaming ("nt") load: no L1 allocation, served by the XCD's L2 without the coherence
// actions of an sc1 load -- for partners that were verified to share that L2 (XCD-local hand-off)
__device__ __forceinline__ void glds16_nt(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(lds_addr)
      : "memory");
}
// wait until at most k (wave-uniform; an under-estimate only waits longer) vector memory operations of
// this wave are in flight
__device__ __forceinline__ void wait_vm(int k) {
  // A computed jump into a table of (s_waitcnt vmcnt(i); s_branch end) pairs: seven scalar instructions whatever k is.  As a C++
  // switch the sixteen cases come out of the structuriser as a tree of compares FOLLOWED by a chain of flag tests -- ~25 scalar
  // instructions and a handful of branches per call, four calls per phase of the 16-wave forward scan.
  const unsigned kk = __builtin_amdgcn_readfirstlane((unsigned)(k < 0 ? 0 : (k > 15 ? 15 : k)));      // (wave-uniform by contract)
  unsigned off;
  asm volatile(
      "s_getpc_b64 vcc\n\t"                      // = address of the next instruction (A); vcc as the 64-bit scratch pair
      "s_lshl_b32 %0, %1, 3\n\t"                 // A + 0:  8 bytes per table entry
      "s_add_u32 %0, %0, 20\n\t"                 // A + 4:  the table starts 20 bytes behind A
      "s_add_u32 vcc_lo, vcc_lo, %0\n\t"         // A + 8
      "s_addc_u32 vcc_hi, vcc_hi, 0\n\t"         // A + 12
      "s_setpc_b64 vcc\n\t"                      // A + 16
      "s_waitcnt vmcnt(0)\n\ts_branch .Lwv_end_%=\n\t"
      "s_waitcnt vmcnt(1)\n\ts_branch .Lwv_end_%=\n\t"
      "s_waitcnt vmcnt(2)\n\ts_branch .Lwv_end_%=\n\t"
      "s_waitcnt vmcnt(3)\n\ts_branch .Lwv_end_%=\n\t"
      "s_waitcnt vmcnt(4)\n\ts_branch .Lwv_end_%=\n\t"
      "s_waitcnt vmcnt(5)\n\ts_branch .Lwv_end_%=\n\t"
      "s_waitcnt vmcnt(6)\n\ts_branch .Lwv_end_%=\n\t"
      "s_waitcnt vmcnt(7)\n\ts_branch .Lwv_end_%=\n\t"
      "s_waitcnt vmcnt(8)\n\ts_branch .Lwv_end_%=\n\t"
      "s_waitcnt vmcnt(9)\n\ts_branch .Lwv_end_%=\n\t"
      "s_waitcnt vmcnt(10)\n\ts_branch .Lwv_end_%=\n\t"
      "s_waitcnt vmcnt(11)\n\ts_branch .Lwv_end_%=\n\t"
      "s_waitcnt vmcnt(12)\n\ts_branch .Lwv_end_%=\n\t"
      "s_waitcnt vmcnt(13)\n\ts_branch .Lwv_end_%=\n\t"
      "s_waitcnt vmcnt(14)\n\ts_branch .Lwv_end_%=\n\t"
      "s_waitcnt vmcnt(15)\n"
      ".Lwv_end_%=:"
      : "=&s"(off)
      : "s"(kk)
      : "memory", "scc", "vcc");
}
typedef __attribute__((address_space(3))) void lds_void_t;

// XCD-local hand-off.  Workgroup b runs on XCD b % 8, and the wide scans place all workgroups that
// exchange data (one row group, all column groups) on one XCD.  Data written with PLAIN stores then
// stays in that XCD's L2, where the partners' L1-bypassing (sc1) loads find it: half the latency of the
// write-through path and no fabric traffic (tools/micro/handoff_xcd.hip: 284 vs 585 ns one way).  Across
// XCDs the same combination is NOT coherent, so the placement is not assumed but checked at the start
// of every launch: each workgroup posts (launch token, its XCC id) write-through, reads its partners'
// posts and takes the local path only if all of them sit on its own XCD.
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
// called by all threads; slot of workgroup b = slots[b]; partners = blockIdx of column group j for j < n_partners
template <typename F>
__device__ __forceinline__ bool xcd_local_group(unsigned* slots, unsigned gen, int n_partners, F partner_block, int* lds_flag,
                                                unsigned* status) {
  const unsigned mine = xcc_id();
  if (threadIdx.x == 0) __hip_atomic_store(slots + blockIdx.x, (gen << 4) | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x < 64) {
    bool same = true;
    if ((int)threadIdx.x < n_partners) {
      const unsigned* slot = slots + partner_block((int)threadIdx.x);
      unsigned v = 0;
      bool got = false;
      for (unsigned spin = 0; spin < SPIN_LIMIT; ++spin) {
        v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 4) == gen) { got = true; break; }
        if ((spin & 63) == 63 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;
        __builtin_amdgcn_s_sleep(2);
      }
      if (!got) __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      same = got && (v & 15u) == mine;
    }
    const bool all_same = __all(same);
    if (threadIdx.x == 0) *lds_flag = all_same ? 1 : 0;
  }
  __syncthreads();
  return *lds_flag != 0;
}

// Sentinel hand-off: exchange buffers are pre-filled with 0xFFFF halfwords (a bf16 NaN pattern
// that neither h = o * tanh(c) nor a finite gradient ever rounds to); a 16-byte granule is valid
// once none of its eight halfwords is the sentinel.  Checking all eight makes a granule that
// became visible only in part count as not yet there.
__device__ __forceinline__ bool granule_valid(uint4 v) {
  auto bad = [](unsigned x) { return ((x & 0xFFFFu) == 0xFFFFu) || ((x >> 16) == 0xFFFFu); };
  return !(bad(v.x) || bad(v.y) || bad(v.z) || bad(v.w));
}
// the same test in three operations per dword, for waves that check whole tiles: a halfword of x is 0xFFFF iff the
// halfword of ~x is zero, and (~x - 0x00010001) & x & 0x80008000 is non-zero iff ~x has a zero halfword ("haszero");
// the bits of several dwords may be OR-ed before the mask is applied
__device__ __forceinline__ unsigned sentinel_bits(uint4 v) {
  auto f = [](unsigned x) { return (~x - 0x00010001u) & x; };
  return f(v.x) | f(v.y) | f(v.z) | f(v.w);
}
__device__ __forceinline__ bool sentinel_free(unsigned bits) { return (bits & 0x80008000u) == 0; }

// ... for waves that check many granules: ONE packed instruction per dword -- the running maximum of the two halfword columns
// (v_pk_max_u16) has a 0xFFFF halfword iff some dword had one in that column; three where sentinel_bits needs thirteen per granule
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b) {
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
}
__device__ __forceinline__ unsigned sentinel_acc(unsigned m, uint4 v) {
  u16x2_t a = __builtin_bit_cast(u16x2_t, m);
  a = __builtin_elementwise_max(a, __builtin_bit_cast(u16x2_t, v.x));
  a = __builtin_elementwise_max(a, __builtin_bit_cast(u16x2_t, v.y));
  a = __builtin_elementwise_max(a, __builtin_bit_cast(u16x2_t, v.z));
  a = __builtin_elementwise_max(a, __builtin_bit_cast(u16x2_t, v.w));
  return __builtin_bit_cast(unsigned, a);
}
__device__ __forceinline__ bool sentinel_acc_free(unsigned m) { return (m & 0xFFFFu) != 0xFFFFu && (m >> 16) != 0xFFFFu; }

// v_exp_f32 / v_rcp_f32 forms (1 ulp each): the scans are latency chains, and the
// training path computes in bf16 anyway
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  return 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.f;
}

// Every publishing wave counts itself in after draining its own stores (valid form "each storing wave
// for itself", MI355X_MICROARCH.md); consumers wait for waves x workgroups arrivals.  (A/B on one box:
// funnelling the arrivals through an LDS counter so that only ONE wave per workgroup adds was 3 % slower.)
__device__ __forceinline__ void signal_wave(unsigned* counter, int lane) {
  if (lane == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one lane polls; true = reached.  Raises/observes the abort word.
__device__ __forceinline__ bool poll_counter(const unsigned* cnt, unsigned target, unsigned* status) {
  for (unsigned spin = 0; spin < SPIN_LIMIT; ++spin) {
    if (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) return true;
    if ((spin & 63) == 63 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}

