// Device helpers of the second-generation scans (lstm_scan2.hip, lstm_scan_fwd8.hip): asynchronous loads the compiler does
// not know about, LDS-DMA forms, the 4 x 4 quad transpose, armed landing zones.  Included inside each file's anonymous
// namespace, after kl_scan_common.h.
#pragma once

typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// ---- asynchronous loads the compiler does not know about (waited for by count, wait_vm) ----
// The destination registers count as written at the end of the asm statement; every consumer sits
// behind a wait + "+v" fence (use_regs) so that nothing reads them before the data has landed.
// The destinations are read-write operands: the caller sets them to all-ones first, which no valid datum is, so
// that "has landed" can be CHECKED after an optimistic counted wait (a register with a load in flight reads
// as its old value).
__device__ __forceinline__ void aload16_glb(u32x4& d, const void* sbase, unsigned voff) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void aload8_glb(u32x2& d, const void* sbase, unsigned voff) {
  asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2" : "+v"(d) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void aload4_glb(unsigned& d, const void* sbase, unsigned voff) {
  asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "+v"(d) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void aload2_glb(unsigned& d, const void* sbase, unsigned voff) {
  asm volatile("s_nop 4\n\tglobal_load_ushort %0, %1, %2" : "+v"(d) : "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void aload16_buf(u32x4& d, __amdgpu_buffer_rsrc_t r, unsigned voff) {
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(d) : "v"(voff), "s"(r) : "memory");
}
__device__ __forceinline__ void use_regs(u32x4& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void use_regs(u32x2& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void use_regs(unsigned& a) { asm volatile("" : "+v"(a)); }

// LDS-DMA of one 1 KiB tile piece: lane l lands at lds_addr + 16 l; source = lane offset + scalar offset
__device__ __forceinline__ void glds16_sc1_s(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr)
      : "memory");
}

// ... as a streaming ("nt") load: served by the XCD's L2 without the coherence actions of an sc1 load -- for partners
// that were verified to share that L2 (XCD-local hand-off, kl_scan_common.h)
__device__ __forceinline__ void glds16_nt_s(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr)
      : "memory");
}

__device__ __forceinline__ float dpp_x1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float dpp_x2(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
}
// lane k of a quad holds m[r] = M[k][r]; afterwards m[r] = M[r][k]
__device__ __forceinline__ void quad_transpose(f32x4& m, int k) {
  const bool odd = k & 1, hi = k & 2;
  {
    const float s0 = odd ? m[0] : m[1], s1 = odd ? m[2] : m[3];
    const float r0 = dpp_x1(s0), r1 = dpp_x1(s1);
    if (odd) { m[0] = r0; m[2] = r1; } else { m[1] = r0; m[3] = r1; }
  }
  {
    const float s0 = hi ? m[0] : m[2], s1 = hi ? m[1] : m[3];
    const float r0 = dpp_x2(s0), r1 = dpp_x2(s1);
    if (hi) { m[0] = r0; m[1] = r1; } else { m[2] = r0; m[3] = r1; }
  }
}

__device__ __forceinline__ float u2f(unsigned x) { return __builtin_bit_cast(float, x); }

// Every LDS landing zone of a DMA is ARMED with 0xFFFFFFFF words before the DMA is issued and checked after the
// counted wait: the count is a good estimate of "landed", not a proof -- with stores among the younger operations the
// wait was observed to pass before an older load's data had arrived (B = 1536: a few lanes of the gate-input pieces
// still held the previous phase's values), and a tile buffer still holds the VALID-looking tile of two phases ago.
// 0xFFFFFFFF is neither a finite float, nor a pair of finite bf16, nor a table-row offset.
__device__ __forceinline__ void arm16(unsigned char* p) {
  *reinterpret_cast<uint4*>(p) = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
}
// a 16-byte piece of a tile is there once none of its four dwords is all-ones (armed LDS, or the sentinel the
// producers' buffer was pre-filled with): dword writes are atomic and two finite bf16 never make 0xFFFFFFFF
__device__ __forceinline__ bool piece_there(const unsigned char* p) {
  const u32x4 v = *reinterpret_cast<const u32x4*>(p);
  return max(max(v.x, v.y), max(v.z, v.w)) != 0xFFFFFFFFu;
}
__device__ __forceinline__ bool landed16(const unsigned char* p) {      // (re-read on every call)
  asm volatile("" ::: "memory");
  const u32x4 v = *reinterpret_cast<const u32x4*>(p);
  return v.x != 0xFFFFFFFFu && v.y != 0xFFFFFFFFu && v.z != 0xFFFFFFFFu && v.w != 0xFFFFFFFFu;
}

// plain LDS-DMA of one 1 KiB piece (lane l lands at lds_addr + 16 l); source = lane offset into the buffer
__device__ __forceinline__ void glds16_plain(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(lds_addr)
      : "memory");
}
// ... of 256 bytes (lane l lands at lds_addr + 4 l); source = lane offset + scalar offset
__device__ __forceinline__ void glds4_plain_s(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 4\n\tbuffer_load_dword %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr)
      : "memory");
}

// ... of 16 bytes per lane from rsrc[voff + soff] (plain load: data of an earlier kernel); lane l lands at lds_addr + 16 l
__device__ __forceinline__ void glds16_plain_s(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr)
      : "memory");
}

// ... coherent across XCDs (hand-off flags)
__device__ __forceinline__ void glds4_sc1_s(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 4\n\tbuffer_load_dword %1, %2, %3 offen sc1 lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr)
      : "memory");
}

// wave-wide reductions by DPP (no LDS crossbar: `__shfl_xor` is a ds_bpermute of ~100 cycles each, eighteen of them in a
// dependent chain per softmax row).  Every step combines with a lane pattern inside the 16-lane rows, the last two carry the
// row results across (row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3); the result is lane 63's, broadcast
// through a scalar.  Lanes a step does not write see the operation's identity.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float identity, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_max(float v) {
  const float ninf = -INFINITY;
  v = fmaxf(v, dpp_f<0xB1, 0xF>(ninf, v));
  v = fmaxf(v, dpp_f<0x4E, 0xF>(ninf, v));
  v = fmaxf(v, dpp_f<0x141, 0xF>(ninf, v));
  v = fmaxf(v, dpp_f<0x140, 0xF>(ninf, v));
  v = fmaxf(v, dpp_f<0x142, 0xA>(ninf, v));
  v = fmaxf(v, dpp_f<0x143, 0xC>(ninf, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1, 0xF>(0.f, v);
  v += dpp_f<0x4E, 0xF>(0.f, v);
  v += dpp_f<0x141, 0xF>(0.f, v);
  v += dpp_f<0x140, 0xF>(0.f, v);
  v += dpp_f<0x142, 0xA>(0.f, v);
  v += dpp_f<0x143, 0xC>(0.f, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ int wave_min_i(int v) {      // (non-negative values: compared as floats' bit patterns would be, done on ints)
  const int big = 0x7fffffff;
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0xB1, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x4E, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x141, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x140, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x142, 0xA, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x143, 0xC, 0xF, false));
  return __builtin_amdgcn_readlane(v, 63);
}

// ... the same inside every 16-lane row of the wave (four rows reduced at once): two quad permutations, then the mirrors of the
// half row and of the row -- after each step the partial result is uniform over twice as many lanes, the last leaves it in all 16
__device__ __forceinline__ float row16_max(float v) {
  const float ninf = -INFINITY;
  v = fmaxf(v, dpp_f<0xB1, 0xF>(ninf, v));
  v = fmaxf(v, dpp_f<0x4E, 0xF>(ninf, v));
  v = fmaxf(v, dpp_f<0x141, 0xF>(ninf, v));
  v = fmaxf(v, dpp_f<0x140, 0xF>(ninf, v));
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0xB1, 0xF>(0.f, v);
  v += dpp_f<0x4E, 0xF>(0.f, v);
  v += dpp_f<0x141, 0xF>(0.f, v);
  v += dpp_f<0x140, 0xF>(0.f, v);
  return v;
}
__device__ __forceinline__ int row16_min_i(int v) {
  const int big = 0x7fffffff;
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0xB1, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x4E, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x141, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x140, 0xF, 0xF, false));
  return v;
}

// one dword from LDS as it is NOW (a landing zone of an LDS-DMA request), by an LDS instruction.  NOT `*(volatile unsigned*)p` with a
// generic pointer: that becomes a FLAT load, which counts on vmcnt as well, and the compiler follows it with `s_waitcnt vmcnt(0)` --
// a wait for every load and store the wave has in flight, hand-counted ones included (found in the register-tile backward
// scan in round 4: it drained the wave's queue at the top of every block).
__device__ __forceinline__ unsigned lds_peek(unsigned lds_addr) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr) : "memory");
  return v;
}
