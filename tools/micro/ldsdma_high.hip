// Does LDS-DMA (buffer_load ... lds, M0 = destination) reach LDS addresses above 64 KiB on gfx950?
// One workgroup, 160 KiB of dynamic LDS; each probe lands 1 KiB at a given LDS byte address and reads it back.
// build: hipcc --offload-arch=gfx950 -O2 -o ldsdma_high ldsdma_high.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void lds_void_t;

__global__ void probe(const unsigned* src, unsigned* out, const unsigned* addrs, int n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x;
  const unsigned base = (unsigned)(size_t)(lds_void_t*)smem;
  for (int i = lane; i < 163840 / 4; i += 64) reinterpret_cast<unsigned*>(smem)[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, 1 << 20, 0x00020000);
  for (int k = 0; k < n; ++k) {
    const unsigned dst = __builtin_amdgcn_readfirstlane(base + addrs[k]);
    unsigned keep;
    const unsigned voff = (unsigned)(k * 1024 + lane * 16);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(dst) : "memory");
    __syncthreads();
    const u32x4 v = *reinterpret_cast<const u32x4*>(smem + addrs[k] + lane * 16);
    out[(k * 64 + lane) * 4 + 0] = v.x; out[(k * 64 + lane) * 4 + 1] = v.y;
    out[(k * 64 + lane) * 4 + 2] = v.z; out[(k * 64 + lane) * 4 + 3] = v.w;
  }
}

int main() {
  const std::vector<unsigned> addrs = {0, 61440, 65536, 66560, 98304, 131072, 147456, 162816};
  const int n = (int)addrs.size();
  std::vector<unsigned> h(n * 256);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0x10000u + (unsigned)i;
  unsigned *src, *out, *da;
  hipMalloc(&src, 1 << 20); hipMalloc(&out, h.size() * 4); hipMalloc(&da, n * 4);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(da, addrs.data(), n * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 163840, 0, src, out, da, n);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  std::vector<unsigned> r(h.size());
  hipMemcpy(r.data(), out, r.size() * 4, hipMemcpyDeviceToHost);
  int bad_total = 0;
  for (int k = 0; k < n; ++k) {
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += r[k * 256 + i] != h[k * 256 + i];
    printf("LDS address %6u: %s (%d of 256 words wrong, first got 0x%x)\n", addrs[k], bad ? "WRONG" : "ok", bad, r[k * 256]);
    bad_total += bad;
  }
  return bad_total ? 2 : 0;
}
