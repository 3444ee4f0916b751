// How fast does ONE CU take in a 128 KiB tile that the 32 workgroups of its XCD all read (the width-1024 backward scan's
// situation: lstm_scan_w32.hip)?  256 workgroups of 512 threads, one per CU; workgroup b reads tile (b % 8) of a ring of
// tiles, REP times, and reports clock64 cycles per tile.
//   mode 0: LDS-DMA, waves 4-7, 32 requests of 1 KiB each (row-contiguous), vmcnt(0) per tile
//   mode 1: the same by all 8 waves (16 requests each)
//   mode 2: register loads (16 x 16 bytes per thread), no LDS
//   mode 3: register loads + ds_write_b128 of every piece
//   mode 4: as 0 but TWO tiles' requests in flight (second tile into the same LDS: throughput only)
// aux: 0 plain, 16 sc1, 2 nt
// build: hipcc --offload-arch=gfx950 -O3 -o tile_ingest tile_ingest.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void lds_void_t;

template <int AUX>
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  if (AUX == 16)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen sc1 lds\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_addr) : "memory");
  else if (AUX == 2)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_addr) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_addr) : "memory");
}

template <int MODE, int AUX>
__global__ __launch_bounds__(512, 1) void ingest(const unsigned* src, unsigned long long* cycles, unsigned* sink, int rep, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned base = (unsigned)(size_t)(lds_void_t*)smem;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, (int)((unsigned)n_tiles * 131072u), 0x00020000);
  unsigned acc = 0;
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int r = 0; r < rep; ++r) {
    const unsigned tile = (unsigned)(((blockIdx.x & 7) + 8 * (r % (n_tiles / 8))) * 131072);
    if (MODE == 0 || MODE == 4) {
      if (wave >= 4) {
#pragma unroll
        for (int j = 0; j < 32; ++j) dma16<AUX>(rs, tile + (unsigned)(((wave - 4) * 32 + j) * 1024 + lane * 16), base + ((wave - 4) * 32 + j) * 1024);
        if (MODE == 4) {
          asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int j = 0; j < 16; ++j) dma16<AUX>(rs, tile + (unsigned)((wave * 16 + j) * 1024 + lane * 16), base + (wave * 16 + j) * 1024);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      u32x4 v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(tile + (unsigned)((wave * 16 + j) * 1024 + lane * 16)), 0, AUX);
      if (MODE == 3) {
#pragma unroll
        for (int j = 0; j < 16; ++j) *reinterpret_cast<u32x4*>(smem + (wave * 16 + j) * 1024 + lane * 16) = v[j];
      } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc ^= v[j].x ^ v[j].w;
      }
    }
    __syncthreads();
    if (MODE != 2) acc ^= *reinterpret_cast<const unsigned*>(smem + ((tid * 260) & 131071 & ~3));
    __syncthreads();
  }
  const unsigned long long t1 = clock64();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int AUX>
void run(const char* name, const unsigned* src, unsigned long long* cyc, unsigned* sink, int rep, int n_tiles) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&ingest<MODE, AUX>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((ingest<MODE, AUX>), dim3(256), dim3(512), 131072, 0, src, cyc, sink, 8, n_tiles);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((ingest<MODE, AUX>), dim3(256), dim3(512), 131072, 0, src, cyc, sink, rep, n_tiles);
  hipEventRecord(e1, 0);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return; }
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto c : h) s += (double)c;
  printf("%-44s %8.0f cycles per tile (clock64), %6.2f us per tile, %6.1f GB/s per CU, %5.2f TB/s over the chip\n", name, s / 256 / rep, ms * 1e3 / rep,
         131072.0 / (ms * 1e-3 / rep) / 1e9, 256 * 131072.0 / (ms * 1e-3 / rep) / 1e12);
}

int main() {
  const int n_tiles = 64, rep = 2000;
  unsigned* src; unsigned long long* cyc; unsigned* sink;
  hipMalloc(&src, (size_t)n_tiles * 131072); hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 64);
  hipMemset(src, 1, (size_t)n_tiles * 131072);
  run<0, 16>("LDS-DMA, 4 waves, sc1", src, cyc, sink, rep, n_tiles);
  run<0, 0>("LDS-DMA, 4 waves, plain", src, cyc, sink, rep, n_tiles);
  run<0, 2>("LDS-DMA, 4 waves, nt", src, cyc, sink, rep, n_tiles);
  run<1, 16>("LDS-DMA, 8 waves, sc1", src, cyc, sink, rep, n_tiles);
  run<1, 0>("LDS-DMA, 8 waves, plain", src, cyc, sink, rep, n_tiles);
  run<4, 16>("LDS-DMA, 4 waves, sc1, two tiles in flight", src, cyc, sink, rep, n_tiles);
  run<2, 16>("register loads, sc1, no LDS", src, cyc, sink, rep, n_tiles);
  run<2, 0>("register loads, plain, no LDS", src, cyc, sink, rep, n_tiles);
  run<3, 16>("register loads + ds_write_b128, sc1", src, cyc, sink, rep, n_tiles);
  run<3, 0>("register loads + ds_write_b128, plain", src, cyc, sink, rep, n_tiles);
  return 0;
}
