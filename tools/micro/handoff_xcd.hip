// Microbenchmark: round-trip time of a two-workgroup ping-pong through global memory for different
// cache policies of the store and of the polling load, for partners on the SAME XCD (block ids 8 apart)
// and on different XCDs.  Prints the XCC id of every block first (is block b on XCD b % 8 ?).
//   hipcc --offload-arch=gfx950 -O3 -o handoff_xcd tools/micro/handoff_xcd.hip && ./handoff_xcd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

template <int LD, int ST>
__device__ __forceinline__ unsigned ld(const unsigned* p) {
  unsigned v;
  if (LD == 0) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == 4) asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <int ST>
__device__ __forceinline__ void st(unsigned* p, unsigned v) {
  if (ST == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
  if (ST == 1) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  if (ST == 2) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (ST == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}

// blocks a and b play; everybody else exits.  word[0]: a -> b, word[32]: b -> a (different cache lines)
template <int LD, int ST>
__global__ void pingpong(unsigned* words, int a, int b, int iters, unsigned long long* cycles, unsigned* xcc, int* failed) {
  if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id();
  if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
  if (threadIdx.x != 0) return;
  const bool first = (int)blockIdx.x == a;
  unsigned* mine = words + (first ? 0 : 32);
  const unsigned* theirs = words + (first ? 32 : 0);
  const unsigned long long t0 = wall_clock64();
  for (int k = 1; k <= iters; ++k) {
    if (first) st<ST>(mine, (unsigned)k);
    unsigned spin = 0;
    while (ld<LD, ST>(theirs) != (unsigned)k) {
      if (++spin > (1u << 22)) { *failed = 1; return; }
    }
    if (!first) st<ST>(mine, (unsigned)k);
  }
  if (first) *cycles = wall_clock64() - t0;
}

template <int LD, int ST>
void run(const char* name, int a, int b, unsigned* words, unsigned long long* cyc, unsigned* xcc, int* failed) {
  const int iters = 2000;
  CHECK(hipMemset(words, 0, 4096));
  CHECK(hipMemset(failed, 0, 4));
  hipLaunchKernelGGL((pingpong<LD, ST>), dim3(256), dim3(64), 0, 0, words, a, b, iters, cyc, xcc, failed);
  CHECK(hipDeviceSynchronize());
  unsigned long long c; int f; unsigned x[256];
  CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(x, xcc, sizeof(x), hipMemcpyDeviceToHost));
  if (f) printf("%-34s blocks %3d(xcc %u) <-> %3d(xcc %u): NEVER SAW THE VALUE (not coherent)\n", name, a, x[a], b, x[b]);
  else printf("%-34s blocks %3d(xcc %u) <-> %3d(xcc %u): %.0f ns per one-way hand-off\n", name, a, x[a], b, x[b], c * 10.0 / iters / 2);   // wall clock: 100 MHz
}

int main() {
  unsigned *words, *xcc; unsigned long long* cyc; int* failed;
  CHECK(hipMalloc(&words, 4096)); CHECK(hipMalloc(&xcc, 1024)); CHECK(hipMalloc(&cyc, 8)); CHECK(hipMalloc(&failed, 4));
  run<2, 2>("warm-up", 0, 8, words, cyc, xcc, failed);
  unsigned x[256];
  CHECK(hipMemcpy(x, xcc, sizeof(x), hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 256; ++i) bad += (x[i] != (unsigned)(i % 8));
  printf("XCC id of block b == b %% 8 for %d of 256 blocks; first 16:", 256 - bad);
  for (int i = 0; i < 16; ++i) printf(" %u", x[i]);
  printf("\n");
  for (int pass = 0; pass < 2; ++pass) {
    const int a = 0, b = pass == 0 ? 8 : 1;
    printf("---- partners on %s\n", pass == 0 ? "the same XCD" : "different XCDs");
    run<2, 2>("load sc1, store sc1", a, b, words, cyc, xcc, failed);
    run<3, 3>("load sc0 sc1, store sc0 sc1", a, b, words, cyc, xcc, failed);
    run<1, 1>("load sc0, store sc0", a, b, words, cyc, xcc, failed);
    run<1, 0>("load sc0, store plain", a, b, words, cyc, xcc, failed);
    run<2, 0>("load sc1, store plain", a, b, words, cyc, xcc, failed);
    run<1, 2>("load sc0, store sc1", a, b, words, cyc, xcc, failed);
    run<4, 0>("load nt, store plain", a, b, words, cyc, xcc, failed);
    run<0, 0>("load plain, store plain", a, b, words, cyc, xcc, failed);
  }
  return 0;
}
