// How many ds_add_f32 (no return) instructions per clock does a CU's LDS sustain when every lane adds to a different word of a
// randomly chosen 64-word row (a scatter-add of gradient rows into per-character buckets)?  And with returning atomics / plain
// read-modify-write for comparison.
// build: hipcc --offload-arch=gfx950 -O3 -o lds_atomic_rate lds_atomic_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE, int NT>
__global__ __launch_bounds__(NT, 1) void k(const unsigned* rows, unsigned long long* cycles, float* sink, int n, int n_rows) {
  extern __shared__ float buckets[];      // [n_rows][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < n_rows * 64; i += NT) buckets[i] = 0.f;
  __syncthreads();
  const unsigned* my = rows + ((size_t)blockIdx.x * (NT / 64) + wave) * n;
  const unsigned long long t0 = clock64();
  float v = 1.f + lane;
  for (int i = 0; i < n; i += 4) {
    unsigned r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = __builtin_amdgcn_readfirstlane(my[i + j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float* p = buckets + r[j] * 64 + lane;
      if (MODE == 0) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // result unused: ds_add_f32
      else *p += v;                                                                                    // plain RMW (racy across waves: rate only)
    }
  }
  __syncthreads();
  const unsigned long long t1 = clock64();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
  if (buckets[tid] == -1.f) sink[0] = 1.f;
}

template <int MODE, int NT>
void run(const char* name, const unsigned* rows, unsigned long long* cyc, float* sink, int n, int n_rows) {
  const size_t lds = (size_t)n_rows * 64 * 4;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), lds, 0, rows, cyc, sink, n, n_rows);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: failed\n", name); return; }
  std::vector<unsigned long long> h(256);
  hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto c : h) s += (double)c;
  const double per_cu = s / 256, instrs = (double)n * (NT / 64);
  printf("%-46s %d waves: %8.0f cycles for %8.0f wave-instructions per CU = %5.2f cycles per instruction\n", name, NT / 64, per_cu, instrs, per_cu / instrs);
}

int main() {
  const int n = 16384, n_rows = 456;
  std::vector<unsigned> h((size_t)256 * 16 * n);
  unsigned s = 12345u;
  for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (s >> 8) % n_rows; }
  unsigned* rows; unsigned long long* cyc; float* sink;
  hipMalloc(&rows, h.size() * 4); hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 64);
  hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  run<0, 512>("ds_add_f32, random row of 456, lane = word", rows, cyc, sink, n, n_rows);
  run<0, 1024>("ds_add_f32, random row of 456, lane = word", rows, cyc, sink, n, n_rows);
  run<1, 512>("plain read-modify-write", rows, cyc, sink, n, n_rows);
  run<1, 1024>("plain read-modify-write", rows, cyc, sink, n, n_rows);
  return 0;
}
