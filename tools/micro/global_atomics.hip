// Micro-benchmark: what a GROUP BY with millions of groups would pay for accumulating straight into an image in HBM —
// n rows, k non-returning 64-bit atomic adds each into image[g·k + lane], g random in [0, ng).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/global_atomics tools/micro/global_atomics.hip && /tmp/global_atomics
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int K>
__global__ __launch_bounds__(256) void accumulate(const uint32_t *g, const int64_t *v, uint64_t n, unsigned long long *image) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint32_t grp = g[i];
    const int64_t x = v[i];
#pragma unroll
    for (int l = 0; l < K; ++l) __hip_atomic_fetch_add(&image[(uint64_t)grp * K + l], (unsigned long long)(x + l), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void fill_groups(uint32_t *g, int64_t *v, uint64_t n, uint32_t ng) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t z = i + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  g[i] = (uint32_t)(z % ng);
  v[i] = (int64_t)(z >> 40);
}

template <int K>
int run(const uint32_t *g, const int64_t *v, uint64_t n, uint32_t ng, unsigned long long *image, int grid) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e9f;
  for (int it = 0; it < 5; ++it) {
    CK(hipMemsetAsync(image, 0, (size_t)ng * K * 8, nullptr));
    CK(hipEventRecord(a, nullptr));
    hipLaunchKernelGGL((accumulate<K>), dim3(grid), dim3(256), 0, nullptr, g, v, n, image);
    CK(hipEventRecord(b, nullptr));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  std::printf("ng=%u k=%d grid=%d: %.3f ms  (%.1f G atomics/s, image %.0f MB)\n", ng, K, grid, best, (double)n * K / best / 1e6, (double)ng * K * 8 / 1e6);
  return 0;
}

int main() {
  const uint64_t n = 59986052;
  uint32_t *g; int64_t *v; unsigned long long *image;
  CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&v, n * 8)); CK(hipMalloc(&image, (size_t)1 << 30));
  for (uint32_t ng : {4096u, 200000u, 2000000u, 15000000u}) {
    hipLaunchKernelGGL(fill_groups, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, nullptr, g, v, n, ng);
    CK(hipDeviceSynchronize());
    for (int grid : {1024, 4096}) {
      if (run<1>(g, v, n, ng, image, grid)) return 1;
      if (run<3>(g, v, n, ng, image, grid)) return 1;
      if (run<7>(g, v, n, ng, image, grid)) return 1;
    }
  }
  return 0;
}
