// Stand-alone measurement (hipcc, no library): rocPRIM's radix_sort_pairs over (u64 key, u32 position) pairs — the sort of the
// sort-based GROUP BY (csrc/group_sort.cpp → join.hip: hj_sort_u64_u32_bits) — by key bits, against what its passes must move.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/sort_pairs.hip -o /tmp/sort_pairs && /tmp/sort_pairs [n = 59986052]
// One LSD pass over 8 key bits reads and writes every pair once: 2 · 12 B per element and pass (+ the histogram's read of the keys).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill_keys(uint64_t *k, uint32_t *v, uint64_t n, uint32_t bits) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x = i * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull; // splitmix-style scramble
  x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
  k[i] = bits >= 64 ? x : x & ((1ull << bits) - 1);
  v[i] = (uint32_t)i;
}

int main(int argc, char **argv) {
  const uint64_t n = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 59986052ull;
  uint64_t *k0, *k1; uint32_t *v0, *v1;
  CHECK(hipMalloc(&k0, n * 8)); CHECK(hipMalloc(&k1, n * 8)); CHECK(hipMalloc(&v0, n * 4)); CHECK(hipMalloc(&v1, n * 4));
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (uint32_t bits : {8u, 16u, 24u, 28u, 32u, 64u}) {
    size_t tb = 0;
    CHECK(rocprim::radix_sort_pairs(nullptr, tb, k0, k1, v0, v1, (size_t)n, 0u, bits, 0));
    void *tmp; CHECK(hipMalloc(&tmp, tb ? tb : 8));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipLaunchKernelGGL(fill_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, k0, v0, n, bits);
      CHECK(hipEventRecord(a, 0));
      CHECK(rocprim::radix_sort_pairs(tmp, tb, k0, k1, v0, v1, (size_t)n, 0u, bits, 0));
      CHECK(hipEventRecord(b, 0));
      CHECK(hipEventSynchronize(b));
      float ms; CHECK(hipEventElapsedTime(&ms, a, b));
      if (rep && ms < best) best = ms; // (the first repetition warms up)
    }
    const double passes = (bits + 7) / 8; // what an 8-bit LSD sort needs
    const double moved = (double)n * 12.0 * 2.0 * passes;
    std::printf("bits %2u  %8.3f ms   %4.0f passes of 8 bits would move %6.2f GB -> %6.0f GB/s of that  (temp %zu MB)\n", bits, best, passes, moved / 1e9, moved / (best * 1e-3) / 1e9, tb >> 20);
    CHECK(hipFree(tmp));
  }
  return 0;
}
