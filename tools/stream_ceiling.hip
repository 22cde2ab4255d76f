// stream_ceiling.hip — what a pure streaming read achieves on this device, as the practical ceiling beside
// the 8 TB/s datasheet peak that bench.py's roofline.frac is quoted against.  Reads `bytes` of HBM once per
// launch with the same access shape as fused_scan_kernel (256-thread workgroups, dwordx4 non-temporal loads,
// U loads in flight per thread), folds them with XOR and writes one word per workgroup.
//   hipcc --offload-arch=gfx950 -O3 -o tools/stream_ceiling tools/stream_ceiling.hip
//   tools/stream_ceiling [bytes] [reps]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int U, bool NT, int STREAMS>
__global__ __launch_bounds__(256) void read_kernel(const v4u *__restrict__ src, size_t n_vec_per_stream, size_t tile_vecs, uint32_t *out) {
  // STREAMS equally long arrays read in lockstep (Q1 reads 5 value columns + 2 code columns)
  const size_t t0 = (size_t)blockIdx.x * tile_vecs;
  v4u acc = {0, 0, 0, 0};
  for (size_t i = threadIdx.x; i < tile_vecs; i += 256 * U) {
    v4u v[STREAMS][U];
#pragma unroll
    for (int s = 0; s < STREAMS; ++s)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const size_t k = t0 + i + (size_t)u * 256;
        const v4u *p = src + (size_t)s * n_vec_per_stream + (k < n_vec_per_stream ? k : 0);
        v[s][u] = NT ? __builtin_nontemporal_load(p) : *p;
      }
#pragma unroll
    for (int s = 0; s < STREAMS; ++s)
#pragma unroll
      for (int u = 0; u < U; ++u) acc ^= v[s][u];
  }
  uint32_t x = acc.x ^ acc.y ^ acc.z ^ acc.w;
  for (int o = 32; o; o >>= 1) x ^= __shfl_xor(x, o);
  if ((threadIdx.x & 63) == 0) atomicXor(&out[blockIdx.x], x);
}

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); return 1; } } while (0)

template <int U, bool NT, int STREAMS>
static int run(const char *name, const v4u *src, size_t bytes, size_t tile_bytes, uint32_t *out, int reps) {
  const size_t n_vec = bytes / 16 / STREAMS, tile_vecs = tile_bytes / 16;
  const unsigned grid = (unsigned)((n_vec + tile_vecs - 1) / tile_vecs);
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  std::vector<float> ms;
  for (int r = 0; r < reps + 3; ++r) {
    CHECK(hipEventRecord(a));
    read_kernel<U, NT, STREAMS><<<grid, 256>>>(src, n_vec, tile_vecs, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float t;
    CHECK(hipEventElapsedTime(&t, a, b));
    if (r >= 3) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const double med = ms[ms.size() / 2], best = ms[0];
  const double gb = (double)(n_vec * 16 * STREAMS) / 1e9;
  printf("%-34s tile %7zu B grid %7u  median %.4f ms = %7.1f GB/s   best %.4f ms = %7.1f GB/s\n", name, tile_bytes, grid, med, gb / med * 1e3, best, gb / best * 1e3);
  return 0;
}

int main(int argc, char **argv) {
  const size_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 10) : 2279469976ull; // Q1 SF10: 59 986 052 rows × 38 B
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  void *src;
  uint32_t *out;
  CHECK(hipMalloc(&src, bytes + (1 << 20)));
  CHECK(hipMemset(src, 1, bytes + (1 << 20)));
  CHECK(hipMalloc(&out, 1 << 24));
  CHECK(hipMemset(out, 0, 1 << 24));
  const v4u *s = (const v4u *)src;
  printf("streaming read of %zu bytes (%d timed launches each)\n", bytes, reps);
  for (size_t tile : {32768ul, 65536ul, 131072ul, 262144ul, 1048576ul}) {
    if (run<4, true, 1>("1 stream, U=4, non-temporal", s, bytes, tile, out, reps)) return 1;
  }
  if (run<8, true, 1>("1 stream, U=8, non-temporal", s, bytes, 131072, out, reps)) return 1;
  if (run<2, true, 1>("1 stream, U=2, non-temporal", s, bytes, 131072, out, reps)) return 1;
  if (run<4, false, 1>("1 stream, U=4, temporal", s, bytes, 131072, out, reps)) return 1;
  if (run<2, true, 5>("5 streams, U=2, non-temporal", s, bytes, 131072, out, reps)) return 1;
  if (run<1, true, 5>("5 streams, U=1, non-temporal", s, bytes, 131072, out, reps)) return 1;
  if (run<4, true, 4>("4 streams, U=4 (Q6-like)", s, std::min<size_t>(bytes, 1679609456ull), 131072, out, reps)) return 1; // never past the allocation
  if (run<4, true, 4>("4 streams, U=4, 16 KiB tiles", s, std::min<size_t>(bytes, 1679609456ull), 16384, out, reps)) return 1;
  CHECK(hipFree(src)); CHECK(hipFree(out));
  return 0;
}
