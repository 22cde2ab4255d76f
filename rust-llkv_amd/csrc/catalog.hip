// catalog.hip — AOT instantiations of the fused scan kernel for gfx950 plus the small
// fixed-function kernels (octant fold, column statistics).
#include <algorithm>
#include <cstdlib>
#include "catalog.hpp"
#include "fused_scan.hip.h"
#include "select.hip.h"

#include <cstring>

namespace llkv {

template <class P> static hipError_t launch_plan(const ScanParams &p, hipStream_t stream) {
  if (p.n_tiles == 0) return hipSuccess;
  const uint32_t grid = P::ACC != 2 && p.scan_grid ? p.scan_grid : p.n_tiles; // scan_grid workgroups stream runs of tiles; 0: one tile per workgroup
  hipLaunchKernelGGL((fused_scan_kernel<P>), dim3(grid), dim3(kBlock), 0, stream, p);
  return hipGetLastError();
}

#define LLKV_CATALOG_ENTRY(STR, ...) {STR, &launch_plan<__VA_ARGS__>, __VA_ARGS__::LANES, __VA_ARGS__::U},

static const CatalogEntry kCatalog[] = {
#include "catalog_entries.inc"
    {nullptr, nullptr, 0, 0}};

const CatalogEntry *catalog_find(const char *type_string) {
  for (const CatalogEntry *e = kCatalog; e->type_string; ++e)
    if (std::strcmp(e->type_string, type_string) == 0) return e;
  return nullptr;
}
int catalog_size() { return (int)(sizeof(kCatalog) / sizeof(kCatalog[0])) - 1; }
const CatalogEntry *catalog_at(int i) { return i >= 0 && i < catalog_size() ? &kCatalog[i] : nullptr; }


hipError_t launch_fold_octants(const FoldParams &f, hipStream_t stream) {
  const uint32_t waves = kBlock / 64;
  hipLaunchKernelGGL(fold_octants_kernel, dim3(kOctants, (f.lanes + waves - 1) / waves), dim3(kBlock), 0, stream, f);
  return hipGetLastError();
}

hipError_t launch_exclusive_scan(const uint64_t *in, uint64_t *out, uint32_t n, hipStream_t stream) {
  hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, stream, in, out, n);
  return hipGetLastError();
}

// ---- column statistics -------------------------------------------------------------
template <class T> __global__ __launch_bounds__(256) void minmax_kernel(const T *v, uint64_t n, long long *out) {
  long long mn = 0x7FFFFFFFFFFFFFFFll, mx = (long long)0x8000000000000000ull;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const long long x = (long long)v[i];
    mn = x < mn ? x : mn;
    mx = x > mx ? x : mx;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    const long long omn = __shfl_xor(mn, off, 64), omx = __shfl_xor(mx, off, 64);
    mn = omn < mn ? omn : mn;
    mx = omx > mx ? omx : mx;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&out[0], mn);
    atomicMax(&out[1], mx);
  }
}

// Float column statistics over the FINITE values (NaN / ±∞ are skipped: they travel through the sums as such):
// out[0] = bits of the largest |v|, out[1] = bits of the smallest non-zero |v| (non-negative doubles order like their bits),
// out[2]: bit 0 when some value is NaN or ±∞, bit 1 when some value is −0.0.
template <class T> __global__ __launch_bounds__(256) void absrange_kernel(const T *v, uint64_t n, unsigned long long *out) {
  double hi = 0.0, lo = __builtin_inf();
  bool wild = false, negz = false;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const double raw = (double)v[i];
    const double x = __builtin_fabs(raw);
    const bool finite = x < __builtin_inf();
    wild |= !finite;
    negz |= (unsigned long long)__double_as_longlong(raw) == 0x8000000000000000ull; // −0.0: the f64 MIN / MAX keep ±0 ties in row order
    hi = (finite && x > hi) ? x : hi;
    lo = (finite && x > 0.0 && x < lo) ? x : lo;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    const double ohi = __shfl_xor(hi, off, 64), olo = __shfl_xor(lo, off, 64);
    hi = ohi > hi ? ohi : hi;
    lo = olo < lo ? olo : lo;
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&out[0], (unsigned long long)__double_as_longlong(hi));
    atomicMin(&out[1], (unsigned long long)__double_as_longlong(lo));
  }
  const unsigned long long flags = (__ballot(wild) != 0 ? 1ull : 0ull) | (__ballot(negz) != 0 ? 2ull : 0ull);
  if (flags && (threadIdx.x & 63) == 0) atomicOr(&out[2], flags);
}
hipError_t launch_absrange_f64(const double *values, uint64_t n, uint64_t *d_bits, hipStream_t stream) {
  hipLaunchKernelGGL((absrange_kernel<double>), dim3(1024), dim3(256), 0, stream, values, n, (unsigned long long *)d_bits);
  return hipGetLastError();
}
hipError_t launch_absrange_f32(const float *values, uint64_t n, uint64_t *d_bits, hipStream_t stream) {
  hipLaunchKernelGGL((absrange_kernel<float>), dim3(1024), dim3(256), 0, stream, values, n, (unsigned long long *)d_bits);
  return hipGetLastError();
}

// Padding rows between ragged chunks (every chunk starts on a 16-row boundary of the image) take a copy of the chunk's
// last value: no tile ever selects them, and the column statistics (min / max, float ranges) then describe the real
// rows only — zeros there widened a positive column's range down to 0, which cost plans their dense group ids.
__global__ __launch_bounds__(256) void fill_padding_kernel(char *col, uint32_t width, const uint64_t *pad /* [n][3]: first row, rows, source row */, uint32_t n) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * 16) return;
  const uint64_t *e = pad + (uint64_t)(i / 16) * 3;
  const uint32_t r = i % 16;
  if (r >= e[1]) return;
  for (uint32_t b = 0; b < width; ++b) col[(e[0] + r) * width + b] = col[e[2] * width + b];
}
hipError_t launch_fill_padding(void *col, uint32_t width, const uint64_t *d_pad, uint32_t n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(fill_padding_kernel, dim3((n * 16 + 255) / 256), dim3(256), 0, stream, (char *)col, width, d_pad, n);
  return hipGetLastError();
}

hipError_t launch_image_fold(const uint64_t *partials, uint64_t *exchange, const uint8_t *lane_ops, uint32_t n_wg, uint32_t ng, uint32_t k, uint32_t owned_mask,
                             uint32_t passes, uint32_t kl, const uint8_t *lane_src, const uint8_t *lane_xf, hipStream_t stream) {
  const uint32_t ngs = (ng + passes - 1) / passes;
  ImageFoldParams f{partials, exchange, lane_ops, n_wg, ng, k, owned_mask, passes, ngs, kl, lane_src, lane_xf};
  hipLaunchKernelGGL(image_fold_kernel, dim3((ng * k + 1 + kFoldCells - 1) / kFoldCells), dim3(256), 0, stream, f);
  return hipGetLastError();
}

hipError_t launch_part_reduce(const uint32_t *offsets, const uint64_t *records, const TileDesc *tiles, uint64_t *out, const uint8_t *lane_ops,
                              const uint8_t *lane_src, const uint8_t *lane_xf, uint32_t n_tiles, uint32_t np, uint32_t ngs, uint32_t ng, uint32_t kl, uint32_t k,
                              hipStream_t stream, uint32_t part0, uint32_t n_parts) {
  const size_t lds = (size_t)(kl - 1) * ngs * 8; // lanes 0 and 1 share a cell (part_reduce_kernel)
  static bool raised = false; // (the attribute belongs to the function, not to a launch)
  if (!raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(part_reduce_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    if (e != hipSuccess) return e;
    raised = true;
  }
  const char *cells = std::getenv("LLKV_HIP_PART_REDUCE_CELLS"); // 4 | 8 (measurement)
  const uint32_t deep = cells ? (std::atoi(cells) == 8 ? 1u : 0u) : (lds > (64u << 10) ? 1u : 0u);
  if (n_parts == 0) return hipSuccess;
  PartReduceParams f{offsets, records, tiles, out, lane_ops, lane_src, lane_xf, n_tiles, np, ngs, ng, kl, k, deep, part0};
  hipLaunchKernelGGL(part_reduce_kernel, dim3(n_parts), dim3(1024), lds, stream, f);
  return hipGetLastError();
}

// *flag |= 1 unless the real rows of the column (the tiles list them, in row order) are strictly ascending
template <class T> __global__ __launch_bounds__(256) void ascending_check_kernel(const T *v, const TileDesc *tiles, uint32_t n_tiles, uint32_t *flag) {
  const TileDesc td = tiles[blockIdx.x];
  bool bad = false;
  for (uint32_t r = threadIdx.x; r < td.rows; r += 256) {
    uint64_t left = td.dev_row + r - 1;
    if (r == 0) {
      if (blockIdx.x == 0) continue;
      const TileDesc before = tiles[blockIdx.x - 1]; // (no tile is empty)
      left = before.dev_row + before.rows - 1;
    }
    bad |= !(v[left] < v[td.dev_row + r]);
  }
  if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}
hipError_t launch_ascending_check(const void *values, uint32_t width, const TileDesc *tiles, uint32_t n_tiles, uint32_t *flag, hipStream_t stream) {
  if (n_tiles == 0) return hipSuccess;
  if (width == 8) hipLaunchKernelGGL((ascending_check_kernel<int64_t>), dim3(n_tiles), dim3(256), 0, stream, (const int64_t *)values, tiles, n_tiles, flag);
  else hipLaunchKernelGGL((ascending_check_kernel<int32_t>), dim3(n_tiles), dim3(256), 0, stream, (const int32_t *)values, tiles, n_tiles, flag);
  return hipGetLastError();
}

// Int64 → the 4-byte key image (engine.hpp: KeyImage); the caller has checked the statistics.  Two rows per thread: 16 B in, 8 B out.
__global__ __launch_bounds__(256) void narrow_i64_kernel(const int64_t *values, uint64_t n, int32_t *out) {
  const uint64_t pairs = n / 2, stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride) {
    const longlong2 v = reinterpret_cast<const longlong2 *>(values)[i];
    reinterpret_cast<int2 *>(out)[i] = make_int2((int)v.x, (int)v.y);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[n - 1] = (int32_t)values[n - 1];
}
hipError_t launch_narrow_i64(const int64_t *values, uint64_t n, int32_t *out, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((n / 2 + 255) / 256 + 1, 256 * 16);
  hipLaunchKernelGGL(narrow_i64_kernel, dim3(grid), dim3(256), 0, stream, values, n, out);
  return hipGetLastError();
}

hipError_t launch_minmax_i64(const int64_t *values, uint64_t n, int64_t *d_minmax, hipStream_t stream) {
  hipLaunchKernelGGL((minmax_kernel<int64_t>), dim3(1024), dim3(256), 0, stream, values, n, (long long *)d_minmax);
  return hipGetLastError();
}
hipError_t launch_minmax_i32(const int32_t *values, uint64_t n, int64_t *d_minmax, hipStream_t stream) {
  hipLaunchKernelGGL((minmax_kernel<int32_t>), dim3(1024), dim3(256), 0, stream, values, n, (long long *)d_minmax);
  return hipGetLastError();
}

} // namespace llkv
