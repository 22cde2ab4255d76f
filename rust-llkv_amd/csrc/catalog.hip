// catalog.hip — AOT instantiations of the fused scan kernel for gfx950 plus the small
// fixed-function kernels (octant fold, column statistics).
#include "catalog.hpp"
#include "fused_scan.hip.h"
#include "select.hip.h"

#include <cstring>

namespace llkv {

template <class P> static hipError_t launch_plan(const ScanParams &p, hipStream_t stream) {
  if (p.n_tiles == 0) return hipSuccess;
  const uint32_t tpw = P::ACC == 1 && p.tiles_per_wg ? p.tiles_per_wg : 1u; // register plans: one tile per workgroup
  hipLaunchKernelGGL((fused_scan_kernel<P>), dim3((p.n_tiles + tpw - 1) / tpw), dim3(kBlock), 0, stream, p);
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

hipError_t launch_minmax_i64(const int64_t *values, uint64_t n, int64_t *d_minmax, hipStream_t stream) {
  hipLaunchKernelGGL((minmax_kernel<int64_t>), dim3(1024), dim3(256), 0, stream, values, n, (long long *)d_minmax);
  return hipGetLastError();
}
hipError_t launch_minmax_i32(const int32_t *values, uint64_t n, int64_t *d_minmax, hipStream_t stream) {
  hipLaunchKernelGGL((minmax_kernel<int32_t>), dim3(1024), dim3(256), 0, stream, values, n, (long long *)d_minmax);
  return hipGetLastError();
}

} // namespace llkv
