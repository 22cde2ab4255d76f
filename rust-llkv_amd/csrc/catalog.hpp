// catalog.hpp — ahead-of-time compiled instantiations of fused_scan_kernel, keyed by the
// plan type string produced by lower_plan() (plan.hpp).  Plans outside the catalog are
// compiled at run time from the same header (jit.cpp).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace llkv {

struct ScanParams;
struct FoldParams;

using ScanLauncher = hipError_t (*)(const ScanParams &p, hipStream_t stream);

struct CatalogEntry {
  const char *type_string;
  ScanLauncher launch;
  int lanes;
  int unroll;
};

const CatalogEntry *catalog_find(const char *type_string);
int catalog_size();
const CatalogEntry *catalog_at(int i);

hipError_t launch_fold_octants(const FoldParams &f, hipStream_t stream);

// Exclusive scan of n uint64 counts (out has n + 1 entries, out[n] = total); one block.
hipError_t launch_exclusive_scan(const uint64_t *in, uint64_t *out, uint32_t n, hipStream_t stream);

// Column statistics (min/max of an integer column already resident in HBM).
hipError_t launch_minmax_i64(const int64_t *values, uint64_t n, int64_t *d_minmax /*[2]*/, hipStream_t stream);
hipError_t launch_minmax_i32(const int32_t *values, uint64_t n, int64_t *d_minmax /*[2]*/, hipStream_t stream);

} // namespace llkv
