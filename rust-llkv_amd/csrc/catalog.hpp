// catalog.hpp — ahead-of-time compiled instantiations of fused_scan_kernel, keyed by the
// plan type string produced by lower_plan() (plan.hpp).  Plans outside the catalog are
// compiled at run time from the same header (jit.cpp).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace llkv {

struct ScanParams;
struct FoldParams;
struct TileDesc;

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
hipError_t launch_ascending_check(const void *values, uint32_t width /*4 or 8, signed*/, const TileDesc *tiles, uint32_t n_tiles, uint32_t *flag, hipStream_t stream);
hipError_t launch_narrow_i64(const int64_t *values, uint64_t n, int32_t *out, hipStream_t stream); // Int64 → the 4-byte key image
hipError_t launch_minmax_i64(const int64_t *values, uint64_t n, int64_t *d_minmax /*[2]*/, hipStream_t stream);
hipError_t launch_minmax_i32(const int32_t *values, uint64_t n, int64_t *d_minmax /*[2]*/, hipStream_t stream);

// Float column statistics over the finite values: d_bits[0] = bits of the largest |v| (initialise to 0), d_bits[1] =
// bits of the smallest non-zero |v| (initialise to the bits of +∞).
hipError_t launch_absrange_f64(const double *values, uint64_t n, uint64_t *d_bits, hipStream_t stream);
hipError_t launch_absrange_f32(const float *values, uint64_t n, uint64_t *d_bits, hipStream_t stream);

// Copies a chunk's last value into the ≤ 15 padding rows behind it; d_pad = n × {first padding row, rows, source row}.
hipError_t launch_fill_padding(void *col, uint32_t width, const uint64_t *d_pad, uint32_t n, hipStream_t stream);

// Shared-image GROUP BY: workgroup images [pass][n_wg][k·ngs + 1] (lane-major, ngs = ⌈ng / passes⌉ groups per slice)
// → exchange image [8][ng·k + 1] (group-major).
hipError_t launch_image_fold(const uint64_t *partials, uint64_t *exchange, const uint8_t *lane_ops, uint32_t n_wg, uint32_t ng, uint32_t k, uint32_t owned_mask,
                             uint32_t passes, uint32_t kl, const uint8_t *lane_src, const uint8_t *lane_xf, hipStream_t stream);

// Partitioned GROUP BY: one workgroup per partition reduces its records (fused_scan.hip.h: part_scatter_body wrote them) in an LDS
// image of (kl − 1) × ngs cells and writes the rows [group][k] of its groups; partitions [part0, part0 + n_parts) of the np.
hipError_t launch_part_reduce(const uint32_t *offsets, const uint64_t *records, const TileDesc *tiles, uint64_t *out, const uint8_t *lane_ops,
                              const uint8_t *lane_src, const uint8_t *lane_xf, uint32_t n_tiles, uint32_t np, uint32_t ngs, uint32_t ng, uint32_t kl, uint32_t k,
                              hipStream_t stream, uint32_t part0, uint32_t n_parts);

} // namespace llkv
