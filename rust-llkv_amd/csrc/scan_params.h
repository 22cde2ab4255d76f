// scan_params.h — plain-old-data shared by host and device code of the fused scan path
// (kernel arguments, tile descriptors, lane algebra constants).  No device code here.
#pragma once

#ifndef __HIPCC_RTC__
#include <stdint.h>
#else
// hiprtc has no <stdint.h>; same widths as the host definitions
typedef unsigned long long uint64_t;
typedef long long int64_t;
typedef unsigned int uint32_t;
typedef int int32_t;
typedef unsigned short uint16_t;
typedef short int16_t;
typedef unsigned char uint8_t;
#endif

namespace llkv {

// Lane algebra: every partial state is a vector of independent 64-bit lanes.
enum LaneOp : int { OP_ADD_F64 = 0, OP_ADD_I64 = 1, OP_MIN_I64 = 2, OP_MAX_I64 = 3, OP_MAX_U64 = 4 };

constexpr int kBlock = 256;
constexpr int kRowsPerThread = 2;
constexpr int kStepRows = kBlock * kRowsPerThread; // 512 rows per block step
constexpr int kMaxCols = 16; // value slots + validity-mask slots of one plan
constexpr int kMaxLits = 48; // literal slots per bank (the MVCC leaf keeps the snapshot's non-committed transaction ids here)
constexpr int kMaxKeys = 4;
constexpr int kOctants = 8; // canonical partition of the chunk list

#ifdef __HIPCC__
#define LLKV_HOST_DEVICE __host__ __device__
#else
#define LLKV_HOST_DEVICE
#endif
// The one hash of every open-addressing table (64-bit finalizer of MurmurHash3): build kernels (join.hip) and the
// run-time compiled probes (select.hip.h) must agree on it.
inline LLKV_HOST_DEVICE uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

struct TileDesc {
  uint64_t dev_row;     // first row of the tile in the device column image
  uint64_t logical_row; // row id of that row (dense ids)
  uint32_t rows;
  uint32_t octant;
};

// Selection-style kernels (select / probe-emit / value-emit) report a checked-arithmetic error inside the
// predicate (Expr::Compare over computed sides) by adding this to the (tile, wave) count of the counting
// pass; real counts stay far below it, so `total >= kPredErrorBit` on the host means "error".
constexpr uint64_t kPredErrorBit = 1ull << 40;

struct ScanParams {
  const void *col[kMaxCols];
  const TileDesc *tiles;
  uint64_t *tile_partials; // [lanes][n_tiles]
  int64_t lit_i[kMaxLits];
  double lit_f[kMaxLits];
  uint32_t key_stride[kMaxKeys];
  uint32_t n_tiles;
  uint32_t sub_rows;          // selection kernels: rows per wave sub-tile (tile_rows / 4)
  // fused scan, LDS-accumulator plans: launched with `scan_grid` workgroups (0 = one per tile); workgroup b streams
  // the consecutive tiles [b·n/g, (b+1)·n/g) and publishes one partial per (tile, wave), so the reduction association
  // is a property of the canonical tile list, never of the launch geometry
  uint32_t scan_grid;
  uint32_t group_base;        // shared-image plans cut into passes: first group of this launch's slice
  const uint64_t *aux_in;     // selection: exclusive offsets per (tile, wave)
  uint64_t *aux_out;          // selection: logical row ids out
  uint64_t *aux_out2;         // selection: device row indices out
  // probe-emit kernels (join → aggregate pipelines): open-addressing table of the build side
  const unsigned long long *ht_owner; // slot → owning build device row (~0 = empty)
  uint64_t ht_mask;
  const void *ht_keys;        // build key column image
  uint32_t ht_key_width;      // 4 or 8
  uint32_t ht_key_signed;
  uint32_t *aux_out32;        // probe-emit: matching slot per emitted row
  // piggy-back fold: the first workgroups of this launch fold the tile partials the PREVIOUS launch of
  // the query left behind (visible through the kernel boundary — no atomics, no fences) while the rest of
  // the grid already streams; the standalone fold_octants_kernel flushes the last execution.
  const uint64_t *prev_partials; // [lanes][n_tiles] of the previous execution, or nullptr
  uint64_t *prev_exchange;       // [kOctants][lanes] image to fold it into
  uint32_t octant_tile_begin[kOctants + 1];
  uint32_t owned_mask;           // octants of this rank; rows of the others are written as zero
  // probe-emit over a statistics-bounded build key: direct addressing instead of hashing.  Bit (k − bm_min) of
  // bm_bits says whether key k is on the build side; its rank among the set bits (bm_prefix = set bits before
  // the word) indexes bm_group, the group id of that key.  A fact table clustered by the key walks these arrays
  // almost sequentially.
  const uint64_t *bm_bits;       // nullptr: hash table (ht_*)
  const uint32_t *bm_prefix;
  // bm_base != nullptr: bm_prefix[w] counts the set bits before word w inside its chunk of 2^bm_chunk_shift words only, and
  // bm_base[w >> bm_chunk_shift] the set bits before the chunk (hj_launch_rank_words: ONE launch ranks the bitmap — chunk-
  // local prefixes need nothing from other workgroups, the last one to finish scans the few chunk totals)
  const uint32_t *bm_base;
  uint32_t bm_chunk_shift;
  const uint32_t *bm_group;
  const uint32_t *bm_unsorted;   // optional; *bm_unsorted == 0: the build list is in key order, the rank is the group id
  int64_t bm_min;
  uint64_t bm_span;              // max − min
  // numeric image of dictionary-coded Utf8 aggregate inputs (SQLite-style coercion, llkv-aggregate/src/lib.rs:400-449):
  // dict_num[slot · 256 + code] = the string parsed as f64, 0.0 when it is not a number
  const double *dict_num;
  // partitioned GROUP BY (group_part.cpp; fused_scan.hip.h: part_scatter_body): every tile of 32 768 rows is put in
  // partition order (a partition = 2^part_shift consecutive group ids), and every partition is then reduced in an LDS
  // image of its own
  uint32_t *part_hist;          // [tile][part_np + 1]: record position where the cell of (tile, partition) starts; [part_np]: where the tile's records end
  const uint32_t *part_offsets; // (unused)
  uint64_t *part_val;           // record r = K words at part_val[r·K]: [0] group id within the partition (lane 0 counts rows), [l] kernel lane l
  uint32_t part_shift;
  uint32_t part_np;             // partitions (≤ kMaxParts)
  uint32_t *part_err;           // arithmetic / predicate error codes, OR-ed
  // key-bits scans (select.hip.h: keybits_body): the OUTPUT bitmap covers keys kb_min … kb_min + kb_span (bm_* describe the
  // set an InKeySet conjunct of the predicate tests).  kb_ranged: the emitted key column is in ascending row order and only
  // keys in [kb_lo, kb_hi] are wanted — a tile whose first key is above or whose last key is below leaves at once (a rank
  // of a fact table clustered by that key needs the dimension rows of its own key range only).
  int64_t kb_min;
  uint64_t kb_span;
  int64_t kb_lo, kb_hi;
  uint32_t kb_ranged;
  // probe-emit, piggy-backed: the launch also zeroes words [0, *zero_n) of zero_k arrays that lie zero_stride words apart
  // (the per-group state the run sums add into, sized by the number of set bits the rank scan left in *zero_n) — the
  // kernel boundary makes them visible to the next launch, and no host round trip has to learn the group count first
  uint32_t zero_k;
  uint64_t *zero_words;
  uint64_t zero_stride;
  const uint32_t *zero_n;
  // … and, when the probe emits key-bit positions (it needs no rank itself), the word ranks of the dimension bitmap: the first
  // rk_chunks (≤ kBlock) workgroups rank one chunk of 2^rk_shift words each before they turn to their tile, the last of them to
  // finish scans the chunk totals (rk_base[rk_chunks] = the number of groups) and publishes count + 1 in rk_state[1].  The group
  // state is then zeroed in slices by the rk_helpers workgroups from rk_help_first on, at their END (select.hip.h: probe_zero_slices)
  // — instead of by share at the start.  rk_state: 4 zeroed words ([3]: a helper gave up waiting for the count).
  const uint64_t *rk_bits;
  uint64_t rk_words;
  uint32_t rk_shift, rk_chunks;
  uint32_t *rk_prefix, *rk_base, *rk_state;
  uint32_t rk_help_first, rk_helpers;
};

constexpr int kMaxOuts = 8;
// Arguments of project_kernel (gather + computed projections over a window of selected rows).
struct ProjParams {
  const void *col[kMaxCols];
  const uint64_t *dev_rows; // window of device row indices
  void *out[kMaxOuts];
  uint64_t *out_valid[kMaxOuts]; // Arrow validity bitmap of a nullable output (64 rows per word), else nullptr
  uint32_t *error_flag;
  uint32_t error_stride;         // rows per error cell (error_flag[i / error_stride]); 0: one cell for the launch
  int64_t lit_i[kMaxLits];
  double lit_f[kMaxLits];
  uint32_t n;
  uint32_t pad_;
};


// Arguments of the group-reduce kernel of the sort-based GROUP BY: one wave per group over the group's segment
// of the sorted selection.
struct ReduceParams {
  const void *col[kMaxCols];
  const uint32_t *perm;      // sorted position → index into the selection arrays
  const uint64_t *dev_rows;  // selection: device row index
  const uint64_t *row_ids;   // selection: logical row id
  const uint64_t *seg_start; // [n_groups + 1] segment bounds in sorted positions
  const uint32_t *order;     // output position → segment (nullptr = identity): groups leave in their final order
  uint64_t *out;             // [n_groups][K] lanes: rows, first row id, aggregate lanes
  // DISTINCT aggregates (all over one column, which sorted last): per sorted position the column's value and whether the
  // row is the first of its group with that value (NULL cells never are); nullptr when the plan has none
  const uint64_t *dval;
  const uint8_t *dhead;
  const uint64_t *first_rows; // … and (rows inside a group are then not in row order) every segment's smallest row id
  uint32_t *error_flag;
  int64_t lit_i[kMaxLits];
  double lit_f[kMaxLits];
  uint64_t n_groups;
};

// Arguments of the standalone fold_octants_kernel.
struct FoldParams {
  const uint64_t *tile_partials; // [lanes][n_tiles]
  uint64_t *exchange;            // [kOctants][lanes]
  const uint8_t *lane_ops;       // [lanes]
  uint32_t octant_tile_begin[kOctants + 1];
  uint32_t n_tiles;
  uint32_t lanes;
  uint32_t owned_mask;
  uint32_t parts_per_tile; // partials per tile: 1, or one per wave for LDS-accumulator plans
};

} // namespace llkv
