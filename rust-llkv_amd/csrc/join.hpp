// join.hpp — device entry points of the integer-key hash join (join.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "scan_params.h"

#include <cstdint>

namespace llkv {

struct JoinKeyColumn {
  const void *values; // device column image
  uint32_t width;     // 4 or 8 bytes
  uint32_t is_signed; // sign-extend 4-byte keys
  // NULL cells: a NULL key matches nothing (llkv-join/src/hash_join.rs:1116-1123,1172-1177) unless the join
  // key says null_equals_null, in which case the reference substitutes a per-type sentinel — and a real key of
  // that value then joins with the NULLs (:1429-1465); restated as is.
  const uint8_t *valid = nullptr; // 1 B/row validity mask or nullptr (no NULL cells)
  uint32_t null_equals_null = 0;
  long long null_sentinel = 0;
};

// One column of a join key in the form both join routes share.  Every cell becomes a canonical (value, is_null)
// pair or "no key" (the row then matches nothing):
//  · integer fast path (one key, Int32/Int64/UInt32/UInt64 on both sides, hash_join.rs:174-198): a NULL under
//    null_equals_null becomes the per-type sentinel VALUE (null_is_value), so a real key of that value joins the
//    NULLs (:1429-1465);
//  · generic typed-key path (any other key list, hash_join.rs:62-148,377-505): values are equal when their types
//    and bits are (floats by bit pattern); values of two different types never are (`values_never_match`); a NULL
//    is KeyValue::Null (equal to nothing) or, under null_equals_null, the marker Utf8("<NULL>") — equal to the
//    other NULLs and to a real string "<NULL>"; key types extract_key_value does not list (Date32, Boolean,
//    Decimal128) make the extraction fail, which drops the row from the build and leaves a probe row unmatched
//    (`unusable`).
// Utf8 columns are dictionary codes; `translate` maps the probe table's codes into the build table's code space
// (0xFFFF: the string does not occur on the build side).
struct JoinKeyPart {
  const void *values;
  const uint8_t *valid;      // 1 B/row validity mask or nullptr
  const uint16_t *translate; // 256 entries (device) or nullptr
  uint32_t width;            // 1, 4 or 8 bytes
  uint32_t is_signed;
  uint32_t null_equals_null;
  uint32_t null_is_value;
  long long null_value;
  uint32_t values_never_match;
  uint32_t unusable;
  // executor rules (normalize_join_column, llkv-executor/src/lib.rs:12405-12427): a Float32 cell is compared as the
  // Float64 it casts to; a UInt64 ≥ 2^63 does not survive the cast to Int64 (NULL: no key)
  uint32_t f32_as_f64;
  uint32_t u64_high_is_null;
};
constexpr uint32_t kMaxJoinKeys = 8; // key pairs of one join (the reference has no limit, hash_join.rs:200-335: kernel arguments have)
struct JoinKeySet {
  JoinKeyPart k[kMaxJoinKeys];
  uint32_t n;
  // 1 B/row mask or nullptr: a row with 0 is not part of the join at all — the reference's scan of the side dropped it
  // (NULL in every user column, GatherNullPolicy::DropNulls): it is not built, matches nothing and a LEFT / ANTI join
  // does not emit it (join_emit.cpp)
  const uint8_t *live;
};

// Build side: distinct keys claim slots of an open-addressing table (slot_owner = row that owns the slot,
// UINT64_MAX = empty); every build row records its slot.  All arrays are device memory.
hipError_t hj_launch_claim(const JoinKeySet &key, const TileDesc *tiles, uint32_t n_tiles, uint32_t tile_rows,
                           unsigned long long *slot_owner, uint64_t cap_mask, uint32_t *slot_of_row /*[dev rows]*/,
                           uint64_t *dev_row_of /*[n_build]: compact index → device row*/,
                           uint64_t *logical_of /*[n_build]*/, const uint64_t *tile_compact_base, hipStream_t s);
// Stable sort of (slot, compact build index) pairs by slot → build rows of one key are contiguous and in
// insertion (row) order.  Returns the rocPRIM status; *tmp_bytes is queried when tmp == nullptr.
hipError_t hj_sort_by_slot(void *tmp, size_t *tmp_bytes, const uint32_t *slot_in, uint32_t *slot_out,
                           const uint32_t *idx_in, uint32_t *idx_out, uint32_t n, uint32_t slot_bits, hipStream_t s);
// seg_start[slot] / seg_count[slot] from the sorted slot list.
hipError_t hj_launch_segments(const uint32_t *sorted_slot, uint32_t n, uint32_t *seg_start, uint32_t *seg_count, hipStream_t s);
// identity index 0..n-1
// Fills `bytes` (rounded up to 16; the block must be 16-byte aligned and that large) with a 64-bit pattern.
hipError_t hj_launch_fill(void *p, uint64_t bytes, uint64_t pattern, hipStream_t s);
hipError_t hj_launch_iota(uint32_t *out, uint32_t n, hipStream_t s);
hipError_t hj_launch_iota_u64(uint64_t *out, uint64_t n, uint64_t first, hipStream_t s); // out[i] = first + i
hipError_t hj_launch_bits_to_bytes(const uint64_t *bits, uint64_t n, uint8_t *out, hipStream_t s); // Arrow validity bits → 1 B/row
// Up to four zero fills in one launch (same alignment rule).
struct FillRanges {
  void *p[4];
  uint64_t bytes[4];
  int n = 0;
  void add(void *ptr, uint64_t b) { if (b) { p[n] = ptr; bytes[n] = b; ++n; } }
};
hipError_t hj_launch_fill_zero_ranges(const FillRanges &r, hipStream_t s);

// Small device → pinned-host copies carried by one workgroup (Readback, engine.hpp): item i = words[i] 32-bit words from
// src[i] to host word dst_word[i].
// When seq != 0 the workgroup finally stores seq into host word flag_word: the host may poll that word instead of
// synchronising the stream.
struct GatherItems {
  const uint32_t *src[12];
  uint32_t dst_word[12], words[12];
  int n;
  uint32_t flag_word, seq;
};
hipError_t hj_launch_readback_gather(const GatherItems &g, uint32_t *host, hipStream_t s);

// Cross product of two row windows, left-major (cross_join_pair llkv-join/src/cartesian.rs:22-80):
// pair i = (l0 + i / rn, r0 + i % rn).
hipError_t hj_launch_cross_pairs(uint64_t l0, uint64_t ln, uint64_t r0, uint64_t rn, uint64_t *out_left, uint64_t *out_right, hipStream_t s);

struct ProbeParams {
  JoinKeySet lkey, rkey;
  const TileDesc *tiles; // probe-side tiles of this window
  uint32_t n_tiles, tile_rows;
  const unsigned long long *slot_owner;
  uint64_t cap_mask;
  const uint32_t *seg_start, *seg_count; // per slot
  const uint32_t *sorted_idx;            // compact build indices grouped by slot, insertion order
  const uint64_t *build_logical;         // compact build index → logical row id
  int32_t join_type;                     // llkv_join_type
  uint64_t *counts;                      // [n_tiles * tile_rows] pairs emitted per probe position
  uint32_t *match_slot;                  // [n_tiles * tile_rows] matching slot or UINT32_MAX
  const uint64_t *offsets;               // exclusive scan of counts
  uint64_t *out_left, *out_right;        // pair output
  // hj_launch_probe_write_rows: device row indices in the batch layout of join_emit.cpp
  const uint64_t *build_dev;             // compact build index → device row
  const uint64_t *cuts;                  // pair index where batch b + 1 of this step starts, ascending
  uint32_t n_cuts;
  const int64_t *batch_shift;            // [n_cuts + 1] position of a batch's pairs in the output − their pair index
};
hipError_t hj_launch_probe_count(const ProbeParams &p, hipStream_t s);
hipError_t hj_launch_probe_write(const ProbeParams &p, hipStream_t s);
// The write pass for device-side materialisation: out_left / out_right receive DEVICE row indices (right: ~0 = the NULL
// padding of a LEFT join) at pair index + batch_shift[batch of the pair] — every batch of the step starts on a multiple
// of 64 rows, so the gather kernels pack validity words that belong to one batch.
hipError_t hj_launch_probe_write_rows(const ProbeParams &p, hipStream_t s);

// Where the reference cuts the pairs of a probe step into batches, from the exclusive scan of the per-position pair
// counts alone (hash_join.rs:1181-1213, :509-565): the step's positions are cut into segments [seg_pos[k], seg_pos[k+1])
// — a scan batch of 65 536 probe rows, or a slice of batch_size rows of it on the generic path — inside a segment a batch
// ends after the probe row that brings it to >= batch_size pairs, and at the segment's end (unless `last_open`: the
// segment goes on in the next step).  `carry_in` / `carry_out`: pairs of the running batch left over by the previous step /
// by this one.  Pass 1 (cuts == nullptr) counts the cuts of every segment into seg_cuts[k]; pass 2 writes
// them at seg_cut_base[k] (the exclusive scan of the counts).
struct CutParams {
  const uint64_t *offsets;
  const uint32_t *seg_pos; // [n_seg + 1]
  uint32_t n_seg, last_open;
  uint64_t batch_size;
  uint64_t carry_in;
  uint64_t *carry_out;     // device word
  uint64_t *seg_cuts;      // [n_seg + 1], pass 1 (entry n_seg = 0)
  const uint64_t *seg_cut_base;
  uint64_t *cuts;          // pass 2
};
hipError_t hj_launch_batch_cuts(const CutParams &p, hipStream_t s);

// live[dev row] = some column of `valid` (1 B/row masks, up to 32) holds a value in this row; *dead += rows without any
struct LiveMaskCols {
  const uint8_t *valid[32];
  uint32_t n;
};
hipError_t hj_launch_live_mask(const LiveMaskCols &cols, const TileDesc *tiles, uint32_t n_tiles, uint8_t *live, unsigned long long *dead, hipStream_t s);

// Cross product of two windows of selected rows (device row lists), left-major: pair i = (lrows[i / rn], rrows[i % rn])
hipError_t hj_launch_cross_rows(const uint64_t *lrows, uint64_t ln, const uint64_t *rrows, uint64_t rn, uint64_t *out_left, uint64_t *out_right, hipStream_t s);

// ---- join → GROUP BY → top-k pipeline pieces (join_agg.cpp) ---------------------------
// Claim the keys of the listed build rows; *dup_flag is set when a key occurs twice.
hipError_t hj_launch_claim_list(const JoinKeyColumn &key, const uint64_t *dev_rows, uint64_t n, unsigned long long *slot_owner,
                                uint64_t cap_mask, uint32_t *dup_flag, hipStream_t s);
// flags[i] = 1 when the foreign key of listed row i is present in the table.
hipError_t hj_launch_semi_flags(const JoinKeyColumn &fk, const uint64_t *dev_rows, uint64_t n, const JoinKeyColumn &set_key,
                                const unsigned long long *set_owner, uint64_t set_mask, uint64_t *flags, hipStream_t s);
hipError_t hj_launch_compact(const uint64_t *in, const uint64_t *flags, const uint64_t *offsets, uint64_t n, uint64_t *out, hipStream_t s);
hipError_t hj_exclusive_scan_u64(void *tmp, size_t *tmp_bytes, const uint64_t *in, uint64_t *out, uint64_t n, hipStream_t s);
hipError_t hj_sort_u32_u64(void *tmp, size_t *tmp_bytes, const uint32_t *kin, uint32_t *kout, const uint64_t *vin, uint64_t *vout,
                           uint64_t n, uint32_t bits, hipStream_t s);
hipError_t hj_sort_u64_u32(void *tmp, size_t *tmp_bytes, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout,
                           uint64_t n, hipStream_t s);
// Left-to-right f64 sum of every run of equal slots (rows of a group in scan order: the reference's
// accumulation order) → sum_by_slot[slot]; topk_key[slot] = descending order key of the sum (slots
// without a group sort last).
hipError_t hj_launch_segment_sums(const uint32_t *sorted_slot, const uint64_t *sorted_val, uint64_t n, double *sum_by_slot,
                                  uint64_t *count_by_slot, hipStream_t s);
// The same over unsorted pairs: runs are summed where they lie; *multi_run is set when a group has two runs
// (count_by_group must start at zero) — the caller then sorts.
// run sums straight from the probe's stripes; flags[0]: some group had two runs, flags[1]: a stripe count carried the predicate-error mark
// The stripes of a probe that emitted key-bit positions instead of group ids (ScanParams::bm_emit_keybit): the group of a
// position = the number of set bits before it (bits == nullptr: the stripes hold group ids already).
struct RankCols {
  const uint64_t *bits;
  const uint32_t *prefix; // set bits before each word, within its chunk of 2^chunk_shift words
  const uint32_t *base;   // set bits before each chunk (nullptr: one chunk)
  uint32_t chunk_shift;
  // optional (run sums): the head of every run leaves its key-bit position at pos_out[group] — the top-k candidates' keys are then
  // one load away (CandidateCols::pos_by_group) instead of a search of the rank structure for the g-th set bit
  uint32_t *pos_out = nullptr;
};
hipError_t hj_launch_run_sums_stripes(const uint32_t *stripe_group, const uint64_t *stripe_val, const uint64_t *counts, uint32_t n_slots, uint32_t stripe,
                                      double *sum_by_group, uint64_t *count_by_group, uint32_t *flags, hipStream_t s, RankCols rank = RankCols{nullptr, nullptr, nullptr, 0},
                                      uint64_t *slice_best = nullptr /* [2 · kTopkSlices], zero: the top-k selection's slice winners (~best key) and group counts, on the way */,
                                      uint64_t *total_pairs = nullptr /* zero: + the number of pairs (with slice_best) */);
// `descending` (optional): raised when a run starts below the group of the pair before it
hipError_t hj_launch_run_sums_dev(const uint32_t *group, const uint64_t *val, const uint64_t *n_dev, uint64_t n_max, double *sum_by_group,
                                  uint64_t *count_by_group, uint32_t *multi_run, hipStream_t s, uint32_t *descending = nullptr);
hipError_t hj_launch_run_sums(const uint32_t *group, const uint64_t *val, uint64_t n, double *sum_by_group, uint64_t *count_by_group,
                              uint32_t *multi_run, hipStream_t s);
hipError_t hj_launch_topk_keys(const double *sum_by_slot, const uint64_t *count_by_slot, uint64_t cap, uint64_t *keys, uint32_t *slots,
                               unsigned long long *n_groups /* += groups */, hipStream_t s);
// One record per candidate slot: {sort key, dim key, sum bits, count, payload[4]} (8 × 8 bytes).
struct CandidateCols {
  JoinKeyColumn key;
  JoinKeyColumn payload[4];
  uint32_t n_payload;
  // Groups identified by the RANK of their key among the set bits of a bitmap (a dimension selected straight into the
  // bitmap from a key column in ascending row order: no list of the selected rows exists).  rank_bits != nullptr: the
  // owner row of group g = the row of the key column (rank_rows rows, ascending) that holds the g-th set bit's key.
  const uint64_t *rank_bits;
  const uint32_t *rank_prefix; // set bits before each word, within its chunk of 2^rank_chunk_shift words
  const uint32_t *rank_base;   // [chunks + 1] set bits before each chunk; the last entry = the number of groups
  uint32_t rank_chunk_shift, rank_chunks;
  uint64_t rank_words;
  int64_t rank_kmin;
  uint64_t rank_rows;
  const uint32_t *pos_by_group; // optional: the key-bit position of every group that has rows (RankCols::pos_out)
};
hipError_t hj_launch_gather_candidates(const uint64_t *sorted_keys, const uint32_t *sorted_slots, uint32_t n, const unsigned long long *slot_owner,
                                       const double *sum_by_slot, const uint64_t *count_by_slot, CandidateCols cols, uint64_t *out /*[n][8]*/, hipStream_t s);

// ---- group ids that do not depend on hash-table timing (sharded fact tables exchange per-group state) ------
// slot_group[slot] = i for the i-th listed dim row (the list is in row order, identical on every rank).
hipError_t hj_launch_slot_groups(const JoinKeyColumn &key, const uint64_t *dev_rows, uint64_t n, const unsigned long long *slot_owner,
                                 uint64_t cap_mask, uint32_t *slot_group, hipStream_t s);
hipError_t hj_launch_map_u32(uint32_t *inout, uint64_t n, const uint32_t *table, hipStream_t s);
// Compaction of the single-pass probe output: stripe `slot` holds counts[slot] pairs at slot·stripe; they move to
// offsets[slot] with the hash slot translated to its group id.
hipError_t hj_launch_compact_stripes(const uint32_t *stripe_slot, const uint64_t *stripe_val, const uint64_t *counts, const uint64_t *offsets,
                                     uint32_t n_slots, uint32_t stripe, const uint32_t *slot_group, uint32_t *out_group, uint64_t *out_val, hipStream_t s,
                                     RankCols rank = RankCols{nullptr, nullptr, nullptr, 0});
// v[i] += delta (wrapping)
hipError_t hj_launch_add_u64(uint64_t *v, uint64_t n, uint64_t delta, hipStream_t s);
// *flag |= 1 when keys[i] < keys[i − 1] for some i (flag zeroed by the caller)
hipError_t hj_launch_unsorted_flag(const uint64_t *keys, uint64_t n, uint32_t *flag, hipStream_t s);
// … and of a single-pass selection (two u64 streams: row ids, device rows)
hipError_t hj_launch_compact_stripes2(const uint64_t *stripe_a, const uint64_t *stripe_b, const uint64_t *counts, const uint64_t *offsets, uint32_t n_slots,
                                      uint32_t stripe, uint64_t *out_a, uint64_t *out_b, hipStream_t s);
// The same, and bit (key − kmin) of every selected row is set in `bits` (a duplicate raises *dup_flag).
hipError_t hj_launch_compact_stripes2_bits(const uint64_t *stripe_a, const uint64_t *stripe_b, const uint64_t *counts, const uint64_t *offsets, uint32_t n_slots,
                                           uint32_t stripe, uint64_t *out_a, uint64_t *out_b, const void *key_values, uint32_t key_width, uint32_t key_signed,
                                           long long kmin, unsigned long long *bits, uint32_t *unsorted_flag, hipStream_t s);
// Direct-address form of a dim table whose key range is bounded by the column statistics: bit (key − kmin) set for
// every listed row (*dup_flag when a key occurs twice; dev_rows == nullptr lists rows 0..n−1), per-word popcounts →
// exclusive scan = rank of each word's first set bit, group_of_rank[rank(key of listed row i)] = i.
// (`slot_group` of hj_launch_compact_stripes may then be nullptr: the stripes already hold group ids.)
hipError_t hj_launch_bitmap_build(const JoinKeyColumn &key, const uint64_t *dev_rows, uint64_t n, long long kmin, unsigned long long *bits,
                                  uint32_t *dup_flag, hipStream_t s);
hipError_t hj_launch_popc_words(const uint64_t *bits, uint64_t n_words, uint32_t *out, hipStream_t s);
hipError_t hj_exclusive_scan_popc(void *tmp, size_t *tmp_bytes, const uint64_t *bits, uint32_t *out, uint64_t n_words, hipStream_t s);
// The same in ONE launch, as chunk-local ranks: prefix[w] = set bits before word w within its chunk of 2^chunk_shift words
// (one workgroup per chunk), base[c] = set bits before chunk c, base[n_chunks] = all set bits — written by the last workgroup
// to finish.  `state`: one zero word (left zero).  n_chunks <= 1024.
hipError_t hj_launch_rank_words(const uint64_t *bits, uint64_t n_words, uint32_t chunk_shift, uint32_t *prefix, uint32_t *base, uint32_t *state, hipStream_t s);
hipError_t hj_exclusive_scan_u32(void *tmp, size_t *tmp_bytes, const uint32_t *in, uint32_t *out, uint64_t n, hipStream_t s);
hipError_t hj_launch_bitmap_groups(const JoinKeyColumn &key, const uint64_t *dev_rows, uint64_t n, long long kmin, const uint64_t *bits,
                                   const uint32_t *prefix, uint64_t n_words, const uint32_t *unsorted /*optional: zero = rank is the list index, nothing to write*/,
                                   uint32_t *dup_flag /*optional: raised when the bitmap holds fewer bits than n*/, uint32_t *group_of_rank, hipStream_t s);
// flags[i] = 1 when the group of sorted pair i has rows on another rank too (local count ≠ global count).
hipError_t hj_launch_straddler_flags(const uint32_t *sorted_group, uint64_t n, const uint64_t *local_cnt, const int64_t *global_cnt,
                                     uint64_t *flags, hipStream_t s);
hipError_t hj_launch_compact_pairs(const uint32_t *group, const uint64_t *val, const uint64_t *flags, const uint64_t *offsets, uint64_t n,
                                   uint32_t *out_group, uint64_t *out_val, hipStream_t s);
// report[g] = rows of group g when this rank alone holds it, else 0 (straddlers are patched in afterwards).
hipError_t hj_launch_report_counts(const uint64_t *local_cnt, const int64_t *global_cnt, uint64_t n, uint64_t *report, hipStream_t s);
hipError_t hj_launch_patch_groups(const uint32_t *groups, const double *sums, const uint64_t *counts, uint64_t n, double *sum_by_group,
                                  uint64_t *report, hipStream_t s);
// gather_candidates for group ids: owner row = dim_rows[group].
// (`sorted_keys` may be nullptr: the key of candidate i is then `keys_by_group[sorted_groups[i]]`.)
// Top-k by selection: the groups are cut into slices, each reports its best order key; the want-th best of those
// bounds the final top `want` from below, and the groups that reach it (in no particular order, at most `cap` of them)
// come back as candidate records {group, dim key, sum bits, count, payload[4]}.
// Two launches, nothing to zero per call and the answer written to the host by the second one: the groups
// are cut into kTopkSlices slices; the last workgroup of the first launch orders the slices' best keys and leaves the
// bound; the last workgroup of the second launch writes {bound, candidates, groups with rows, 0…}[8] and the candidate
// records ([min(candidates, cap)][8]) to `host_out` (pinned, device-visible) and carries the `extra` read-back items.
// `state`: 8 words, zero before the first use (the kernels leave it zero); `best`: 2 · kTopkSlices words.  want ≤ kTopkSlices.
constexpr uint32_t kTopkSlices = 256;
// `n_dev` (optional): the number of groups is still on the device (then `n` only bounds it).
hipError_t hj_launch_topk_select2(const double *sums, const uint64_t *counts, uint64_t n, uint32_t want, uint32_t cap, const uint64_t *dim_rows, CandidateCols cols,
                                  uint64_t *best, uint64_t *state, uint32_t *groups /*[cap]*/, uint64_t *host_out, const GatherItems &extra, uint32_t *extra_host,
                                  hipStream_t s, const uint32_t *n_dev = nullptr, const uint64_t *slice_best = nullptr /* filled by hj_launch_run_sums_stripes: no first launch */);
hipError_t hj_launch_high_halves(const uint64_t *keys, uint64_t n, uint32_t *out, hipStream_t s);
// Range form of a sharded fact table: the first and the last run of a pair stream with their key bits and raw values
// (join.hip: hj_boundary_runs_kernel; out = 8 + 2 · cap words of device memory)
// `n_dev` (optional): the pair count is still on the device
hipError_t hj_launch_boundary_runs(const uint32_t *group, const uint64_t *val, uint64_t n, const uint64_t *n_dev, uint32_t cap, CandidateCols cols, uint64_t *out, hipStream_t s);
// … straight from the probe's stripes (rank.bits != nullptr: they hold key-bit positions)
hipError_t hj_launch_boundary_runs_stripes(const uint32_t *stripe_group, const uint64_t *stripe_val, const uint64_t *counts, uint32_t n_slots, uint32_t stripe, uint32_t cap,
                                           CandidateCols cols, RankCols rank, uint64_t *out, hipStream_t s);
hipError_t hj_launch_gather_group_candidates(const uint64_t *sorted_keys, const uint64_t *keys_by_group, const uint32_t *sorted_groups, uint32_t n, const uint64_t *dim_rows,
                                             const double *sum_by_group, const uint64_t *count_by_group, CandidateCols cols,
                                             uint64_t *out /*[n][8]*/, hipStream_t s);

// ---- sort-based GROUP BY (group_sort.cpp): generic key handling -------------------------------------------
// keys[i] = value − base (as u64) of column `col` at selected row perm[i]: order preserving when base is the
// column minimum (statistics) or the type minimum (i64::MIN ⇔ flipping the sign bit); NULL cell → 0, told
// apart by the validity pass.
// `code_rank` (256 entries, or nullptr): dictionary code → rank of its string, so that code columns sort as strings.
hipError_t hj_launch_gather_sort_keys(const JoinKeyColumn &col, long long base, const uint8_t *code_rank, const uint64_t *dev_rows, const uint32_t *perm,
                                      uint64_t n, uint64_t *keys, hipStream_t s);
// first_rows[g] = logical row id of the first row of segment g (the stable sort keeps row order inside a group).
hipError_t hj_launch_segment_min_rows(const uint64_t *row_ids, const uint32_t *perm, const uint64_t *seg_start, uint64_t n_groups, uint64_t *first_rows, hipStream_t s);
hipError_t hj_launch_first_rows(const uint64_t *row_ids, const uint32_t *perm, const uint64_t *seg_start, uint64_t n_groups, uint64_t *first_rows, hipStream_t s);
hipError_t hj_launch_gather_valid(const JoinKeyColumn &col, const uint64_t *dev_rows, const uint32_t *perm, uint64_t n, uint32_t *out, hipStream_t s);
struct GroupKeySet {
  JoinKeyColumn k[4];
  uint32_t n;
};
// flags[i] = 1 when sorted position i starts a new group (i == 0 or some key differs from position i − 1).
hipError_t hj_launch_group_boundaries(GroupKeySet ks, const uint64_t *dev_rows, const uint32_t *perm, uint64_t n, uint64_t *flags, hipStream_t s);
// seg_start[offsets[i]] = i for every flagged i; seg_start[n_groups] = n.
// per sorted position: what a DISTINCT lane adds for the cell of `col` (numeric 0: the 64-bit cell of an Int64 / Float64 column;
// 1: dict_num[code] of a Utf8 column; 2: a Boolean's 1.0 / 0.0; 3: a Date32's day number as f64) and "first of its group with this cell"
hipError_t hj_launch_distinct_heads(const JoinKeyColumn &col, uint32_t numeric, const double *dict_num, const uint64_t *dev_rows, const uint32_t *perm,
                                    const uint64_t *group_start, uint64_t n, uint64_t *dval, uint8_t *dhead, hipStream_t s);
hipError_t hj_launch_segment_starts(const uint64_t *flags, const uint64_t *offsets, uint64_t n, uint64_t n_groups, uint64_t *seg_start, hipStream_t s);
// Raw key cells of each group's first sorted row: out_vals[k][g] (sign-extended to i64), out_valid[k][g].
hipError_t hj_launch_group_keys(GroupKeySet ks, const uint64_t *dev_rows, const uint32_t *perm, const uint64_t *seg_start, const uint32_t *order, uint64_t n_groups,
                                int64_t *out_vals /*[n_keys][n_groups]*/, uint8_t *out_valid /*[n_keys][n_groups]*/, hipStream_t s);
hipError_t hj_sort_u64_u32_bits(void *tmp, size_t *tmp_bytes, const uint64_t *kin, uint64_t *kout, const uint32_t *vin, uint32_t *vout,
                                uint64_t n, uint32_t end_bit, hipStream_t s);

// ---- DISTINCT aggregates (engine.cpp: Query::distinct_value) ---------------------------------------------
// flags[i] = 1 when sorted value i differs from sorted value i − 1 (the head of a run of equal keys).
hipError_t hj_launch_run_heads(const uint64_t *sorted, uint64_t n, uint64_t *flags, hipStream_t s);
// Left-to-right f64 sum of vals[0..n) (as_int: the values are i64, added as `v as f64`; else f64 bit images):
// strictly sequential up to 65 536 values (bit-exact with the reference's `sum += v`), beyond that blocks of 65 536
// consecutive values (256 threads, strided, fixed butterfly) whose sums are added in block order.  *out = the sum.
hipError_t hj_launch_sum_f64_ordered(const uint64_t *vals, uint64_t n, int as_int, double *out, double *scratch /* ⌈n / 65 536⌉ doubles */, hipStream_t s);

// ---- join → GROUP BY with an aggregate list (join_group.cpp) --------------------------------------------------------------------
// For m group keys: bisection in the qualifying dimension rows sorted by key image (value − base) → the row's payload cells
// (out_payload[c][i], sign-extended; out_valid[c][i] = 0 for a NULL cell) and its position among the qualifying rows in row order
// (out_pos[i]; ~0u when the key is not there).
struct JoinPayloadCols {
  JoinKeyColumn col[4];
  uint32_t n;
};
hipError_t hj_launch_lookup_payload(const uint64_t *sorted_keys, const uint64_t *sorted_rows, const uint32_t *sorted_pos, uint64_t n_dim, const int64_t *probe_keys,
                                    long long base, uint64_t m, const JoinPayloadCols &cols, int64_t *out_payload, uint8_t *out_valid, uint32_t *out_pos, hipStream_t s);

// ---- ordered scans (stream.cpp) -----------------------------------------------------------------------------
hipError_t hj_launch_gather_u64(const uint64_t *in, const uint32_t *perm, uint64_t n, uint64_t *out, hipStream_t s);
// out[i] = in[idx[i]] (64-bit indices: device row → row id of a table whose ids are not its positions)
hipError_t hj_launch_gather_u64_by_row(const uint64_t *in, const uint64_t *idx, uint64_t n, uint64_t *out, hipStream_t s);
hipError_t hj_launch_xor_u64(uint64_t *keys, uint64_t n, uint64_t mask, hipStream_t s);
hipError_t hj_launch_xor_u32(uint32_t *keys, uint64_t n, uint32_t mask, hipStream_t s);

// Does any prefix of vals[0..n) (summed left to right, exactly) leave the i64 range?  *d_flag |= 1 if so.
// `tmp` sized by a first call with tmp == nullptr; d_prefix holds n 16-byte elements.
hipError_t hj_prefix_overflow(void *tmp, size_t *tmp_bytes, const int64_t *vals, uint64_t n, void *d_prefix, uint32_t *d_flag, hipStream_t s);

// ---- partitioned GROUP BY (group_part.cpp) ----------------------------------------------------------------------------
// group_rows = [ng][k] lanes, lane 0 = rows of the group: ids of the groups with rows, ascending
hipError_t hj_select_present_groups(void *tmp, size_t *tmp_bytes, const uint64_t *group_rows, uint32_t k, uint32_t ng, uint32_t *ids, uint32_t *count, hipStream_t s);
hipError_t hj_launch_gather_lane(const uint64_t *group_rows, uint32_t k, uint32_t lane, const uint32_t *ids, uint32_t n, uint64_t *out, hipStream_t s);
struct DenseKeyLayout { // dense group id = Σ code_j · stride_j (plan.hpp: key_strides / key_cards / key_bases / key_nullable)
  uint32_t n;
  uint32_t stride[4], card[4], nullable[4];
  long long base[4];
  const uint32_t *code_rank[4]; // ORDER BY the keys: dictionary code → position in string order (Utf8 keys), else nullptr
};
// ORDER BY the keys (NULLS FIRST, strings by their bytes): the dense id of every listed group with each digit replaced by
// its rank in that order — sorting by it orders the groups as the sort-based route does
hipError_t hj_launch_dense_group_order_keys(const uint32_t *ids, uint32_t n, const DenseKeyLayout &keys, uint64_t *order_keys, hipStream_t s);
hipError_t hj_launch_emit_dense_groups(const uint64_t *group_rows, uint32_t k, const uint32_t *ids, const uint32_t *order, uint32_t n, const DenseKeyLayout &keys,
                                       uint64_t *lanes_out, int64_t *key_vals, uint8_t *key_valid, hipStream_t s);

} // namespace llkv
