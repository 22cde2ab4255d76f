// engine.hpp — internal host-side types of libllkv_hip (see engine.cpp).
#pragma once

#include <hip/hip_runtime.h>

#include "catalog.hpp"
#include "llkv_hip.h"
#include "plan.hpp"
#include "scan_params.h"

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace llkv {

constexpr int kOctantsHost = kOctants;

extern thread_local std::string g_last_error;
int set_error(int code, const std::string &msg);

// HIP call → status: returns from the calling function with the error recorded
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return llkv::set_error(_e == hipErrorNoDevice || _e == hipErrorInvalidDevice ? LLKV_NO_DEVICE : LLKV_INTERNAL, \
                             std::string(#expr) + ": " + hipGetErrorString(_e));                    \
  } while (0)

struct Context {
  std::mutex mu;
  bool ready = false;
  int device = -1;
  hipStream_t stream = nullptr;
  uint32_t cu_count = 256; // hipDeviceProp_t::multiProcessorCount of the bound device (MI355X: 256); sizes the persistent grids
};
extern Context g_ctx;
int ensure_device();

struct DeviceColumn {
  ColumnInfo info;
  void *d_values = nullptr;
  bool has_local_stats = false; // min / max of this rank's rows (info.has_stats / min_i / max_i are table-wide)
  int64_t local_min = 0, local_max = 0;
  bool has_local_fstats = false; // float columns: largest finite |v| of this rank's rows (info.has_fstats / f_absmax are table-wide)
  double local_f_absmax = 0.0, local_f_absmin_nz = 0.0;
  bool local_f_all_finite = false; // … and none of this rank's values is NaN / ±∞
  bool local_f_no_neg_zero = false; // … nor −0.0
  uint8_t *d_valid = nullptr; // 1 B/row validity mask (info.nullable), same row layout as d_values
  void *d_hi = nullptr;       // Decimal128 values beyond 64 bits (info.wide128): d_values holds the low halves, this the high halves
  bool owned = false;
  // rows the buffers above can hold (slack included); 0 = exactly the table's image + slack as allocated at staging.  An append
  // (llkv_hip_table_append_chunks) that outgrows it moves the column into a larger buffer with headroom.
  uint64_t cap_rows = 0, valid_cap_rows = 0;
};

// Device buffer read by slot `s` of a lowered plan: the field's values, its validity mask, or the high halves of a wide
// Decimal128 column.
inline const void *slot_buffer(const std::map<uint32_t, DeviceColumn> &cols, const LoweredPlan &p, size_t s) {
  const DeviceColumn &c = cols.at(p.slot_fields[s]);
  const uint8_t part = s < p.slot_is_valid.size() ? p.slot_is_valid[s] : 0;
  return part == 1 ? (const void *)c.d_valid : part == 2 ? (const void *)c.d_hi : (const void *)c.d_values;
}

struct TileSet {
  TileDesc *d_tiles = nullptr;
  uint32_t n_tiles = 0;
  uint32_t tile_rows = 0;
  // every kTileSampleStride-th tile, for selectivity estimates (stream.cpp: run_selection_lowered)
  TileDesc *d_sample = nullptr;
  uint32_t n_sample = 0;
  uint64_t sample_rows = 0;
  uint32_t octant_tile_begin[kOctantsHost + 1] = {0};
};

// A 4-byte image of an Int64 column whose statistics fit 32 bits (table.cpp: get_key_image): what the streaming scans of the join
// pipeline read in place of the 8-byte column — the order-key column is a third of the bytes the Q3 probe streams, two thirds of the
// order-bits scan's.  Built on first use (one pass: 8 B in, 4 B out per row), kept with the table, same row layout; an append
// (a new generation) drops it.  Sparse lookups (an owner's key, a payload) stay on the column itself.
struct KeyImage {
  void *d = nullptr;
  ColumnInfo info; // the column's, with dtype = Int32
};

struct Table {
  uint16_t table_id = 0;
  uint32_t rank = 0, world = 1;
  std::vector<uint64_t> global_chunk_rows;
  uint32_t octant_chunk_begin[kOctantsHost + 1] = {0};
  uint32_t owned_mask = 0xff;
  uint32_t first_chunk = 0, n_local_chunks = 0;
  uint64_t total_rows = 0, local_rows = 0, local_logical_start = 0;
  std::vector<uint64_t> chunk_dev_off; // local chunk → first row in the device image (+ end)
  uint64_t dev_rows = 0;
  std::map<uint32_t, DeviceColumn> cols;
  std::map<uint32_t, TileSet> tilesets;
  std::map<uint32_t, KeyImage> key_images;
  // Row ids that are not the positions 0 … n − 1 (llkv_hip_table_set_row_ids): the id of every local row, in the row layout of
  // the column images; nullptr = dense ids.  Everything inside works on positions; the calls that REPORT row ids translate.
  uint64_t *d_row_ids = nullptr;
  uint64_t row_ids_cap = 0;   // rows d_row_ids can hold
  uint64_t last_row_id = 0;   // the id of the table's last row (ids ascend strictly; appended chunks must continue above it)
  // llkv_hip_table_append_chunks: every append is a new generation of the image — buffers may have moved, statistics and tile
  // lists have changed — and a query prepared over an older one refuses to launch (prepare it again: lowering + a cache lookup)
  uint64_t generation = 0;
  std::vector<void *> retired; // tile lists of older generations (freed with the table: a stale handle may still name them)
  std::mutex mu;
  ~Table();
};

void compute_layout(Table &t);
uint32_t octant_of_chunk(const Table &t, uint32_t global_chunk);
void build_tiles_host(const Table &t, uint32_t tile_rows, std::vector<TileDesc> &tiles,
                      uint32_t (&octant_tile_begin)[kOctantsHost + 1]);

// run-time compiled plan (jit.cpp)
enum class JitKind : int { Scan = 0, Select = 1, Project = 2, Probe = 3, Emit = 4, Reduce = 5, Image = 6, KeyBits = 7, Part = 8 };
struct JitKernel {
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;  // scan / select-count / project
  hipFunction_t fn2 = nullptr; // select-write
};
int jit_compile(JitKind kind, const std::string &type_string, JitKernel *out, std::string *err);
int jit_launch(const JitKernel &k, const ScanParams &p, hipStream_t stream);
int jit_launch_raw(hipFunction_t fn, uint32_t grid, void *params, size_t bytes, hipStream_t stream, uint32_t block = kBlock);
void jit_shutdown();

struct GroupKey { // GroupKeyValue: String or Int (llkv-executor/src/lib.rs:99-106)
  bool is_int = false;
  bool is_null = false; // GroupKeyValue::Null
  int64_t i = 0;
  std::string s;
  // ORDER BY key ASC with NULLS FIRST (the caller re-sorts the handful of groups for any other order)
  bool operator<(const GroupKey &o) const { return (is_null || o.is_null) ? (is_null && !o.is_null) : (is_int ? i < o.i : s < o.s); }
  bool operator==(const GroupKey &o) const { return (is_null || o.is_null) ? (is_null == o.is_null) : (is_int ? i == o.i : s == o.s); }
};
// The finished groups of a dense / shared-image plan live in two flat arrays (keys: n_groups × n_keys, values: n_groups ×
// n_aggs): a vector pair per group was two heap allocations each — as much time as the kernel for 2 526 groups.
struct GroupStore {
  size_t n = 0, n_keys = 0, n_values = 0;
  std::vector<GroupKey> keys;
  std::vector<llkv_value> values;
  void reset(size_t groups, size_t keys_per_group, size_t values_per_group) {
    n = 0; n_keys = keys_per_group; n_values = values_per_group;
    keys.clear(); values.clear();
    keys.reserve(groups * keys_per_group);
    values.resize(groups * values_per_group);
  }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  const GroupKey &key(size_t g, size_t k) const { return keys[g * n_keys + k]; }
  llkv_value &value(size_t g, size_t a) { return values[g * n_values + a]; }
  const llkv_value &value(size_t g, size_t a) const { return values[g * n_values + a]; }
};

// Result of a sort-based GROUP BY, kept as the arrays the device produced (already in output order, pinned host
// memory) and finalized cell by cell on request: millions of groups cost no per-group host objects.
struct LazyGroups {
  bool active = false;
  uint64_t n = 0;
  int k = 0;                       // lanes per group: rows, first row id, aggregate lanes
  uint32_t n_keys = 0;
  const uint64_t *lanes = nullptr; // [n][k]
  const int64_t *key_vals = nullptr;   // [n_keys][n] raw key cells (dictionary code / integer)
  const uint8_t *key_valid = nullptr;  // [n_keys][n]
  const LoweredPlan *plan = nullptr;   // aggregate finalization
  std::vector<const ColumnInfo *> key_cols;
};

struct Scratch;

// Message of a device-side arithmetic error code (fused_scan.hip.h: kErrOverflow = 1, kErrDivZero = 2), as the
// reference's arrow kernels word it (Error::Internal).
inline const char *arith_error_message(uint64_t code) {
  return (code & 2u) ? "Divide by zero" : "Arithmetic overflow: Overflow happened in a computed projection";
}

// Partitioned GROUP BY (group_part.cpp): up to 2^24 dense groups with order-free lanes.
struct PartGroupBy;
int part_groupby_prepare(const Table *table, const llkv_filter *filters, uint32_t n_filters, const llkv_eval_op *ops, uint32_t n_ops,
                         const uint32_t *key_fields, uint32_t n_keys, const llkv_aggregate_spec *aggs, uint32_t n_aggs, bool order_by_keys, PartGroupBy **out);
int part_groupby_run(PartGroupBy *p, LazyGroups *out);
void part_groupby_free(PartGroupBy *p);
const LoweredPlan *part_groupby_plan(const PartGroupBy *p);

// Sort-based GROUP BY (group_sort.cpp): any number of groups, any state width.  A query the partitioned route admits is
// handed to it instead (sorted_groupby_partitioned() says so).
struct SortedGroupBy;
bool sorted_groupby_partitioned(const SortedGroupBy *s);
void join_group_state_free(struct JoinGroupState *s); // join_group.cpp
struct KeySetView;
// `key_set` / `key_set_field`: one more conjunct of the selection — the integer column must be in the key set (join → GROUP BY,
// join_group.cpp: the keys of the qualifying dimension rows; the bitmap belongs to the caller and outlives the object)
int sorted_groupby_prepare(const Table *table, const llkv_filter *filters, uint32_t n_filters, const llkv_eval_op *ops, uint32_t n_ops,
                           const uint32_t *key_fields, uint32_t n_keys, const llkv_aggregate_spec *aggs, uint32_t n_aggs,
                           bool order_by_keys, SortedGroupBy **out, const KeySetView *key_set = nullptr, uint32_t key_set_field = 0);
int sorted_groupby_run(SortedGroupBy *s, LazyGroups *out);
void sorted_groupby_free(SortedGroupBy *s);
// Sharded table: the ranks' partial groups ([n_keys][n] key cells and validity, [n][k] lanes per rank, rank order)
// become the table-wide groups of `out` (group_sort.cpp).
int sorted_groupby_merge(SortedGroupBy *s, uint32_t world, const uint64_t *rank_groups, const int64_t *const *key_values,
                         const uint8_t *const *key_valid, const uint64_t *const *lanes, LazyGroups *out);

struct Query {
  const Table *table = nullptr;
  uint64_t table_generation = 0; // the table's generation this query was lowered over (Table::generation)
  LazyGroups lazy;
  SortedGroupBy *sorted = nullptr; // set when the dense GROUP BY kernel cannot hold the groups: executions run synchronously in launch()
  struct JoinGroupState *join_state = nullptr; // join → GROUP BY (join_group.cpp): the dimension side's key set and sorted rows
  LoweredPlan plan;
  const CatalogEntry *entry = nullptr;
  JitKernel jit;
  const TileSet *tiles = nullptr;
  ScanParams params;
  double *d_dict_num = nullptr; // ScanParams::dict_num of plans that read dictionary codes as numbers
  // ring of exchange images so that up to `depth` executions are in flight: the host
  // finalizes execution i while the GPU already runs i+1
  static constexpr uint32_t kMaxDepth = 8;
  uint32_t depth = 1;
  uint64_t n_launched = 0, n_submitted = 0, n_collected = 0;
  hipEvent_t copied[kMaxDepth] = {nullptr};
  // two tile-partial images: execution i+1 scans into one while its first workgroups fold the other
  uint64_t *d_tile_partials = nullptr; // [2][lanes][n_tiles]
  size_t partials_len = 0;
  uint8_t *d_lane_ops = nullptr;       // standalone fold kernel
  FoldParams fold;
  bool pending = false;                // the latest scan's tile partials are not folded yet
  uint32_t pending_slot = 0, pending_pb = 0;
  hipStream_t pending_stream = nullptr;
  uint64_t *d_empty_image = nullptr;    // exchange image of an execution without tiles
  hipEvent_t ev_fold[kMaxDepth] = {nullptr}; // exchange image of the slot complete
  hipStream_t slot_stream[kMaxDepth] = {nullptr};
  std::string route_note;              // which kernel family serves the plan, and why the cheaper ones declined
  uint32_t image_grid = 0;             // shared-image plans: workgroups of the scan (= images the fold combines)
  bool host_mapped = false;            // single rank: the kernel writes the image straight into pinned host memory
  uint64_t *d_exchange = nullptr; // [kMaxDepth][kOctants][lanes]
  uint64_t *h_exchange = nullptr; // pinned, same shape
  size_t h_exchange_bytes = 0;    // size class of the pinned block (pinned_acquire)
  bool order_by_keys = false;
  uint32_t n_user_aggs = 0, n_user_keys = 0;
  GroupStore groups;
  // ungrouped SUM/AVG(Int64) without overflow-excluding statistics: plan that emits the argument values of
  // the selected rows in row order, for the exact prefix-overflow check (index = aggregate, empty = n/a)
  std::vector<LoweredPlan> exact_plans;
  int exact_prefix_overflow(size_t agg, bool *overflow);
  // ungrouped DISTINCT aggregates (COUNT / SUM / TOTAL / AVG): a value-emission plan per aggregate, evaluated
  // at finish by a sort-based pipeline (index = aggregate; kind < 0 = not a DISTINCT aggregate)
  struct DistinctAgg {
    int kind = -1;
    bool is_f64 = false;
    LoweredPlan plan;
    int32_t key_dtype = LLKV_DT_INT64;  // what the emitted 64-bit key stands for: Int64 / Float64 values, a Utf8 dictionary code, Boolean, Date32, the 64-bit image of a Decimal128
    int32_t precision = 0, scale = 0;   // Decimal128
    std::vector<double> key_numeric;    // Utf8: array_value_to_numeric of each dictionary entry
  };
  std::vector<DistinctAgg> distinct;
  int emit_values(const LoweredPlan &ep, Scratch *vals, uint64_t *n);
  int distinct_set(size_t agg, Scratch *dv, uint64_t *m);
  int distinct_value(size_t agg, llkv_value *out);
  // sharded tables: this rank's distinct values on the host / the merged result (llkv_hip_query_distinct_partial, merge_distinct)
  std::vector<std::vector<uint64_t>> distinct_host;
  int distinct_partial(size_t agg, const uint64_t **values, uint64_t *n);
  int merge_distinct(size_t agg, uint32_t world, const uint64_t *counts, const uint64_t *const *values);
  bool profiling = false;
  uint32_t profile_every = 1; // bracket every n-th scan with HIP events
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  size_t events_used = 0;
  uint64_t launches = 0;

  size_t exchange_len() const { return (size_t)kOctants * (size_t)plan.lanes; }
  int launch(hipStream_t stream);
  int flush_pending();
  int wait_folded(hipStream_t stream);
  int all_reduce(hipStream_t stream); // comm.cpp: the exchange image of the oldest unsubmitted execution, summed over the ranks
  int submit(hipStream_t stream);
  int collect();
  int finish(hipStream_t stream);
  int finish_from_exchange(const uint64_t *exchange);
  ~Query();
};

int prepare_query(const Table *table, const llkv_filter *filters, uint32_t n_filters, const llkv_eval_op *ops,
                  uint32_t n_ops, const uint32_t *key_fields, uint32_t n_keys, const llkv_aggregate_spec *aggs,
                  uint32_t n_aggs, bool grouped, bool order_by_keys, Query **out);

int get_tileset(const Table &t, uint32_t tile_rows, const TileSet **out);
// *out = nullptr when the column does not qualify (not Int64, no statistics, a value outside 32 bits, fewer than `min_rows` rows)
int get_key_image(const Table &t, uint32_t field, uint64_t min_rows, const KeyImage **out);

// `configured_thread_count` of the reference's shared Rayon pool (llkv-threading/src/lib.rs:13-31): LLKV_MAX_THREADS
// when it parses to a positive number, else the detected parallelism (affinity mask ∧ cgroup quota, what
// std::thread::available_parallelism reports).  Bounds the library's host-side worker threads (dispatch.cpp).
uint32_t host_thread_limit();

// Caching device scratch allocator (hipMalloc/hipFree cost ~100 µs each; operator pipelines allocate dozens of
// temporaries per call).  Blocks are reused by capacity; everything is released at llkv_hip_shutdown.
void *scratch_alloc(size_t bytes);
void scratch_free(void *p);
bool scratch_can_hold(size_t bytes); // whether scratch_alloc(bytes) could succeed now (admission of memory-hungry routes)
void scratch_release_all();
struct Scratch { // RAII temporary
  void *p = nullptr;
  ~Scratch() { if (p) scratch_free(p); }
  int alloc(size_t bytes) {
    if (p) scratch_free(p);
    p = scratch_alloc(bytes);
    return p ? LLKV_OK : set_error(LLKV_INTERNAL, "device scratch allocation of " + std::to_string(bytes) + " bytes failed");
  }
  template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// Staging lanes (memory.cpp): host chunks → pinned rings → HBM, every piece complete on return.
struct StagePiece {
  void *d_dst;
  const void *h_src;
  size_t bytes;
};
int stage_to_device(const std::vector<StagePiece> &pieces);
int stage_from_pinned(void *d_dst, const void *h_pinned, size_t bytes); // (a block of pinned_acquire)
void staging_totals(uint64_t *bytes, double *seconds);
void staging_release();
uint32_t part_block_threads(); // threads of the partitioned GROUP BY's scatter workgroups (group_part.cpp)
void staging_prime(); // llkv_hip_init: the copy lanes and the first registration of the process
// HBM → pageable host memory through the staging lanes (pinned rings, one copier thread each); the streams used are
// the lanes' own: the data must be complete on the device before the call.
int fetch_to_host(void *h_dst, const void *d_src, size_t bytes);

// Recycled pinned host memory (engine.cpp): *bytes is rounded up to the block actually handed out.
void *pinned_acquire(size_t *bytes);
void pinned_release(void *p, size_t bytes);
void pinned_release_all();
void pinned_stats(uint64_t *cached, uint64_t *outstanding); // bytes in the cache / handed out and not yet released

// Result buffers of llkv_hip_free-able arrays: pinned (recycled) blocks for large ones; result_release returns false
// for a pointer it did not hand out (a plain malloc).
void *result_acquire(size_t bytes);
bool result_release(void *p);

// Small device → host read-backs (counts, flags, a few candidate records) through a pinned buffer of the calling
// thread: a copy into pageable memory is staged by the runtime and costs ~25 µs more per call.
//   Readback rb; rb.add(&n, d_n, 8); rb.add(&flag, d_flag, 4); rc = rb.wait(stream);
struct GatherItems; // join.hpp
struct Readback {
  static constexpr size_t kBytes = 64 << 10;
  struct Item { void *dst; const void *src; size_t off, bytes; };
  Item items[12];
  int n = 0, launched = 0;
  size_t used = 0;
  uint32_t seq = 0; // what the gathering workgroup stores behind the slab when it is done (0: nobody does)
  hipStream_t stream = nullptr;
  int add(void *host_dst, const void *device_src, size_t bytes, hipStream_t s);
  // an item some kernel of the caller writes itself: *slab = where (pinned, device-visible)
  int reserve(void *host_dst, size_t bytes, hipStream_t s, void **slab);
  // hands the items not yet launched to a kernel of the caller (join.hip: readback_gather): *host = the slab's base
  int take(GatherItems *g, uint32_t **host);
  int flush(); // launches the gather of the items added so far (their device sources may be released afterwards)
  // flush + waits for the gather (polling its done word: a stream synchronisation costs ~20 µs of wake-up; the stream
  // may still be finishing the kernel's epilogue on return) + delivers the values
  int wait();
};

// Selection vector of a predicate over a table image (stream.cpp): ascending logical row ids
// and the matching device row indices.
struct Selection {
  uint64_t n = 0;
  uint64_t *d_ids = nullptr;
  uint64_t *d_dev = nullptr;
  ~Selection();
};
int run_selection(const Table *t, const llkv_filter *filters, uint32_t n_filters, const llkv_eval_op *ops,
                  uint32_t n_ops, Selection *sel, const uint32_t *drop_null_fields = nullptr, uint32_t n_drop_null_fields = 0);
// `key_set`: the bitmap an InKeySet conjunct of the plan tests (plan.hpp: lower_selection_in_set)
struct KeySetView {
  const uint64_t *bits;
  int64_t kmin;
  uint64_t span;
};
// `single_pass`: 1 = evaluate the predicate once (striped output + compaction), 0 = count pass + write pass, −1 = decide
// from a sample of the tiles: one pass moves 48 B per selected row beside the predicate columns, two passes read the
// predicate columns twice and move 16 B per selected row — one pass pays when few rows pass or the predicate is wide
// or gathers from a table
constexpr uint32_t kTileSampleStride = 64;
// `sink` (single-pass form only): the compaction of the selection also sets bit (key − kmin) of every selected row in a
// bitmap the caller zeroed (and *dup_flag when a bit was set already) — the dim table of a join built while its rows
// are being compacted, instead of one more pass over the row list
struct BitmapSink {
  const void *key_values; // key column image
  uint32_t key_width;     // 4 or 8
  uint32_t key_signed;
  long long kmin;
  unsigned long long *bits;
  uint32_t *unsorted_flag; // raised when the selected rows are not in ascending key order (equal keys included)
};
// `sync` = false: the compaction is left running on the stream (the caller keeps launching behind it on the same stream
// and synchronises before anybody else could look)
int run_selection_lowered(const Table *t, const LoweredPlan &plan, Selection *sel, const KeySetView *key_set = nullptr, int single_pass = -1,
                          const BitmapSink *sink = nullptr, bool sync = true);

int run_join(const Table *left, const Table *right, const llkv_join_key *keys, uint32_t n_keys,
             const llkv_join_options *options, llkv_on_join_batch on_batch, void *user);
int run_join_batches(const Table *left, const Table *right, const llkv_join_key *keys, uint32_t n_keys,
                     const llkv_join_options *options, const llkv_join_output *output, llkv_on_join_record_batch on_batch, void *user);
int join_output_names_c(const llkv_join_output *output, int32_t join_type, int32_t key_rules, char **names, uint32_t *n_names);

void fold_exchange_host(const uint64_t *exchange, const uint8_t *lane_ops, uint32_t lanes, uint64_t *state);
int finalize_value(const AggOut &a, const uint64_t *group_lanes, int base, llkv_value *out, std::string *err, bool prefixes_checked);

} // namespace llkv
