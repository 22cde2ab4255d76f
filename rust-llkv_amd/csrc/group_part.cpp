// group_part.cpp — partitioned GROUP BY: more dense groups than a few LDS-sized slices cover (tens of thousands …
// 2^24), every lane order-free (the shared-image lowering: integer adds, min / max, f64 sums on exact grids).
//   scatter  part_scatter_body   one workgroup per tile of 32 768 rows: rows per partition (2^shift consecutive group
//                                ids) → workgroup scan → every selected row's record (group within the partition | row
//                                within the tile, lanes 2 …) at its position inside the tile's own window of the record array, the
//                                tile's cells in partition order; the cell table [tile][partition] for the reduction
//   reduce   part_reduce_kernel  one workgroup per partition: its cell of every tile → an LDS image → rows [group][lane]
//   groups   rocPRIM select + sort  the groups that have rows, in first-appearance order (lane 1 = the smallest row id,
//                                llkv-executor/src/lib.rs:5065-5089) or key order; their lanes and decoded key cells
//   In key order over one integer key the groups of a range of partitions (one workgroup per CU) are final when that range is
//   reduced: select, emit and copy-out of the range run on a second stream beside the reduction of the next (PartGroupBy::run).
//   Records of ≤ 4 words over ≤ 512 partitions leave the scatter as whole 128-byte lines (part_scatter_body<…, LINES>).
// Against the sort-based route (group_sort.cpp) for 60 M rows in 2 M groups: no radix passes over all the rows and no
// random gathers of the argument columns — the columns are streamed (the second sweep of a tile finds them in the L2)
// and the records written and read once.  No global atomics and no global scan.
// The result is handed over like the sort-based route's (LazyGroups: cells are finalized on request).
#include "catalog.hpp"
#include "engine.hpp"
#include "join.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace llkv {

int finalize_value(const AggOut &a, const uint64_t *g, int base, llkv_value *out, std::string *err, bool prefixes_checked);

constexpr uint32_t kPartTileRowsHost = 32768; // fused_scan.hip.h: kPartTileRows
constexpr uint32_t kMaxPartsHost = 4096;      // fused_scan.hip.h: kMaxParts
constexpr size_t kPartImageBytes = 128u << 10; // LDS image of one partition, at most

struct PartGroupBy {
  const Table *table = nullptr;
  LoweredPlan plan;
  std::vector<uint32_t> key_fields;
  JitKernel kernel;
  uint32_t ngs = 0, np = 0, shift = 0; // groups per partition (2^shift), partitions
  bool order_by_keys = false;
  bool lines = false;                  // the scatter writes whole 128-byte lines
  bool ids_in_key_order = false;       // integer keys without NULL cells: ascending group ids are ascending keys
  uint32_t *d_code_rank = nullptr;     // [key][256] dictionary code → position in string order (Utf8 keys, ORDER BY the keys)
  double *d_dict_num = nullptr;
  uint8_t *d_lane_tables = nullptr; // [kl] ops of the kernel lanes, [k] source lane, [k] transform
  void *h_lanes = nullptr, *h_kv = nullptr, *h_kvalid = nullptr;
  size_t cap_lanes = 0, cap_kv = 0, cap_kvalid = 0;
  // key order over one integer key: the groups of a range of partitions are final when its reduction is — their copy-out runs
  // on a second stream beside the reduction of the next range
  static constexpr uint32_t kRanges = 8;
  hipStream_t copy_stream = nullptr;
  hipEvent_t range_done[kRanges] = {};
  uint32_t *h_counts = nullptr; // pinned: groups with rows per range, then the scatter's error word
  int run(LazyGroups *out);
  ~PartGroupBy() {
    scratch_free(d_dict_num);
    scratch_free(d_lane_tables);
    scratch_free(d_code_rank);
    if (h_lanes) (void)hipHostFree(h_lanes);
    if (h_kv) (void)hipHostFree(h_kv);
    if (h_kvalid) (void)hipHostFree(h_kvalid);
    if (h_counts) (void)hipHostFree(h_counts);
    for (hipEvent_t e : range_done) if (e) (void)hipEventDestroy(e);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
  }
};

// Threads of one scatter workgroup.  The scatter is a chain of LDS phases between barriers with the record stores at its
// end: smaller workgroups leave room for several on a CU, and one's stores drain while another ranks its rows.
uint32_t part_block_threads() {
  static const uint32_t b = [] {
    const char *e = std::getenv("LLKV_HIP_PART_BLOCK");
    const int v = e ? std::atoi(e) : 0;
    return v == 256 || v == 512 || v == 1024 ? (uint32_t)v : 1024u;
  }();
  return b;
}

void part_groupby_free(PartGroupBy *p) { delete p; }
const LoweredPlan *part_groupby_plan(const PartGroupBy *p) { return &p->plan; }

namespace {
int pinned_reserve(void **p, size_t *cap, size_t bytes) {
  if (bytes <= *cap) return LLKV_OK;
  if (*p) (void)hipHostFree(*p);
  *p = nullptr;
  *cap = 0;
  HIP_TRY(hipHostMalloc(p, bytes + bytes / 4 + 64, hipHostMallocDefault));
  *cap = bytes + bytes / 4 + 64;
  return LLKV_OK;
}
} // namespace

// Admission: what the shared-image lowering takes (statistics-bounded keys, order-free lanes) with up to 2^24 dense
// groups; in first-appearance order or in key order.  Over a sharded table every rank reduces its own rows and the
// partial groups are merged like the sort-based route's (sorted_groupby_merge: the lanes are order-free, lane 1 is a
// table-wide row id).
int part_groupby_prepare(const Table *table, const llkv_filter *filters, uint32_t n_filters, const llkv_eval_op *ops, uint32_t n_ops,
                         const uint32_t *key_fields, uint32_t n_keys, const llkv_aggregate_spec *aggs, uint32_t n_aggs, bool order_by_keys, PartGroupBy **out) {
  if (table->local_rows >= (1ull << 32)) return set_error(LLKV_UNSUPPORTED, "partitioned GROUP BY: more than 2^32 rows");
  auto resolve = [&](uint32_t fid) -> const ColumnInfo * {
    auto it = table->cols.find(fid);
    return it == table->cols.end() ? nullptr : &it->second.info;
  };
  std::unique_ptr<PartGroupBy> g(new PartGroupBy());
  g->table = table;
  g->key_fields.assign(key_fields, key_fields + n_keys);
  std::string err;
  int rc;
  if ((rc = lower_plan(resolve, filters, n_filters, ops, n_ops, key_fields, n_keys, aggs, n_aggs, /*grouped=*/true, /*track_first=*/true, &g->plan, &err,
                       /*image=*/true, /*partitioned=*/true)))
    return set_error(rc, err);
  const LoweredPlan &p = g->plan;
  const uint32_t kl = (uint32_t)p.k_image;
  // ORDER BY the keys: the dense group id orders the groups as the keys do when every key is an integer column without
  // NULL cells (the first key is the most significant digit); a NULL group's code is the key's last (NULLS FIRST wants
  // it first) and a dictionary code does not sort as its string: those groups are sorted by the id with every digit
  // replaced by its rank (hj_launch_dense_group_order_keys)
  g->order_by_keys = order_by_keys;
  g->ids_in_key_order = true;
  for (uint32_t j = 0; j < n_keys; ++j) g->ids_in_key_order &= p.key_is_int[j] && !p.key_nullable[j];
  // groups per partition: a power of two whose image fits, and small enough that the reduction has a few hundred
  // workgroups to run
  uint32_t shift = 0;
  while ((size_t)(2u << shift) * (kl - 1) * 8 <= kPartImageBytes) ++shift; // (kl − 1 cells per group: rows and first row share one)
  // (records of ≤ 4 words can leave as whole lines when the partitions number ≤ 512: 489 of them 1.16 + 0.89 ms for scatter + reduce;
  // wider records gain less from the longer runs than the reduction loses to its larger image — 977: 1.35 + 0.76, 489: 1.29 + 0.89)
  const bool line_form = kl - 1 <= 4 && part_block_threads() == 1024 && !std::getenv("LLKV_HIP_PART_NO_LINES");
  while (shift > 6 && ((uint64_t)p.ng >> shift) < (line_form ? 256u : 512u)) --shift;
  g->shift = shift;
  g->ngs = 1u << shift;
  g->np = (uint32_t)(((uint64_t)p.ng + g->ngs - 1) >> shift);
  if (g->np > kMaxPartsHost)
    return set_error(LLKV_UNSUPPORTED, "partitioned GROUP BY: " + std::to_string(p.ng) + " groups × " + std::to_string(kl) + " lanes need more than " +
                                           std::to_string(kMaxPartsHost) + " partitions");
  // records of ≤ 4 words over ≤ 512 partitions leave as whole 128-byte lines (fused_scan.hip.h: part_scatter_body<…, LINES>)
  g->lines = line_form && g->np <= 512;
  if ((rc = jit_compile(JitKind::Part, g->lines ? p.type_string + ";lines" : p.type_string, &g->kernel, &err))) return set_error(rc, err);
  hipStream_t s = g_ctx.stream;
  if (!p.dict_num.empty()) { // numeric images of the dictionaries some aggregate reads (DictNum<slot>)
    std::vector<double> image((size_t)kMaxCols * 256, 0.0);
    for (auto &d : p.dict_num) std::copy(d.second.begin(), d.second.end(), image.begin() + (size_t)d.first * 256);
    g->d_dict_num = (double *)scratch_alloc(image.size() * 8);
    if (!g->d_dict_num) return set_error(LLKV_INTERNAL, "device allocation failed");
    HIP_TRY(hipMemcpyAsync(g->d_dict_num, image.data(), image.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s)); // `image` is a local
  }
  // lane tables: ops of the kernel lanes (exchange lane j comes from kernel lane image_src[j], so that lane's op is
  // exchange lane j's), then source and transform per exchange lane
  const uint32_t k = (uint32_t)p.k;
  std::vector<uint8_t> tables(kl + 2 * k, 0);
  for (uint32_t j = 0; j < k; ++j) {
    const uint8_t src = j < p.image_src.size() ? p.image_src[j] : (uint8_t)j;
    tables[src] = p.lane_ops[j];
    tables[kl + j] = src;
    tables[kl + k + j] = j < p.image_xf.size() ? p.image_xf[j] : 0;
  }
  g->d_lane_tables = (uint8_t *)scratch_alloc(tables.size());
  if (!g->d_lane_tables) return set_error(LLKV_INTERNAL, "device allocation failed");
  HIP_TRY(hipMemcpyAsync(g->d_lane_tables, tables.data(), tables.size(), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (order_by_keys && !g->ids_in_key_order) {
    std::vector<uint32_t> ranks((size_t)n_keys * 256, 0);
    for (uint32_t j = 0; j < n_keys; ++j) {
      const ColumnInfo &ci = table->cols.at(key_fields[j]).info;
      if (ci.dtype != LLKV_DT_UTF8) continue;
      std::vector<uint32_t> idx(std::min<size_t>(ci.dictionary.size(), 256));
      for (size_t i = 0; i < idx.size(); ++i) idx[i] = (uint32_t)i;
      std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return ci.dictionary[a] < ci.dictionary[b]; });
      for (size_t r = 0; r < idx.size(); ++r) ranks[(size_t)j * 256 + idx[r]] = (uint32_t)r;
    }
    g->d_code_rank = (uint32_t *)scratch_alloc(ranks.size() * 4);
    if (!g->d_code_rank) return set_error(LLKV_INTERNAL, "device allocation failed");
    HIP_TRY(hipMemcpyAsync(g->d_code_rank, ranks.data(), ranks.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  { // admission by memory: every run keeps (k_image − 1) words for every record position of the table (SF10, 6 lanes: 2.4 GB,
    // a 4 GB block of the pool) — a table too large for that is the sort-based route's (LLKV_UNSUPPORTED: the caller falls through)
    const TileSet *ts = nullptr;
    if ((rc = get_tileset(*table, kPartTileRowsHost, &ts))) return rc;
    const uint64_t record_bytes = (uint64_t)(g->plan.k_image - 1) * ts->n_tiles * kPartTileRowsHost * 8;
    if (!scratch_can_hold(record_bytes))
      return set_error(LLKV_UNSUPPORTED, "partitioned GROUP BY: " + std::to_string(record_bytes >> 20) + " MiB of records do not fit the free HBM");
  }
  *out = g.release();
  return LLKV_OK;
}

int PartGroupBy::run(LazyGroups *out) {
  *out = LazyGroups{};
  out->active = true;
  out->plan = &plan;
  out->k = plan.k;
  out->n_keys = (uint32_t)key_fields.size();
  for (uint32_t f : key_fields) out->key_cols.push_back(&table->cols.at(f).info);
  if (plan.always_false || table->local_rows == 0) return LLKV_OK;
  hipStream_t s = g_ctx.stream;
  int rc;
  const bool trace = std::getenv("LLKV_HIP_TRACE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char *what) {
    if (!trace) return;
    (void)hipStreamSynchronize(s);
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[llkv group_part] %-22s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  const LoweredPlan &p = plan;
  const uint32_t k = (uint32_t)p.k, kl = (uint32_t)p.k_image, ng = p.ng, n_keys = (uint32_t)key_fields.size();
  const TileSet *ts = nullptr;
  if ((rc = get_tileset(*table, kPartTileRowsHost, &ts))) return rc;
  const uint32_t n_tiles = ts->n_tiles;
  const uint64_t cells = (uint64_t)n_tiles * (np + 1);
  const uint64_t cap = (uint64_t)n_tiles * kPartTileRowsHost; // tile t owns the records [t · 32 768, (t + 1) · 32 768)
  if (cap >= (1ull << 32)) return set_error(LLKV_UNSUPPORTED, "partitioned GROUP BY: more than 2^32 record positions");
  Scratch cell_table, flags, rec_val, group_rows, ids;
  if ((rc = cell_table.alloc(cells * 4)) || (rc = flags.alloc(16)) || (rc = rec_val.alloc((uint64_t)(kl - 1) * cap * 8)) || (rc = group_rows.alloc((uint64_t)ng * k * 8)) ||
      (rc = ids.alloc((uint64_t)ng * 4)))
    return rc;
  HIP_TRY(hipMemsetAsync(flags.p, 0, 16, s)); // [0] error codes, [1] number of groups
  ScanParams sp;
  std::memset(&sp, 0, sizeof sp);
  for (size_t i = 0; i < p.slot_fields.size(); ++i) sp.col[i] = slot_buffer(table->cols, p, i);
  for (size_t i = 0; i < p.lit_i.size(); ++i) sp.lit_i[i] = p.lit_i[i];
  for (size_t i = 0; i < p.lit_f.size(); ++i) sp.lit_f[i] = p.lit_f[i];
  for (size_t i = 0; i < p.key_strides.size(); ++i) sp.key_stride[i] = p.key_strides[i];
  sp.dict_num = d_dict_num;
  sp.tiles = ts->d_tiles;
  sp.n_tiles = n_tiles;
  sp.part_hist = cell_table.as<uint32_t>();
  sp.part_val = rec_val.as<uint64_t>();
  sp.part_shift = shift;
  sp.part_np = np;
  sp.part_err = flags.as<uint32_t>();
  if ((rc = jit_launch_raw(kernel.fn, n_tiles, &sp, sizeof sp, s, lines ? 1024u : part_block_threads()))) return rc;
  mark("scatter");
  DenseKeyLayout kl_keys;
  std::memset(&kl_keys, 0, sizeof kl_keys);
  kl_keys.n = n_keys;
  for (uint32_t j = 0; j < n_keys; ++j) {
    kl_keys.stride[j] = p.key_strides[j];
    kl_keys.card[j] = p.key_cards[j];
    kl_keys.nullable[j] = p.key_nullable[j];
    kl_keys.base[j] = p.key_bases[j];
    kl_keys.code_rank[j] = d_code_rank && !p.key_is_int[j] ? d_code_rank + (size_t)j * 256 : nullptr;
  }
  uint32_t n_groups = 0;
  Scratch lanes_d, kv_d, kvalid_d;
  // (a range is one workgroup per CU: smaller launches leave CUs idle — four ranges of 122 partitions: 2.46 ms for reduce + copy-out,
  // eight of 61: 3.72, against 2.77 one after the other)
  const uint32_t per_range = std::max<uint32_t>(g_ctx.cu_count, (np + kRanges - 1) / kRanges);
  const uint32_t n_ranges = (np + per_range - 1) / per_range;
  const bool ranges = order_by_keys && ids_in_key_order && n_keys == 1 && n_ranges >= 2 && (uint64_t)ng * (k * 8 + 9) <= (256ull << 20) &&
                      !std::getenv("LLKV_HIP_PART_NO_OVERLAP");
  if (ranges) {
    // ---- key order, one integer key: ascending group ids are the output order, so the groups of partitions [p0, p1) can leave
    // as soon as those partitions are reduced — select, emit and copy-out of a range run on `copy_stream` while the main
    // stream reduces the next range (the copy-out, 114 MB for 2 M groups, is the longest phase)
    if (!copy_stream) HIP_TRY(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    for (hipEvent_t &e : range_done) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (!h_counts) HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h_counts), 64, hipHostMallocDefault));
    Scratch counts_d, tmp2;
    if ((rc = counts_d.alloc(64)) || (rc = lanes_d.alloc((uint64_t)ng * k * 8)) || (rc = kv_d.alloc((uint64_t)ng * 8)) || (rc = kvalid_d.alloc(ng))) return rc;
    if ((rc = pinned_reserve(&h_lanes, &cap_lanes, (size_t)ng * k * 8)) || (rc = pinned_reserve(&h_kv, &cap_kv, (size_t)ng * 8)) || (rc = pinned_reserve(&h_kvalid, &cap_kvalid, ng)))
      return rc;
    HIP_TRY(hipMemsetAsync(counts_d.p, 0, 64, s));
    uint32_t part_at[kRanges + 1], group_at[kRanges + 1];
    for (uint32_t c = 0; c <= n_ranges; ++c) {
      part_at[c] = std::min<uint32_t>(np, c * per_range);
      group_at[c] = (uint32_t)std::min<uint64_t>((uint64_t)part_at[c] * ngs, ng);
    }
    group_at[n_ranges] = ng;
    size_t tb_max = 8;
    for (uint32_t c = 0; c < n_ranges; ++c) {
      size_t tb = 0;
      HIP_TRY(hj_select_present_groups(nullptr, &tb, group_rows.as<uint64_t>() + (uint64_t)group_at[c] * k, k, group_at[c + 1] - group_at[c], ids.as<uint32_t>() + group_at[c],
                                       counts_d.as<uint32_t>() + c, s));
      tb_max = std::max(tb_max, tb);
    }
    if ((rc = tmp2.alloc(tb_max))) return rc;
    for (uint32_t c = 0; c < n_ranges; ++c) {
      HIP_TRY(launch_part_reduce(cell_table.as<uint32_t>(), rec_val.as<uint64_t>(), ts->d_tiles, group_rows.as<uint64_t>(), d_lane_tables, d_lane_tables + kl,
                                 d_lane_tables + kl + k, n_tiles, np, ngs, ng, kl, k, s, part_at[c], part_at[c + 1] - part_at[c]));
      if (group_at[c + 1] > group_at[c]) { // ids relative to the range's first group
        size_t tb = tb_max;
        HIP_TRY(hj_select_present_groups(tmp2.p, &tb, group_rows.as<uint64_t>() + (uint64_t)group_at[c] * k, k, group_at[c + 1] - group_at[c], ids.as<uint32_t>() + group_at[c],
                                         counts_d.as<uint32_t>() + c, s));
      }
      HIP_TRY(hipMemcpyAsync(h_counts + c, counts_d.as<uint32_t>() + c, 4, hipMemcpyDeviceToHost, s));
      if (c + 1 == n_ranges) HIP_TRY(hipMemcpyAsync(h_counts + kRanges, flags.p, 4, hipMemcpyDeviceToHost, s)); // the scatter's error word
      HIP_TRY(hipEventRecord(range_done[c], s));
    }
    uint64_t at = 0;
    for (uint32_t c = 0; c < n_ranges; ++c) {
      HIP_TRY(hipEventSynchronize(range_done[c]));
      const uint32_t n_c = h_counts[c];
      if (n_c == 0) continue;
      DenseKeyLayout range_keys = kl_keys; // the ids of the range start at 0: its first group's key is the base
      range_keys.base[0] += (long long)group_at[c];
      HIP_TRY(hj_launch_emit_dense_groups(group_rows.as<uint64_t>() + (uint64_t)group_at[c] * k, k, ids.as<uint32_t>() + group_at[c], nullptr, n_c, range_keys,
                                          lanes_d.as<uint64_t>() + at * k, kv_d.as<int64_t>() + at, kvalid_d.as<uint8_t>() + at, copy_stream));
      HIP_TRY(hipMemcpyAsync(static_cast<char *>(h_lanes) + at * k * 8, lanes_d.as<uint64_t>() + at * k, (size_t)n_c * k * 8, hipMemcpyDeviceToHost, copy_stream));
      HIP_TRY(hipMemcpyAsync(static_cast<char *>(h_kv) + at * 8, kv_d.as<int64_t>() + at, (size_t)n_c * 8, hipMemcpyDeviceToHost, copy_stream));
      HIP_TRY(hipMemcpyAsync(static_cast<char *>(h_kvalid) + at, kvalid_d.as<uint8_t>() + at, n_c, hipMemcpyDeviceToHost, copy_stream));
      at += n_c;
    }
    HIP_TRY(hipStreamSynchronize(copy_stream));
    if (h_counts[kRanges]) return set_error(LLKV_INTERNAL, arith_error_message(h_counts[kRanges]));
    n_groups = (uint32_t)at;
    mark("reduce + copy out");
    if (n_groups == 0) return LLKV_OK;
  } else {
  HIP_TRY(launch_part_reduce(cell_table.as<uint32_t>(), rec_val.as<uint64_t>(), ts->d_tiles, group_rows.as<uint64_t>(), d_lane_tables, d_lane_tables + kl,
                             d_lane_tables + kl + k, n_tiles, np, ngs, ng, kl, k, s, 0, np));
  mark("partition reduce");
  // ---- the groups that have rows, in first-appearance order --------------------------------------------------------
  Scratch tmp2;
  {
    size_t tb = 0;
    HIP_TRY(hj_select_present_groups(nullptr, &tb, group_rows.as<uint64_t>(), k, ng, ids.as<uint32_t>(), flags.as<uint32_t>() + 1, s));
    if ((rc = tmp2.alloc(tb ? tb : 8))) return rc;
    HIP_TRY(hj_select_present_groups(tmp2.p, &tb, group_rows.as<uint64_t>(), k, ng, ids.as<uint32_t>(), flags.as<uint32_t>() + 1, s));
  }
  uint32_t host_flags[2] = {0, 0};
  {
    Readback rb;
    if ((rc = rb.add(host_flags, flags.p, 8, s)) || (rc = rb.wait())) return rc;
  }
  if (host_flags[0]) return set_error(LLKV_INTERNAL, arith_error_message(host_flags[0]));
  n_groups = host_flags[1];
  mark("present groups");
  if (n_groups == 0) return LLKV_OK;
  Scratch first_d, first_s, ord_in, ord_out, tmp3;
  const uint32_t *order = nullptr;
  if (n_groups > 1 && !(order_by_keys && ids_in_key_order)) {
    if ((rc = first_d.alloc((uint64_t)n_groups * 8)) || (rc = first_s.alloc((uint64_t)n_groups * 8)) || (rc = ord_in.alloc((uint64_t)n_groups * 4)) ||
        (rc = ord_out.alloc((uint64_t)n_groups * 4)))
      return rc;
    uint32_t bits = 1;
    if (order_by_keys) { // by the keys: NULLS FIRST, strings by their bytes
      HIP_TRY(hj_launch_dense_group_order_keys(ids.as<uint32_t>(), n_groups, kl_keys, first_d.as<uint64_t>(), s));
      while (bits < 64 && ((uint64_t)ng >> bits) != 0) ++bits;
    } else { // first appearance: lane 1 = the smallest row id
      HIP_TRY(hj_launch_gather_lane(group_rows.as<uint64_t>(), k, 1, ids.as<uint32_t>(), n_groups, first_d.as<uint64_t>(), s));
      while (bits < 64 && (table->total_rows >> bits) != 0) ++bits;
    }
    HIP_TRY(hj_launch_iota(ord_in.as<uint32_t>(), n_groups, s));
    size_t tb = 0;
    HIP_TRY(hj_sort_u64_u32_bits(nullptr, &tb, first_d.as<uint64_t>(), first_s.as<uint64_t>(), ord_in.as<uint32_t>(), ord_out.as<uint32_t>(), n_groups, bits, s));
    if ((rc = tmp3.alloc(tb ? tb : 8))) return rc;
    HIP_TRY(hj_sort_u64_u32_bits(tmp3.p, &tb, first_d.as<uint64_t>(), first_s.as<uint64_t>(), ord_in.as<uint32_t>(), ord_out.as<uint32_t>(), n_groups, bits, s));
    order = ord_out.as<uint32_t>();
  }
  mark("output order");
  if ((rc = lanes_d.alloc((uint64_t)n_groups * k * 8)) || (rc = kv_d.alloc((uint64_t)n_groups * n_keys * 8)) || (rc = kvalid_d.alloc((uint64_t)n_groups * n_keys))) return rc;
  HIP_TRY(hj_launch_emit_dense_groups(group_rows.as<uint64_t>(), k, ids.as<uint32_t>(), order, n_groups, kl_keys, lanes_d.as<uint64_t>(), kv_d.as<int64_t>(),
                                      kvalid_d.as<uint8_t>(), s));
  mark("emit groups");
  if ((rc = pinned_reserve(&h_lanes, &cap_lanes, (size_t)n_groups * k * 8)) || (rc = pinned_reserve(&h_kv, &cap_kv, (size_t)n_groups * n_keys * 8)) ||
      (rc = pinned_reserve(&h_kvalid, &cap_kvalid, (size_t)n_groups * n_keys)))
    return rc;
  HIP_TRY(hipMemcpyAsync(h_lanes, lanes_d.p, (size_t)n_groups * k * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(h_kv, kv_d.p, (size_t)n_groups * n_keys * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(h_kvalid, kvalid_d.p, (size_t)n_groups * n_keys, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  mark("copy out");
  }
  // only the aggregates whose finalize can fail are visited now; cells are decoded on request
  const uint64_t *lanes = static_cast<const uint64_t *>(h_lanes);
  for (size_t a = 0; a < p.aggs.size(); ++a) {
    if (p.aggs[a].fin != AggFinal::SumI64 && p.aggs[a].fin != AggFinal::AvgI64) continue;
    for (uint64_t g = 0; g < n_groups; ++g) {
      llkv_value v;
      std::string err;
      if ((rc = finalize_value(p.aggs[a], lanes + g * (size_t)k, 2, &v, &err, false))) return set_error(rc, err);
    }
  }
  out->n = n_groups;
  out->lanes = lanes;
  out->key_vals = static_cast<const int64_t *>(h_kv);
  out->key_valid = static_cast<const uint8_t *>(h_kvalid);
  mark("host checks");
  return LLKV_OK;
}

int part_groupby_run(PartGroupBy *p, LazyGroups *out) { return p->run(out); }

} // namespace llkv
