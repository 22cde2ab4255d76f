// group_sort.cpp — sort-based GROUP BY: the route for any number of groups and any state width
// (execute_group_by_single_table llkv-executor/src/lib.rs:4405-4542 hashes a per-row key Vec and copies the
// filtered table; here: no hash table, no copy).
//   predicate        → selection vector (row order)                         select_body (count / scan / write)
//   keys             → stable LSD radix sorts of the selection, last key first, each limited to the bits the
//                      column's statistics leave (rocPRIM); NULL cells by one more 1-bit pass (NULLS FIRST)
//   group boundaries → flags → scan → segment starts; the key cells of each group's first row
//   aggregates       → group_reduce_body: one wave per group over its segment, the lane groups and the host
//                      finalize of the dense kernel (rows, first row id, SUM/AVG/MIN/MAX/COUNT lanes …)
// The sort is stable, so rows keep their order inside a group and a group's first row is its first appearance.
// Output order: first appearance (the reference's), or ascending keys with NULLS FIRST.
#include "engine.hpp"
#include "join.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <unordered_map>
#include <vector>

namespace llkv {

int finalize_value(const AggOut &a, const uint64_t *g, int base, llkv_value *out, std::string *err, bool prefixes_checked);

struct SortedGroupBy {
  const Table *table = nullptr;
  LoweredPlan sel_plan, red_plan;
  std::vector<uint32_t> key_fields;
  bool order_by_keys = false;
  JitKernel red_kernel;
  // result arrays of the latest execution, pinned host memory reused across executions
  void *h_lanes = nullptr, *h_kv = nullptr, *h_kvalid = nullptr;
  size_t cap_lanes = 0, cap_kv = 0, cap_kvalid = 0;
  // table-wide groups after sorted_groupby_merge (a sharded table: every rank ran its own rows)
  std::vector<uint64_t> m_lanes;
  std::vector<int64_t> m_kv;
  std::vector<uint8_t> m_kvalid;
  PartGroupBy *part = nullptr; // the partitioned route answers instead (group_part.cpp)
  // join → GROUP BY (join_group.cpp): one more conjunct of the selection — the integer column `key_set_field` must be in the
  // key set of the dimension rows that qualify (a bitmap the caller owns, alive as long as this object)
  bool has_key_set = false;
  KeySetView key_set{nullptr, 0, 0};
  int run(LazyGroups *out);
  ~SortedGroupBy() {
    if (part) part_groupby_free(part);
    if (h_lanes) (void)hipHostFree(h_lanes);
    if (h_kv) (void)hipHostFree(h_kv);
    if (h_kvalid) (void)hipHostFree(h_kvalid);
  }
};

void sorted_groupby_free(SortedGroupBy *s) { delete s; }
bool sorted_groupby_partitioned(const SortedGroupBy *s) { return s && s->part; }

// ---- sharded tables: merge of the ranks' partial groups (host) -------------------------------------------------------
// Every rank ran the query over its own chunks: its groups are partial states (the lanes of the reduce plan), keyed
// by the raw key cells (integers; Utf8 as codes of the table-wide dictionary).  The ranks' states of one key are
// combined lane by lane IN RANK ORDER — rank order is row order, so "first row" minima and the order of f64 partial
// sums are those of the table — and the groups are put into the order one device would have produced: by key (NULLS
// first, strings by dictionary order) or by first appearance.
namespace {
inline uint64_t combine_lane(int op, uint64_t a, uint64_t b) { // fused_scan.hip.h lane ops
  switch (op) {
  case 0: { double x, y; std::memcpy(&x, &a, 8); std::memcpy(&y, &b, 8); const double z = x + y; uint64_t r; std::memcpy(&r, &z, 8); return r; }
  case 1: return a + b;
  case 2: return (int64_t)b < (int64_t)a ? b : a;
  case 3: return (int64_t)b > (int64_t)a ? b : a;
  default: return b > a ? b : a;
  }
}
struct KeyTupleHost {
  int64_t v[4];
  uint8_t valid[4];
  bool operator==(const KeyTupleHost &o) const { return std::memcmp(v, o.v, sizeof v) == 0 && std::memcmp(valid, o.valid, sizeof valid) == 0; }
};
struct KeyTupleHash {
  size_t operator()(const KeyTupleHost &k) const {
    uint64_t h = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 4; ++i) { h ^= (uint64_t)k.v[i] + k.valid[i]; h *= 0xff51afd7ed558ccdull; h ^= h >> 32; }
    return (size_t)h;
  }
};
} // namespace

int sorted_groupby_merge(SortedGroupBy *s, uint32_t world, const uint64_t *rank_groups, const int64_t *const *key_values,
                         const uint8_t *const *key_valid, const uint64_t *const *lanes, LazyGroups *out) {
  const uint32_t n_keys = (uint32_t)s->key_fields.size();
  const LoweredPlan &lane_plan = s->part ? *part_groupby_plan(s->part) : s->red_plan; // (lane_ops: the first k entries are one group's)
  const int K = lane_plan.k;
  const std::vector<uint8_t> &ops = lane_plan.lane_ops;
  std::unordered_map<KeyTupleHost, uint64_t, KeyTupleHash> index;
  std::vector<KeyTupleHost> keys;
  std::vector<uint64_t> state;
  for (uint32_t r = 0; r < world; ++r) {
    const uint64_t n = rank_groups[r];
    if (n && (!key_values[r] || !key_valid[r] || !lanes[r])) return set_error(LLKV_INVALID_ARGUMENT, "partial groups of a rank are missing");
    for (uint64_t g = 0; g < n; ++g) {
      KeyTupleHost kt;
      std::memset(&kt, 0, sizeof kt);
      for (uint32_t k = 0; k < n_keys; ++k) {
        kt.valid[k] = key_valid[r][(size_t)k * n + g] ? 1 : 0;
        kt.v[k] = kt.valid[k] ? key_values[r][(size_t)k * n + g] : 0;
      }
      const uint64_t *src = lanes[r] + (size_t)g * K;
      auto it = index.find(kt);
      if (it == index.end()) {
        index.emplace(kt, keys.size());
        keys.push_back(kt);
        state.insert(state.end(), src, src + K);
      } else {
        uint64_t *dst = state.data() + (size_t)it->second * K;
        for (int l = 0; l < K; ++l) dst[l] = combine_lane(ops[(size_t)l], dst[l], src[l]);
      }
    }
  }
  const uint64_t n = keys.size();
  std::vector<uint64_t> order(n);
  std::iota(order.begin(), order.end(), 0ull);
  if (s->order_by_keys) {
    std::vector<std::vector<uint32_t>> rank_of(n_keys); // Utf8: dictionary code → position in string order
    for (uint32_t k = 0; k < n_keys; ++k) {
      const ColumnInfo &ci = s->table->cols.at(s->key_fields[k]).info;
      if (ci.dtype != LLKV_DT_UTF8) continue;
      std::vector<uint32_t> idx(ci.dictionary.size());
      std::iota(idx.begin(), idx.end(), 0u);
      std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return ci.dictionary[a] < ci.dictionary[b]; });
      rank_of[k].resize(idx.size());
      for (size_t i = 0; i < idx.size(); ++i) rank_of[k][idx[i]] = (uint32_t)i;
    }
    std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) {
      for (uint32_t k = 0; k < n_keys; ++k) {
        const KeyTupleHost &x = keys[a], &y = keys[b];
        if (x.valid[k] != y.valid[k]) return x.valid[k] < y.valid[k]; // NULLS FIRST
        if (!x.valid[k]) continue;
        int64_t xv = x.v[k], yv = y.v[k];
        if (!rank_of[k].empty()) { xv = (uint64_t)xv < rank_of[k].size() ? rank_of[k][(size_t)xv] : xv; yv = (uint64_t)yv < rank_of[k].size() ? rank_of[k][(size_t)yv] : yv; }
        if (xv != yv) return xv < yv;
      }
      return false;
    });
  } else { // first appearance (llkv-executor/src/lib.rs:5065-5089): lane 1 = row id of the group's first row
    std::sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return state[(size_t)a * K + 1] < state[(size_t)b * K + 1]; });
  }
  s->m_lanes.resize((size_t)n * K);
  s->m_kv.resize((size_t)n * n_keys);
  s->m_kvalid.resize((size_t)n * n_keys);
  for (uint64_t i = 0; i < n; ++i) {
    const uint64_t g = order[i];
    std::memcpy(s->m_lanes.data() + (size_t)i * K, state.data() + (size_t)g * K, (size_t)K * 8);
    for (uint32_t k = 0; k < n_keys; ++k) {
      s->m_kv[(size_t)k * n + i] = keys[g].v[k];
      s->m_kvalid[(size_t)k * n + i] = keys[g].valid[k];
    }
  }
  out->n = n;
  out->lanes = s->m_lanes.data();
  out->key_vals = s->m_kv.data();
  out->key_valid = s->m_kvalid.data();
  return LLKV_OK;
}

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

// Admission: GROUP BY shapes the dense kernel turned down for capacity reasons only.
int sorted_groupby_prepare(const Table *table, const llkv_filter *filters, uint32_t n_filters, const llkv_eval_op *ops, uint32_t n_ops,
                           const uint32_t *key_fields, uint32_t n_keys, const llkv_aggregate_spec *aggs, uint32_t n_aggs,
                           bool order_by_keys, SortedGroupBy **out, const KeySetView *key_set, uint32_t key_set_field) {
  if (n_keys == 0 || n_keys > 4) return set_error(LLKV_UNSUPPORTED, "sort-based GROUP BY takes 1..4 keys");
  auto resolve = [&](uint32_t fid) -> const ColumnInfo * {
    auto it = table->cols.find(fid);
    return it == table->cols.end() ? nullptr : &it->second.info;
  };
  for (uint32_t k = 0; k < n_keys; ++k) {
    const ColumnInfo *ci = resolve(key_fields[k]);
    if (!ci) return set_error(LLKV_INVALID_ARGUMENT, "column '" + std::to_string(key_fields[k]) + "' not found in GROUP BY input");
    switch (ci->dtype) {
    case LLKV_DT_INT64: case LLKV_DT_INT32: case LLKV_DT_DATE32: case LLKV_DT_UINT32: case LLKV_DT_UINT64: case LLKV_DT_UTF8: break;
    case LLKV_DT_FLOAT64: case LLKV_DT_FLOAT32: case LLKV_DT_DECIMAL128:
      return set_error(LLKV_INVALID_ARGUMENT, std::string("GROUP BY does not support column type ") + dtype_name(ci->dtype));
    default: return set_error(LLKV_UNSUPPORTED, std::string("GROUP BY over ") + dtype_name(ci->dtype));
    }
  }
  std::unique_ptr<SortedGroupBy> s(new SortedGroupBy());
  s->table = table;
  s->order_by_keys = order_by_keys;
  s->key_fields.assign(key_fields, key_fields + n_keys);
  std::string err;
  int rc;
  // statistics-bounded keys and order-free lanes (what the shared-image lowering takes): the partitioned route — no sort
  // of the rows, no gathers
  if (!key_set && !std::getenv("LLKV_HIP_GROUP_NO_IMAGE") && !std::getenv("LLKV_HIP_GROUP_NO_PART") &&
      part_groupby_prepare(table, filters, n_filters, ops, n_ops, key_fields, n_keys, aggs, n_aggs, order_by_keys, &s->part) == LLKV_OK) {
    *out = s.release();
    return LLKV_OK;
  }
  if (key_set) {
    if (n_ops) return set_error(LLKV_UNSUPPORTED, "a key-set conjunct beside a predicate program");
    s->has_key_set = true;
    s->key_set = *key_set;
    if ((rc = lower_selection_in_set(resolve, filters, n_filters, key_set_field, &s->sel_plan, &err))) return set_error(rc, err);
  } else if ((rc = lower_selection(resolve, filters, n_filters, ops, n_ops, nullptr, 0, &s->sel_plan, &err))) return set_error(rc, err);
  if ((rc = lower_reduce(resolve, aggs, n_aggs, &s->red_plan, &err))) return set_error(rc, err);
  if (s->red_plan.has_distinct() && table->world != 1)
    return set_error(LLKV_UNSUPPORTED, "DISTINCT aggregates inside GROUP BY over a sharded table (the ranks' partial groups cannot be merged)");
  if ((rc = jit_compile(JitKind::Reduce, s->red_plan.type_string, &s->red_kernel, &err))) return set_error(rc, err);
  *out = s.release();
  return LLKV_OK;
}

namespace {
int key_column_of(const Table *t, uint32_t field, JoinKeyColumn *kc, long long *base, uint32_t *bits) {
  const DeviceColumn &c = t->cols.at(field);
  std::memset(kc, 0, sizeof *kc);
  kc->values = c.d_values;
  kc->valid = c.info.nullable ? c.d_valid : nullptr;
  *bits = 64;
  switch (c.info.dtype) {
  case LLKV_DT_INT64: kc->width = 8; kc->is_signed = 1; *base = INT64_MIN; break;
  case LLKV_DT_UINT64: kc->width = 8; kc->is_signed = 0; *base = 0; break;
  case LLKV_DT_INT32: case LLKV_DT_DATE32: kc->width = 4; kc->is_signed = 1; *base = INT32_MIN; *bits = 32; break;
  case LLKV_DT_UINT32: kc->width = 4; kc->is_signed = 0; *base = 0; *bits = 32; break;
  default: kc->width = 1; kc->is_signed = 0; *base = 0; *bits = 8; break; // dictionary codes
  }
  if (c.info.has_stats && kc->is_signed) { // only the bits the value range needs are sorted
    *base = c.info.min_i;
    const unsigned __int128 range = (unsigned __int128)((__int128)c.info.max_i - (__int128)c.info.min_i);
    uint32_t b = 1;
    while (b < 64 && (range >> b) != 0) ++b;
    *bits = b;
  }
  return LLKV_OK;
}
} // namespace

int SortedGroupBy::run(LazyGroups *out) {
  if (part) return part_groupby_run(part, out);
  *out = LazyGroups{};
  out->active = true;
  out->plan = &red_plan;
  out->k = red_plan.k;
  out->n_keys = (uint32_t)key_fields.size();
  for (uint32_t f : key_fields) out->key_cols.push_back(&table->cols.at(f).info);
  hipStream_t s = g_ctx.stream;
  int rc;
  // LLKV_HIP_TRACE=1: phase times on stderr (each mark synchronizes the stream)
  const bool trace = std::getenv("LLKV_HIP_TRACE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char *what) {
    if (!trace) return;
    (void)hipStreamSynchronize(s);
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[llkv group_sort] %-22s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  Selection sel;
  if ((rc = run_selection_lowered(table, sel_plan, &sel, has_key_set ? &key_set : nullptr))) return rc;
  const uint64_t n = sel.n;
  mark("selection");
  if (n == 0) return LLKV_OK;
  if (n >= (1ull << 32)) return set_error(LLKV_UNSUPPORTED, "more than 2^32 selected rows in a sort-based GROUP BY");
  const uint32_t n_keys = (uint32_t)key_fields.size();

  // ---- sort the selection by the keys (LSD: last key first; every pass is stable) ------------------------
  Scratch perm_a, perm_b, keys_a, keys_b, vkeys_a, vkeys_b, tmp, rank_d;
  if ((rc = perm_a.alloc(n * 4)) || (rc = perm_b.alloc(n * 4)) || (rc = keys_a.alloc(n * 8)) || (rc = keys_b.alloc(n * 8))) return rc;
  HIP_TRY(hj_launch_iota(perm_a.as<uint32_t>(), (uint32_t)n, s));
  uint32_t *perm = perm_a.as<uint32_t>(), *perm_other = perm_b.as<uint32_t>();
  GroupKeySet ks;
  std::memset(&ks, 0, sizeof ks);
  ks.n = n_keys;
  // DISTINCT aggregates: their column is the least significant sort key (NULL cells first, then by value), so that
  // inside a group equal values are neighbours
  const bool has_distinct = red_plan.has_distinct();
  JoinKeyColumn dcol;
  std::memset(&dcol, 0, sizeof dcol);
  Scratch ddict; // numeric image of the dictionary a DISTINCT aggregate is over
  // where the DISTINCT argument's cell of selection entry j lies: its device row in the column — or j itself in the values a computed
  // argument was evaluated to (one Int64 / Float64 per selected row, PlanValue semantics: the group's temp column)
  const uint64_t *drows = sel.d_dev;
  Scratch dvalues, dvalid_bits, dvalid_bytes, diota, derr;
  if (has_distinct && red_plan.distinct_proj) {
    const LoweredPlan &proj = *red_plan.distinct_proj;
    JitKernel pk;
    std::string perr;
    if ((rc = jit_compile(JitKind::Project, proj.type_string, &pk, &perr))) return set_error(rc, perr);
    const bool nullable = !proj.out_nullable.empty() && proj.out_nullable[0];
    if ((rc = dvalues.alloc(n * 8)) || (rc = diota.alloc(n * 8)) || (rc = derr.alloc(4)) || (nullable && ((rc = dvalid_bits.alloc((n + 63) / 64 * 8 + 8)) || (rc = dvalid_bytes.alloc(n))))) return rc;
    ProjParams q;
    std::memset(&q, 0, sizeof q);
    for (size_t sl = 0; sl < proj.slot_fields.size(); ++sl) q.col[sl] = slot_buffer(table->cols, proj, sl);
    for (size_t i = 0; i < proj.lit_i.size(); ++i) q.lit_i[i] = proj.lit_i[i];
    for (size_t i = 0; i < proj.lit_f.size(); ++i) q.lit_f[i] = proj.lit_f[i];
    q.dev_rows = sel.d_dev;
    q.n = (uint32_t)n;
    q.out[0] = dvalues.p;
    q.out_valid[0] = nullable ? dvalid_bits.as<uint64_t>() : nullptr;
    q.error_flag = derr.as<uint32_t>();
    q.error_stride = 0;
    HIP_TRY(hipMemsetAsync(derr.p, 0, 4, s));
    if ((rc = jit_launch_raw(pk.fn, (uint32_t)((n + kBlock - 1) / kBlock), &q, sizeof q, s))) return rc;
    if (nullable) HIP_TRY(hj_launch_bits_to_bytes(dvalid_bits.as<uint64_t>(), n, dvalid_bytes.as<uint8_t>(), s));
    HIP_TRY(hj_launch_iota_u64(diota.as<uint64_t>(), n, 0, s));
    uint32_t perr_bits = 0;
    Readback rb;
    if ((rc = rb.add(&perr_bits, derr.p, 4, s)) || (rc = rb.wait())) return rc;
    if (perr_bits) return set_error(LLKV_INTERNAL, "Arithmetic overflow in a DISTINCT aggregate's argument");
    dcol.values = dvalues.p;
    dcol.valid = nullable ? dvalid_bytes.as<uint8_t>() : nullptr;
    dcol.width = 8;
    dcol.is_signed = 0;
    drows = diota.as<uint64_t>();
  } else if (has_distinct) {
    const DeviceColumn &dc = table->cols.at((uint32_t)red_plan.distinct_field);
    dcol.values = dc.d_values;
    dcol.valid = dc.info.nullable ? dc.d_valid : nullptr;
    dcol.width = dc.info.dtype == LLKV_DT_DATE32 ? 4 : (dc.info.dtype == LLKV_DT_INT64 || dc.info.dtype == LLKV_DT_FLOAT64 || dc.info.dtype == LLKV_DT_DECIMAL128) ? 8 : 1; // (the 64-bit image of a decimal; dictionary codes and Booleans: a byte)
    dcol.is_signed = dc.info.dtype == LLKV_DT_DATE32; // otherwise only equality matters: the cell's pattern (Float64: "by bit pattern", llkv-aggregate/src/lib.rs:252-331)
  }
  if (has_distinct) {
    if (red_plan.distinct_numeric == 1) {
      if ((rc = ddict.alloc(256 * 8))) return rc;
      HIP_TRY(hipMemcpyAsync(ddict.p, red_plan.distinct_dict_num.data(), 256 * 8, hipMemcpyHostToDevice, s)); // (the plan outlives the run)
    }
    HIP_TRY(hj_launch_gather_sort_keys(dcol, 0, nullptr, drows, perm, n, keys_a.as<uint64_t>(), s));
    size_t tb = 0;
    const uint32_t dbits = dcol.width == 1 ? 8 : 64;
    HIP_TRY(hj_sort_u64_u32_bits(nullptr, &tb, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), perm, perm_other, n, dbits, s));
    if ((rc = tmp.alloc(tb ? tb : 8))) return rc;
    HIP_TRY(hj_sort_u64_u32_bits(tmp.p, &tb, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), perm, perm_other, n, dbits, s));
    std::swap(perm, perm_other);
    if (dcol.valid) {
      if ((rc = vkeys_a.alloc(n * 4)) || (rc = vkeys_b.alloc(n * 4))) return rc;
      HIP_TRY(hj_launch_gather_valid(dcol, drows, perm, n, vkeys_a.as<uint32_t>(), s));
      size_t vb = 0;
      HIP_TRY(hj_sort_by_slot(nullptr, &vb, vkeys_a.as<uint32_t>(), vkeys_b.as<uint32_t>(), perm, perm_other, (uint32_t)n, 1, s));
      HIP_TRY(hipStreamSynchronize(s)); // tmp is in use by the sort before
      if ((rc = tmp.alloc(vb ? vb : 8))) return rc;
      HIP_TRY(hj_sort_by_slot(tmp.p, &vb, vkeys_a.as<uint32_t>(), vkeys_b.as<uint32_t>(), perm, perm_other, (uint32_t)n, 1, s));
      std::swap(perm, perm_other);
    }
    HIP_TRY(hipStreamSynchronize(s)); // tmp is reallocated by the next pass
  }
  for (int k = (int)n_keys - 1; k >= 0; --k) {
    long long base;
    uint32_t bits;
    if ((rc = key_column_of(table, key_fields[k], &ks.k[k], &base, &bits))) return rc;
    const uint8_t *code_rank = nullptr;
    const ColumnInfo &ci = table->cols.at(key_fields[k]).info;
    if (order_by_keys && ci.dtype == LLKV_DT_UTF8) { // ORDER BY the key: codes sort as their strings do
      std::vector<uint32_t> idx(ci.dictionary.size());
      std::iota(idx.begin(), idx.end(), 0u);
      std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return ci.dictionary[a] < ci.dictionary[b]; });
      uint8_t rank[256] = {0};
      for (size_t r = 0; r < idx.size(); ++r) rank[idx[r]] = (uint8_t)r;
      if ((rc = rank_d.alloc(256))) return rc;
      HIP_TRY(hipMemcpyAsync(rank_d.p, rank, 256, hipMemcpyHostToDevice, s));
      HIP_TRY(hipStreamSynchronize(s)); // `rank` is a stack array
      code_rank = rank_d.as<uint8_t>();
    }
    HIP_TRY(hj_launch_gather_sort_keys(ks.k[k], base, code_rank, sel.d_dev, perm, n, keys_a.as<uint64_t>(), s));
    if (n_keys == 1 && !ks.k[k].valid && !has_distinct) {
      // GROUP BY the column the table is clustered by (a primary-key order): the selection already is in key order —
      // no sort, and the reduction then streams the argument columns instead of gathering them
      Scratch unsorted;
      uint32_t is_unsorted = 0;
      Readback rb;
      if ((rc = unsorted.alloc(4))) return rc;
      HIP_TRY(hipMemsetAsync(unsorted.p, 0, 4, s));
      HIP_TRY(hj_launch_unsorted_flag(keys_a.as<uint64_t>(), n, unsorted.as<uint32_t>(), s));
      if ((rc = rb.add(&is_unsorted, unsorted.p, 4, s)) || (rc = rb.wait())) return rc;
      if (!is_unsorted && !std::getenv("LLKV_HIP_GROUP_ALWAYS_SORT")) {
        std::swap(keys_a.p, keys_b.p); // the boundary pass reads the "sorted" images from keys_b
        break;
      }
    }
    size_t tb = 0;
    HIP_TRY(hj_sort_u64_u32_bits(nullptr, &tb, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), perm, perm_other, n, bits, s));
    if ((rc = tmp.alloc(tb ? tb : 8))) return rc;
    HIP_TRY(hj_sort_u64_u32_bits(tmp.p, &tb, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), perm, perm_other, n, bits, s));
    std::swap(perm, perm_other);
    if (ks.k[k].valid) { // NULL cells (key image 0) are told apart — and put first — by one more 1-bit pass
      if (!vkeys_a.p && ((rc = vkeys_a.alloc(n * 4)) || (rc = vkeys_b.alloc(n * 4)))) return rc;
      HIP_TRY(hj_launch_gather_valid(ks.k[k], sel.d_dev, perm, n, vkeys_a.as<uint32_t>(), s));
      size_t vb = 0;
      HIP_TRY(hj_sort_by_slot(nullptr, &vb, vkeys_a.as<uint32_t>(), vkeys_b.as<uint32_t>(), perm, perm_other, (uint32_t)n, 1, s));
      HIP_TRY(hipStreamSynchronize(s)); // tmp may be in use by the previous sort
      if ((rc = tmp.alloc(vb ? vb : 8))) return rc;
      HIP_TRY(hj_sort_by_slot(tmp.p, &vb, vkeys_a.as<uint32_t>(), vkeys_b.as<uint32_t>(), perm, perm_other, (uint32_t)n, 1, s));
      std::swap(perm, perm_other);
    }
    HIP_TRY(hipStreamSynchronize(s)); // tmp is reallocated by the next pass
  }

  mark("key sorts");
  // ---- group boundaries → segment starts -----------------------------------------------------------------
  Scratch flags, offs, seg;
  if ((rc = flags.alloc((n + 1) * 8)) || (rc = offs.alloc((n + 1) * 8))) return rc;
  HIP_TRY(hipMemsetAsync(flags.p, 0, (n + 1) * 8, s));
  // one NULL-free key: the sorted key images of the last pass are the keys — a streaming compare instead of two
  // random gathers per position
  if (n_keys == 1 && !ks.k[0].valid) HIP_TRY(hj_launch_run_heads(keys_b.as<uint64_t>(), n, flags.as<uint64_t>(), s));
  else HIP_TRY(hj_launch_group_boundaries(ks, sel.d_dev, perm, n, flags.as<uint64_t>(), s));
  {
    size_t tb = 0;
    HIP_TRY(hj_exclusive_scan_u64(nullptr, &tb, flags.as<uint64_t>(), offs.as<uint64_t>(), n + 1, s));
    if ((rc = tmp.alloc(tb ? tb : 8))) return rc;
    HIP_TRY(hj_exclusive_scan_u64(tmp.p, &tb, flags.as<uint64_t>(), offs.as<uint64_t>(), n + 1, s));
  }
  uint64_t n_groups = 0;
  HIP_TRY(hipMemcpyAsync(&n_groups, offs.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if ((rc = seg.alloc((n_groups + 1) * 8))) return rc;
  HIP_TRY(hj_launch_segment_starts(flags.as<uint64_t>(), offs.as<uint64_t>(), n, n_groups, seg.as<uint64_t>(), s));
  mark("boundaries");

  // ---- output order -----------------------------------------------------------------------------------------
  // ORDER BY the keys: the segments already are in key order (NULLS FIRST).  Otherwise first appearance
  // (llkv-executor/src/lib.rs:5065-5089): sort the segments by the row id of their first row.
  Scratch first_d, first_s, ord_in, ord_out;
  const uint32_t *order = nullptr;
  if (has_distinct) { // the rows of a group are sorted by the argument too: its first appearance is its smallest row id
    if ((rc = first_d.alloc(n_groups * 8))) return rc;
    HIP_TRY(hj_launch_segment_min_rows(sel.d_ids, perm, seg.as<uint64_t>(), n_groups, first_d.as<uint64_t>(), s));
  }
  if (!order_by_keys && n_groups > 1) {
    if ((!has_distinct && (rc = first_d.alloc(n_groups * 8))) || (rc = first_s.alloc(n_groups * 8)) || (rc = ord_in.alloc(n_groups * 4)) || (rc = ord_out.alloc(n_groups * 4))) return rc;
    if (!has_distinct) HIP_TRY(hj_launch_first_rows(sel.d_ids, perm, seg.as<uint64_t>(), n_groups, first_d.as<uint64_t>(), s));
    HIP_TRY(hj_launch_iota(ord_in.as<uint32_t>(), (uint32_t)n_groups, s));
    uint32_t bits = 1;
    while (bits < 64 && (table->total_rows >> bits) != 0) ++bits;
    size_t tb = 0;
    HIP_TRY(hj_sort_u64_u32_bits(nullptr, &tb, first_d.as<uint64_t>(), first_s.as<uint64_t>(), ord_in.as<uint32_t>(), ord_out.as<uint32_t>(), n_groups, bits, s));
    HIP_TRY(hipStreamSynchronize(s));
    if ((rc = tmp.alloc(tb ? tb : 8))) return rc;
    HIP_TRY(hj_sort_u64_u32_bits(tmp.p, &tb, first_d.as<uint64_t>(), first_s.as<uint64_t>(), ord_in.as<uint32_t>(), ord_out.as<uint32_t>(), n_groups, bits, s));
    order = ord_out.as<uint32_t>();
  }
  mark("output order");

  // ---- per-group reduction, written in output order ------------------------------------------------------------
  const LoweredPlan &rp = red_plan;
  const int K = rp.k;
  Scratch lanes_d, err_d, kv_d, kvalid_d;
  if ((rc = lanes_d.alloc(n_groups * (size_t)K * 8)) || (rc = err_d.alloc(4)) || (rc = kv_d.alloc(n_groups * n_keys * 8)) || (rc = kvalid_d.alloc(n_groups * n_keys)))
    return rc;
  HIP_TRY(hipMemsetAsync(err_d.p, 0, 4, s));
  ReduceParams p;
  std::memset(&p, 0, sizeof p);
  for (size_t i = 0; i < rp.slot_fields.size(); ++i) p.col[i] = slot_buffer(table->cols, rp, i);
  for (size_t i = 0; i < rp.lit_i.size(); ++i) p.lit_i[i] = rp.lit_i[i];
  for (size_t i = 0; i < rp.lit_f.size(); ++i) p.lit_f[i] = rp.lit_f[i];
  p.perm = perm;
  p.dev_rows = sel.d_dev;
  p.row_ids = sel.d_ids;
  p.seg_start = seg.as<uint64_t>();
  p.order = order;
  p.out = lanes_d.as<uint64_t>();
  p.error_flag = err_d.as<uint32_t>();
  p.n_groups = n_groups;
  Scratch dval, dhead;
  if (has_distinct) {
    if ((rc = dval.alloc(n * 8)) || (rc = dhead.alloc(n))) return rc;
    HIP_TRY(hj_launch_distinct_heads(dcol, red_plan.distinct_numeric, ddict.as<double>(), drows, perm, flags.as<uint64_t>(), n, dval.as<uint64_t>(), dhead.as<uint8_t>(), s));
    p.dval = dval.as<uint64_t>();
    p.dhead = dhead.as<uint8_t>();
    p.first_rows = first_d.as<uint64_t>();
  }
  // lanes per group: a wave, or 8 lanes when the groups average fewer than 16 rows
  const bool narrow = n / n_groups < 16;
  const uint64_t groups_per_block = kBlock / (narrow ? 8 : 64);
  if ((rc = jit_launch_raw(narrow ? red_kernel.fn2 : red_kernel.fn, (uint32_t)((n_groups + groups_per_block - 1) / groups_per_block), &p, sizeof p, s))) return rc;
  HIP_TRY(hj_launch_group_keys(ks, sel.d_dev, perm, seg.as<uint64_t>(), order, n_groups, kv_d.as<int64_t>(), kvalid_d.as<uint8_t>(), s));
  mark("group reduce");

  if ((rc = pinned_reserve(&h_lanes, &cap_lanes, n_groups * (size_t)K * 8)) || (rc = pinned_reserve(&h_kv, &cap_kv, n_groups * n_keys * 8)) ||
      (rc = pinned_reserve(&h_kvalid, &cap_kvalid, n_groups * n_keys)))
    return rc;
  uint32_t errflag = 0;
  HIP_TRY(hipMemcpyAsync(h_lanes, lanes_d.p, n_groups * (size_t)K * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(h_kv, kv_d.p, n_groups * n_keys * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(h_kvalid, kvalid_d.p, n_groups * n_keys, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&errflag, err_d.p, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (errflag) return set_error(LLKV_INTERNAL, arith_error_message(errflag));
  mark("copy out");

  // ---- host: only the aggregates whose finalize can fail are visited now; cells are decoded on request ----------
  const uint64_t *lanes = static_cast<const uint64_t *>(h_lanes);
  for (size_t a = 0; a < rp.aggs.size(); ++a) {
    if (rp.aggs[a].fin != AggFinal::SumI64 && rp.aggs[a].fin != AggFinal::AvgI64) continue;
    for (uint64_t g = 0; g < n_groups; ++g) {
      llkv_value v;
      std::string err;
      // a GROUP BY sum that may have overflowed in a prefix is handed back, as on the dense route
      if ((rc = finalize_value(rp.aggs[a], lanes + g * (size_t)K, 2, &v, &err, false))) return set_error(rc, err);
    }
  }
  out->n = n_groups;
  out->lanes = lanes;
  out->key_vals = static_cast<const int64_t *>(h_kv);
  out->key_valid = static_cast<const uint8_t *>(h_kvalid);
  mark("host checks");
  return LLKV_OK;
}

int sorted_groupby_run(SortedGroupBy *s, LazyGroups *out) { return s->run(out); }

} // namespace llkv
