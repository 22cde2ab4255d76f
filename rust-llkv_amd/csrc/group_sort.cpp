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
#include <cstring>
#include <numeric>
#include <vector>

namespace llkv {

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return set_error(LLKV_INTERNAL, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

int finalize_value(const AggOut &a, const uint64_t *g, int base, llkv_value *out, std::string *err, bool prefixes_checked);

struct SortedGroupBy {
  const Table *table = nullptr;
  LoweredPlan sel_plan, red_plan;
  std::vector<uint32_t> key_fields;
  bool order_by_keys = false;
  JitKernel red_kernel;
  int run(std::vector<GroupResult> *groups);
};

void sorted_groupby_free(SortedGroupBy *s) { delete s; }

// Admission: GROUP BY shapes the dense kernel turned down for capacity reasons only.
int sorted_groupby_prepare(const Table *table, const llkv_filter *filters, uint32_t n_filters, const llkv_eval_op *ops, uint32_t n_ops,
                           const uint32_t *key_fields, uint32_t n_keys, const llkv_aggregate_spec *aggs, uint32_t n_aggs,
                           bool order_by_keys, SortedGroupBy **out) {
  if (table->world != 1) return set_error(LLKV_UNSUPPORTED, "sort-based GROUP BY on a sharded table");
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
  if ((rc = lower_selection(resolve, filters, n_filters, ops, n_ops, nullptr, 0, &s->sel_plan, &err))) return set_error(rc, err);
  if ((rc = lower_reduce(resolve, aggs, n_aggs, &s->red_plan, &err))) return set_error(rc, err);
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

int SortedGroupBy::run(std::vector<GroupResult> *groups) {
  groups->clear();
  hipStream_t s = g_ctx.stream;
  int rc;
  Selection sel;
  if ((rc = run_selection_lowered(table, sel_plan, &sel))) return rc;
  const uint64_t n = sel.n;
  if (n == 0) return LLKV_OK;
  if (n >= (1ull << 32)) return set_error(LLKV_UNSUPPORTED, "more than 2^32 selected rows in a sort-based GROUP BY");
  const uint32_t n_keys = (uint32_t)key_fields.size();

  // ---- sort the selection by the keys (LSD: last key first; every pass is stable) ------------------------
  Scratch perm_a, perm_b, keys_a, keys_b, vkeys_a, vkeys_b, tmp;
  if ((rc = perm_a.alloc(n * 4)) || (rc = perm_b.alloc(n * 4)) || (rc = keys_a.alloc(n * 8)) || (rc = keys_b.alloc(n * 8))) return rc;
  HIP_TRY(hj_launch_iota(perm_a.as<uint32_t>(), (uint32_t)n, s));
  uint32_t *perm = perm_a.as<uint32_t>(), *perm_other = perm_b.as<uint32_t>();
  GroupKeySet ks;
  std::memset(&ks, 0, sizeof ks);
  ks.n = n_keys;
  for (int k = (int)n_keys - 1; k >= 0; --k) {
    long long base;
    uint32_t bits;
    if ((rc = key_column_of(table, key_fields[k], &ks.k[k], &base, &bits))) return rc;
    HIP_TRY(hj_launch_gather_sort_keys(ks.k[k], base, sel.d_dev, perm, n, keys_a.as<uint64_t>(), s));
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

  // ---- group boundaries → segment starts -----------------------------------------------------------------
  Scratch flags, offs, seg;
  if ((rc = flags.alloc((n + 1) * 8)) || (rc = offs.alloc((n + 1) * 8))) return rc;
  HIP_TRY(hipMemsetAsync(flags.p, 0, (n + 1) * 8, s));
  HIP_TRY(hj_launch_group_boundaries(ks, sel.d_dev, perm, n, flags.as<uint64_t>(), s));
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

  // ---- per-group reduction ---------------------------------------------------------------------------------
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
  p.out = lanes_d.as<uint64_t>();
  p.error_flag = err_d.as<uint32_t>();
  p.n_groups = n_groups;
  const uint64_t waves_per_block = kBlock / 64;
  if ((rc = jit_launch_raw(red_kernel.fn, (uint32_t)((n_groups + waves_per_block - 1) / waves_per_block), &p, sizeof p, s))) return rc;
  HIP_TRY(hj_launch_group_keys(ks, sel.d_dev, perm, seg.as<uint64_t>(), n_groups, kv_d.as<int64_t>(), kvalid_d.as<uint8_t>(), s));

  std::vector<uint64_t> lanes(n_groups * (size_t)K);
  std::vector<int64_t> kv(n_groups * n_keys);
  std::vector<uint8_t> kvalid(n_groups * n_keys);
  uint32_t errflag = 0;
  HIP_TRY(hipMemcpyAsync(lanes.data(), lanes_d.p, lanes.size() * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(kv.data(), kv_d.p, kv.size() * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(kvalid.data(), kvalid_d.p, kvalid.size(), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&errflag, err_d.p, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (errflag) return set_error(LLKV_INTERNAL, "Arithmetic overflow: Overflow happened in a computed projection");

  // ---- host: keys, finalize, order ---------------------------------------------------------------------------
  groups->resize(n_groups);
  for (uint64_t g = 0; g < n_groups; ++g) {
    GroupResult &gr = (*groups)[g];
    const uint64_t *gl = &lanes[g * (size_t)K];
    gr.first_row = gl[1];
    gr.keys.resize(n_keys);
    for (uint32_t k = 0; k < n_keys; ++k) {
      const DeviceColumn &c = table->cols.at(key_fields[k]);
      GroupKey &gk = gr.keys[k];
      gk.is_int = c.info.dtype != LLKV_DT_UTF8;
      gk.is_null = !kvalid[(size_t)k * n_groups + g];
      if (gk.is_null) continue;
      const int64_t v = kv[(size_t)k * n_groups + g];
      if (gk.is_int) gk.i = v;
      else gk.s = (uint64_t)v < c.info.dictionary.size() ? c.info.dictionary[(size_t)v] : std::string();
    }
    gr.values.resize(rp.aggs.size());
    for (size_t a = 0; a < rp.aggs.size(); ++a) {
      std::string err;
      // a GROUP BY sum that may have overflowed in a prefix is handed back, as on the dense route
      if ((rc = finalize_value(rp.aggs[a], gl, 2, &gr.values[a], &err, false))) return set_error(rc, err);
    }
  }
  // first-appearance order (llkv-executor/src/lib.rs:5065-5089), then ORDER BY keys ASC (NULLS FIRST)
  std::sort(groups->begin(), groups->end(), [](const GroupResult &a, const GroupResult &b) { return a.first_row < b.first_row; });
  if (order_by_keys) std::stable_sort(groups->begin(), groups->end(), [](const GroupResult &a, const GroupResult &b) { return a.keys < b.keys; });
  return LLKV_OK;
}

int sorted_groupby_run(SortedGroupBy *s, std::vector<GroupResult> *groups) { return s->run(groups); }

} // namespace llkv
