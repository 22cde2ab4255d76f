// join_agg.cpp — llkv_hip_join_groupby_topk: the TPC-H Q3 shape on the GPU (see include/llkv_hip.h).
//   dim2 (customer) filter → key set            select + claim
//   dim  (orders)   filter ⋉ dim2 → hash table  select + semi flags + scan/compact + claim
//   fact (lineitem) filter ⋈ dim → (slot, value) pairs in scan order   probe-emit (count/scan/write)
//   stable sort by slot → per-group left-to-right f64 sums (the reference's order) → top-k
#include "engine.hpp"
#include "join.hpp"

#include <algorithm>
#include <cstring>
#include <vector>

namespace llkv {

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) return set_error(LLKV_INTERNAL, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

namespace {
using DB = Scratch;

int int_key_column(const Table *t, uint32_t field, JoinKeyColumn *out) {
  auto it = t->cols.find(field);
  if (it == t->cols.end()) return set_error(LLKV_NOT_FOUND, "field " + std::to_string(field) + " not found");
  if (it->second.info.nullable) return set_error(LLKV_UNSUPPORTED, "key column with NULL cells in the join-aggregate pipeline");
  out->values = it->second.d_values;
  switch (it->second.info.dtype) {
  case LLKV_DT_INT64: case LLKV_DT_UINT64: out->width = 8; out->is_signed = 1; return LLKV_OK;
  case LLKV_DT_INT32: case LLKV_DT_DATE32: out->width = 4; out->is_signed = 1; return LLKV_OK;
  case LLKV_DT_UINT32: out->width = 4; out->is_signed = 0; return LLKV_OK;
  default: return set_error(LLKV_UNSUPPORTED, std::string("integer column expected, got ") + dtype_name(it->second.info.dtype));
  }
}

struct HashSet {
  DB owner;
  uint64_t cap = 0;
  int build(const JoinKeyColumn &key, const uint64_t *d_rows, uint64_t n, bool *dup, hipStream_t s) {
    cap = 1024;
    while (cap < 2 * n) cap <<= 1;
    int rc = owner.alloc(cap * 8);
    if (rc) return rc;
    DB flag;
    if ((rc = flag.alloc(4))) return rc;
    HIP_TRY(hipMemsetAsync(owner.p, 0xFF, cap * 8, s));
    HIP_TRY(hipMemsetAsync(flag.p, 0, 4, s));
    HIP_TRY(hj_launch_claim_list(key, d_rows, n, (unsigned long long *)owner.p, cap - 1, (uint32_t *)flag.p, s));
    uint32_t f = 0;
    HIP_TRY(hipMemcpyAsync(&f, flag.p, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *dup = f != 0;
    return LLKV_OK;
  }
};
} // namespace

int run_join_groupby_topk(const llkv_join_side *fact, const llkv_join_side *dim, uint32_t dim_fk_field, const llkv_join_side *dim2,
                          const uint32_t *payload_fields, uint32_t n_payload, const llkv_expr_token *sum_expr, uint32_t sum_expr_len,
                          uint32_t limit, llkv_join_group_row *out_rows, uint32_t *out_n, uint64_t *out_total_groups) {
  int rc = ensure_device();
  if (rc) return rc;
  if (!fact || !dim || !fact->table || !dim->table || !out_rows || !out_n) return set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  if (n_payload > 4) return set_error(LLKV_UNSUPPORTED, "more than 4 payload columns");
  const Table *tf = reinterpret_cast<const Table *>(fact->table), *td = reinterpret_cast<const Table *>(dim->table);
  const Table *t2 = dim2 ? reinterpret_cast<const Table *>(dim2->table) : nullptr;
  hipStream_t s = g_ctx.stream;
  *out_n = 0;
  if (out_total_groups) *out_total_groups = 0;

  // ---- dim2 key set -------------------------------------------------------------------
  HashSet set2;
  JoinKeyColumn k2{}, fk{};
  if (t2) {
    Selection sel2;
    if ((rc = run_selection(t2, dim2->filters, dim2->n_filters, nullptr, 0, &sel2))) return rc;
    if ((rc = int_key_column(t2, dim2->key_field, &k2)) || (rc = int_key_column(td, dim_fk_field, &fk))) return rc;
    bool dup = false;
    if ((rc = set2.build(k2, sel2.d_dev, sel2.n, &dup, s))) return rc;
  }
  // ---- dim rows: filter [⋉ dim2] ------------------------------------------------------
  Selection seld;
  if ((rc = run_selection(td, dim->filters, dim->n_filters, nullptr, 0, &seld))) return rc;
  DB kept; // device rows of the qualifying dim rows
  uint64_t n_dim = seld.n;
  const uint64_t *d_dim_rows = seld.d_dev;
  if (t2 && seld.n) {
    DB flags, offs, tmp;
    if ((rc = flags.alloc(seld.n * 8)) || (rc = offs.alloc((seld.n + 1) * 8))) return rc;
    HIP_TRY(hj_launch_semi_flags(fk, seld.d_dev, seld.n, k2, (const unsigned long long *)set2.owner.p, set2.cap - 1, (uint64_t *)flags.p, s));
    size_t tb = 0;
    HIP_TRY(hj_exclusive_scan_u64(nullptr, &tb, (const uint64_t *)flags.p, (uint64_t *)offs.p, seld.n, s));
    if ((rc = tmp.alloc(tb))) return rc;
    HIP_TRY(hj_exclusive_scan_u64(tmp.p, &tb, (const uint64_t *)flags.p, (uint64_t *)offs.p, seld.n, s));
    uint64_t last_off = 0, last_flag = 0;
    HIP_TRY(hipMemcpyAsync(&last_off, (uint64_t *)offs.p + seld.n - 1, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&last_flag, (uint64_t *)flags.p + seld.n - 1, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    n_dim = last_off + last_flag;
    if ((rc = kept.alloc(n_dim * 8))) return rc;
    HIP_TRY(hj_launch_compact(seld.d_dev, (const uint64_t *)flags.p, (const uint64_t *)offs.p, seld.n, (uint64_t *)kept.p, s));
    HIP_TRY(hipStreamSynchronize(s));
    d_dim_rows = (const uint64_t *)kept.p;
  }
  if (n_dim == 0) return LLKV_OK;
  // ---- dim hash table -------------------------------------------------------------------
  JoinKeyColumn kd{};
  if ((rc = int_key_column(td, dim->key_field, &kd))) return rc;
  HashSet ht;
  bool dup = false;
  if ((rc = ht.build(kd, d_dim_rows, n_dim, &dup, s))) return rc;
  if (dup) return set_error(LLKV_UNSUPPORTED, "dimension key is not unique: groups are not identified by the dim row");
  if (ht.cap >= (1ull << 32)) return set_error(LLKV_UNSUPPORTED, "dimension too large");

  // ---- fact probe-emit -------------------------------------------------------------------
  auto resolve = [&](uint32_t fid) -> const ColumnInfo * {
    auto it = tf->cols.find(fid);
    return it == tf->cols.end() ? nullptr : &it->second.info;
  };
  LoweredPlan plan;
  std::string err;
  if ((rc = lower_probe(resolve, fact->filters, fact->n_filters, fact->key_field, sum_expr, sum_expr_len, &plan, &err))) return set_error(rc, err);
  if (plan.always_false || tf->local_rows == 0) return LLKV_OK;
  JitKernel k;
  if ((rc = jit_compile(JitKind::Probe, plan.type_string, &k, &err))) return set_error(rc, err);
  const TileSet *ts = nullptr;
  if ((rc = get_tileset(*tf, 8192, &ts))) return rc;
  const uint32_t n_slots = ts->n_tiles * (kBlock / 64);
  DB counts, offsets;
  if ((rc = counts.alloc((size_t)n_slots * 8)) || (rc = offsets.alloc((size_t)(n_slots + 1) * 8))) return rc;
  ScanParams p;
  std::memset(&p, 0, sizeof p);
  for (size_t i = 0; i < plan.slot_fields.size(); ++i) p.col[i] = slot_buffer(tf->cols, plan, i);
  for (size_t i = 0; i < plan.lit_i.size(); ++i) p.lit_i[i] = plan.lit_i[i];
  for (size_t i = 0; i < plan.lit_f.size(); ++i) p.lit_f[i] = plan.lit_f[i];
  p.tiles = ts->d_tiles;
  p.n_tiles = ts->n_tiles;
  p.sub_rows = 8192 / (kBlock / 64);
  p.tile_partials = (uint64_t *)counts.p;
  p.ht_owner = (const unsigned long long *)ht.owner.p;
  p.ht_mask = ht.cap - 1;
  p.ht_keys = kd.values;
  p.ht_key_width = kd.width;
  p.ht_key_signed = kd.is_signed;
  if ((rc = jit_launch_raw(k.fn, ts->n_tiles, &p, sizeof p, s))) return rc;
  {
    DB tmp;
    size_t tb = 0;
    // offsets[n_slots] (the total) is not produced by an exclusive scan of n_slots entries: scan n_slots + 1
    // entries whose last count is 0
    DB counts1;
    if ((rc = counts1.alloc((size_t)(n_slots + 1) * 8))) return rc;
    HIP_TRY(hipMemsetAsync(counts1.p, 0, (size_t)(n_slots + 1) * 8, s));
    HIP_TRY(hipMemcpyAsync(counts1.p, counts.p, (size_t)n_slots * 8, hipMemcpyDeviceToDevice, s));
    HIP_TRY(hj_exclusive_scan_u64(nullptr, &tb, (const uint64_t *)counts1.p, (uint64_t *)offsets.p, n_slots + 1, s));
    if ((rc = tmp.alloc(tb))) return rc;
    HIP_TRY(hj_exclusive_scan_u64(tmp.p, &tb, (const uint64_t *)counts1.p, (uint64_t *)offsets.p, n_slots + 1, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  uint64_t n_pairs = 0;
  HIP_TRY(hipMemcpy(&n_pairs, (uint64_t *)offsets.p + n_slots, 8, hipMemcpyDeviceToHost));
  if (n_pairs >= kPredErrorBit) return set_error(LLKV_INTERNAL, "Arithmetic overflow: Overflow happened in a comparison");
  if (n_pairs == 0) return LLKV_OK;
  DB e_slot, e_val, s_slot, s_val;
  if ((rc = e_slot.alloc(n_pairs * 4)) || (rc = e_val.alloc(n_pairs * 8)) || (rc = s_slot.alloc(n_pairs * 4)) || (rc = s_val.alloc(n_pairs * 8))) return rc;
  p.aux_in = (const uint64_t *)offsets.p;
  p.aux_out32 = (uint32_t *)e_slot.p;
  p.aux_out = (uint64_t *)e_val.p;
  if ((rc = jit_launch_raw(k.fn2, ts->n_tiles, &p, sizeof p, s))) return rc;
  // ---- stable sort by slot, per-group sums in scan order ---------------------------------------
  uint32_t bits = 1;
  while ((1ull << bits) < ht.cap) ++bits;
  {
    DB tmp;
    size_t tb = 0;
    HIP_TRY(hj_sort_u32_u64(nullptr, &tb, (const uint32_t *)e_slot.p, (uint32_t *)s_slot.p, (const uint64_t *)e_val.p, (uint64_t *)s_val.p, n_pairs, bits, s));
    if ((rc = tmp.alloc(tb))) return rc;
    HIP_TRY(hj_sort_u32_u64(tmp.p, &tb, (const uint32_t *)e_slot.p, (uint32_t *)s_slot.p, (const uint64_t *)e_val.p, (uint64_t *)s_val.p, n_pairs, bits, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  DB sums, cnts, tk_keys, tk_slots, tk_keys_s, tk_slots_s, n_groups_d;
  if ((rc = sums.alloc(ht.cap * 8)) || (rc = cnts.alloc(ht.cap * 8)) || (rc = tk_keys.alloc(ht.cap * 8)) || (rc = tk_slots.alloc(ht.cap * 4)) ||
      (rc = tk_keys_s.alloc(ht.cap * 8)) || (rc = tk_slots_s.alloc(ht.cap * 4)) || (rc = n_groups_d.alloc(8)))
    return rc;
  HIP_TRY(hipMemsetAsync(cnts.p, 0, ht.cap * 8, s));
  HIP_TRY(hipMemsetAsync(n_groups_d.p, 0, 8, s));
  HIP_TRY(hj_launch_segment_sums((const uint32_t *)s_slot.p, (const uint64_t *)s_val.p, n_pairs, (double *)sums.p, (uint64_t *)cnts.p, s));
  HIP_TRY(hj_launch_topk_keys((const double *)sums.p, (const uint64_t *)cnts.p, ht.cap, (uint64_t *)tk_keys.p, (uint32_t *)tk_slots.p,
                              (unsigned long long *)n_groups_d.p, s));
  {
    DB tmp;
    size_t tb = 0;
    HIP_TRY(hj_sort_u64_u32(nullptr, &tb, (const uint64_t *)tk_keys.p, (uint64_t *)tk_keys_s.p, (const uint32_t *)tk_slots.p, (uint32_t *)tk_slots_s.p, ht.cap, s));
    if ((rc = tmp.alloc(tb))) return rc;
    HIP_TRY(hj_sort_u64_u32(tmp.p, &tb, (const uint64_t *)tk_keys.p, (uint64_t *)tk_keys_s.p, (const uint32_t *)tk_slots.p, (uint32_t *)tk_slots_s.p, ht.cap, s));
  }
  // ---- candidates → host: the first (limit + slack) groups by descending sum, gathered in ONE kernel + ONE copy ----
  const uint32_t want = (uint32_t)std::min<uint64_t>(ht.cap, (uint64_t)limit + 64);
  CandidateCols cc;
  std::memset(&cc, 0, sizeof cc);
  cc.key = kd;
  cc.n_payload = n_payload;
  for (uint32_t i = 0; i < n_payload; ++i) if ((rc = int_key_column(td, payload_fields[i], &cc.payload[i]))) return rc;
  DB cand_d;
  if ((rc = cand_d.alloc((size_t)want * 64 + 8))) return rc;
  HIP_TRY(hj_launch_gather_candidates((const uint64_t *)tk_keys_s.p, (const uint32_t *)tk_slots_s.p, want, (const unsigned long long *)ht.owner.p,
                                      (const double *)sums.p, (const uint64_t *)cnts.p, cc, (uint64_t *)cand_d.p, s));
  std::vector<uint64_t> hc((size_t)want * 8);
  uint64_t n_groups = 0;
  HIP_TRY(hipMemcpyAsync(hc.data(), cand_d.p, (size_t)want * 64, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(&n_groups, n_groups_d.p, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  uint32_t n_cand = 0;
  while (n_cand < want && hc[(size_t)n_cand * 8] != ~0ull) ++n_cand;
  if (n_cand == want && (uint64_t)want < ht.cap && n_cand > limit && limit > 0 && hc[(size_t)(limit - 1) * 8] == hc[(size_t)(want - 1) * 8])
    return set_error(LLKV_UNSUPPORTED, "more than 64 groups tie on the LIMIT boundary");
  std::vector<llkv_join_group_row> cand(n_cand);
  for (uint32_t i = 0; i < n_cand; ++i) {
    const uint64_t *c = &hc[(size_t)i * 8];
    llkv_join_group_row &g = cand[i];
    g.key = (int64_t)c[1];
    std::memcpy(&g.sum, &c[2], 8);
    g.count = c[3];
    for (int k = 0; k < 4; ++k) g.payload[k] = (int64_t)c[4 + k];
  }
  // ORDER BY sum DESC, payload[0] ASC (arrow lexsort, llkv-executor/src/lib.rs:13847-13864); LIMIT
  std::stable_sort(cand.begin(), cand.end(), [&](const llkv_join_group_row &a, const llkv_join_group_row &b) {
    if (a.sum != b.sum) return a.sum > b.sum;
    return n_payload ? a.payload[0] < b.payload[0] : false;
  });
  const uint32_t n = std::min<uint32_t>(limit, n_cand);
  for (uint32_t i = 0; i < n; ++i) out_rows[i] = cand[i];
  *out_n = n;
  if (out_total_groups) *out_total_groups = n_groups;
  return LLKV_OK;
}

} // namespace llkv

extern "C" llkv_status llkv_hip_join_groupby_topk(const llkv_join_side *fact, const llkv_join_side *dim, uint32_t dim_fk_field,
                                                  const llkv_join_side *dim2, const uint32_t *payload_fields, uint32_t n_payload,
                                                  const llkv_expr_token *sum_expr, uint32_t sum_expr_len, uint32_t limit,
                                                  llkv_join_group_row *out_rows, uint32_t *out_n, uint64_t *out_total_groups) {
  return (llkv_status)llkv::run_join_groupby_topk(fact, dim, dim_fk_field, dim2, payload_fields, n_payload, sum_expr, sum_expr_len, limit,
                                                  out_rows, out_n, out_total_groups);
}
