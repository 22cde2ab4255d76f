// join_group.cpp — join → GROUP BY with ANY aggregate list (include/llkv_hip.h: llkv_hip_join_groupby_prepare / _rows).
//
// The executor's multi-table route aggregates whatever the SELECT lists over the joined batches (try_execute_hash_join
// llkv-executor/src/lib.rs:3780-4052 → execute_group_by_from_batches :4544-4755, ORDER BY :13762-13868, LIMIT :10925-10955).
// llkv_hip_join_groupby_topk (join_agg.cpp) is the hand-tuned form of ONE shape — a single SUM, ORDER BY sum DESC, payload[0];
// this file is the general one, for the same star shape   fact ⋈ dim [⋉ dim2]  GROUP BY dim.key [, payload …]:
//   COUNT(*) / COUNT / SUM / TOTAL / AVG / MIN / MAX over fact-side expressions, several of them, any ORDER BY over the
//   aggregates, the payload and the key, LIMIT or none.
// A unique dimension key makes "the group of a joined row" the fact row's own key, so the join dissolves into
//   1. the key set of the qualifying dimension rows  — dim2's selection → bitmap; dim's selection (AndThen<filters, InKeySet<fk>>)
//                                                      → bitmap of its keys (a key that occurs twice is refused)
//   2. GROUP BY the fact key over the fact rows that pass   filters ∧ InKeySet<fact key>   — the sort-based GROUP BY
//      (group_sort.cpp) with one more conjunct: every accumulator, the PlanValue argument semantics (decimals included), NULL
//      handling and the sharded merge of partial groups are its own; a fact table clustered by the key needs no sort
//   3. per result group: bisection of its key in the qualifying dimension rows sorted by key → payload cells and the row's
//      position (the last tie-break of the order), ORDER BY / LIMIT on the host over the finalized cells.
// No joined row ever exists: the fact columns are read once (predicate columns, then the arguments of the rows that join).
#include "engine.hpp"
#include "join.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

namespace llkv {

int finalize_value(const AggOut &a, const uint64_t *g, int base, llkv_value *out, std::string *err, bool prefixes_checked);

namespace {
constexpr uint64_t kMaxSetSpan = 1ull << 32; // 512 MiB of bitmap at most

int key_col_of(const Table *t, uint32_t field, JoinKeyColumn *out, const ColumnInfo **info, bool allow_null) {
  auto it = t->cols.find(field);
  if (it == t->cols.end()) return set_error(LLKV_NOT_FOUND, "field " + std::to_string(field) + " not found");
  const ColumnInfo &ci = it->second.info;
  if (ci.nullable && !allow_null) return set_error(LLKV_UNSUPPORTED, "join key column with NULL cells in the join → GROUP BY pipeline");
  std::memset(out, 0, sizeof *out);
  out->values = it->second.d_values;
  out->valid = ci.nullable ? it->second.d_valid : nullptr;
  switch (ci.dtype) {
  case LLKV_DT_INT64: out->width = 8; out->is_signed = 1; break;
  case LLKV_DT_INT32: case LLKV_DT_DATE32: out->width = 4; out->is_signed = 1; break;
  case LLKV_DT_UINT32: out->width = 4; out->is_signed = 0; break;
  default: return set_error(LLKV_UNSUPPORTED, std::string("integer column expected in the join → GROUP BY pipeline, got ") + dtype_name(ci.dtype));
  }
  if (info) *info = &ci;
  return LLKV_OK;
}

struct KeyBitmap {
  Scratch bits;
  uint32_t *flag_p = nullptr;
  int64_t kmin = 0;
  uint64_t span = 0;
  KeySetView view() const { return KeySetView{static_cast<const uint64_t *>(bits.p), kmin, span}; }
  // bit (key − min) for every listed row; *dup is set when two rows carried one key
  int build(const ColumnInfo &ci, const JoinKeyColumn &key, const uint64_t *d_rows, uint64_t n, bool *dup, hipStream_t s) {
    if (!ci.has_stats || ci.max_i < ci.min_i || (uint64_t)ci.max_i - (uint64_t)ci.min_i >= kMaxSetSpan)
      return set_error(LLKV_UNSUPPORTED, "join → GROUP BY: the key column has no statistics-bounded range for a direct key set");
    kmin = ci.min_i;
    span = (uint64_t)ci.max_i - (uint64_t)ci.min_i;
    const uint64_t n_words = span / 64 + 1;
    int rc = bits.alloc(n_words * 8 + 32);
    if (rc) return rc;
    flag_p = reinterpret_cast<uint32_t *>(static_cast<char *>(bits.p) + n_words * 8);
    HIP_TRY(hj_launch_fill(bits.p, n_words * 8 + 32, 0, s));
    HIP_TRY(hj_launch_bitmap_build(key, d_rows, n, kmin, (unsigned long long *)bits.p, flag_p, s));
    uint32_t f = 0;
    HIP_TRY(hipMemcpyAsync(&f, flag_p, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *dup = f != 0;
    return LLKV_OK;
  }
};
} // namespace

struct JoinGroupState {
  const Table *td = nullptr;
  KeyBitmap set2, set;
  // the qualifying dimension rows sorted by key image (value − key_base): keys, device rows, positions in row order
  Scratch keys, rows, pos;
  uint64_t n_dim = 0;
  long long key_base = 0;
  uint32_t dim_key_field = 0;
};
void join_group_state_free(JoinGroupState *s) { delete s; }

struct JoinRows {
  uint32_t n_aggs = 0, n_payload = 0;
  uint64_t total_groups = 0;
  std::vector<int64_t> keys;
  std::vector<int64_t> payload;      // [n][n_payload]
  std::vector<uint8_t> payload_null; // [n][n_payload]
  std::vector<uint64_t> group_index;
  std::vector<llkv_value> values;    // [n][n_aggs]
  size_t n() const { return keys.size(); }
};

static int join_groupby_prepare(const llkv_join_side *fact, const llkv_join_side *dim, uint32_t dim_fk_field, const llkv_join_side *dim2,
                                const llkv_aggregate_spec *aggs, uint32_t n_aggs, Query **out) {
  int rc = ensure_device();
  if (rc) return rc;
  if (!fact || !dim || !fact->table || !dim->table || !out || (n_aggs && !aggs)) return set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  if (n_aggs == 0) return set_error(LLKV_INVALID_ARGUMENT, "aggregate query requires at least one aggregate expression");
  const Table *tf = reinterpret_cast<const Table *>(fact->table), *td = reinterpret_cast<const Table *>(dim->table);
  const Table *t2 = dim2 ? reinterpret_cast<const Table *>(dim2->table) : nullptr;
  if (td->world != 1 || (t2 && t2->world != 1))
    return set_error(LLKV_INVALID_ARGUMENT, "dimension tables are replicated: stage them whole (world = 1) on every rank; only the fact table is sharded");
  hipStream_t s = g_ctx.stream;
  std::unique_ptr<JoinGroupState, void (*)(JoinGroupState *)> st(new JoinGroupState(), join_group_state_free);
  st->td = td;
  st->dim_key_field = dim->key_field;
  std::string err;

  // ---- 1. the dimension rows that qualify: filters [⋉ dim2] -------------------------------------------------------------------
  Selection seld;
  if (t2) {
    JoinKeyColumn k2;
    const ColumnInfo *k2i = nullptr;
    if ((rc = key_col_of(t2, dim2->key_field, &k2, &k2i, false))) return rc;
    Selection sel2;
    if ((rc = run_selection(t2, dim2->filters, dim2->n_filters, nullptr, 0, &sel2))) return rc;
    bool dup2 = false;
    if ((rc = st->set2.build(*k2i, k2, sel2.d_dev, sel2.n, &dup2, s))) return rc; // (a key twice in dim2 changes nothing for a semi join)
    auto resolve_d = [&](uint32_t fid) -> const ColumnInfo * {
      auto it = td->cols.find(fid);
      return it == td->cols.end() ? nullptr : &it->second.info;
    };
    LoweredPlan dplan;
    if ((rc = lower_selection_in_set(resolve_d, dim->filters, dim->n_filters, dim_fk_field, &dplan, &err))) return set_error(rc, err);
    const KeySetView v2 = st->set2.view();
    if ((rc = run_selection_lowered(td, dplan, &seld, &v2))) return rc;
  } else if ((rc = run_selection(td, dim->filters, dim->n_filters, nullptr, 0, &seld))) {
    return rc;
  }
  JoinKeyColumn kd;
  const ColumnInfo *kdi = nullptr;
  if ((rc = key_col_of(td, dim->key_field, &kd, &kdi, false))) return rc;
  bool dup = false;
  if ((rc = st->set.build(*kdi, kd, seld.d_dev, seld.n, &dup, s))) return rc;
  if (dup) return set_error(LLKV_UNSUPPORTED, "join → GROUP BY: the dimension key is not unique among the qualifying rows (a group would not be a dimension row)");

  // ---- … sorted by key, for the payload lookups of the result groups ----------------------------------------------------------
  st->n_dim = seld.n;
  st->key_base = kdi->min_i;
  if (seld.n >= (1ull << 32)) return set_error(LLKV_UNSUPPORTED, "more than 2^32 qualifying dimension rows");
  if (seld.n) {
    Scratch perm, keys_in, perm_out, tmp;
    if ((rc = perm.alloc(seld.n * 4)) || (rc = keys_in.alloc(seld.n * 8)) || (rc = st->keys.alloc(seld.n * 8)) || (rc = st->rows.alloc(seld.n * 8)) ||
        (rc = st->pos.alloc(seld.n * 4)))
      return rc;
    HIP_TRY(hj_launch_iota(perm.as<uint32_t>(), (uint32_t)seld.n, s));
    if (kdi->ascending) { // a selection of a strictly ascending column is in key order already
      HIP_TRY(hj_launch_gather_sort_keys(kd, st->key_base, nullptr, seld.d_dev, perm.as<uint32_t>(), seld.n, st->keys.as<uint64_t>(), s));
      HIP_TRY(hipMemcpyAsync(st->pos.p, perm.p, seld.n * 4, hipMemcpyDeviceToDevice, s));
    } else {
      HIP_TRY(hj_launch_gather_sort_keys(kd, st->key_base, nullptr, seld.d_dev, perm.as<uint32_t>(), seld.n, keys_in.as<uint64_t>(), s));
      size_t tb = 0;
      HIP_TRY(hj_sort_u64_u32(nullptr, &tb, keys_in.as<uint64_t>(), st->keys.as<uint64_t>(), perm.as<uint32_t>(), st->pos.as<uint32_t>(), seld.n, s));
      if ((rc = tmp.alloc(tb ? tb : 8))) return rc;
      HIP_TRY(hj_sort_u64_u32(tmp.p, &tb, keys_in.as<uint64_t>(), st->keys.as<uint64_t>(), perm.as<uint32_t>(), st->pos.as<uint32_t>(), seld.n, s));
    }
    HIP_TRY(hj_launch_gather_u64(seld.d_dev, st->pos.as<uint32_t>(), seld.n, st->rows.as<uint64_t>(), s));
    HIP_TRY(hipStreamSynchronize(s)); // (the temporaries above are released on return)
  }

  // ---- 2. GROUP BY the fact key over   filters ∧ InKeySet<fact key>   ---------------------------------------------------------
  JoinKeyColumn kf;
  if ((rc = key_col_of(tf, fact->key_field, &kf, nullptr, false))) return rc;
  std::unique_ptr<Query> q(new Query());
  q->table = tf;
  q->table_generation = tf->generation;
  q->order_by_keys = true; // ascending keys: the same order on every rank count (the caller's ORDER BY is applied by _rows)
  q->n_user_aggs = n_aggs;
  q->n_user_keys = 1;
  const KeySetView view = st->set.view();
  const uint32_t key_field = fact->key_field;
  if ((rc = sorted_groupby_prepare(tf, fact->filters, fact->n_filters, nullptr, 0, &key_field, 1, aggs, n_aggs, true, &q->sorted, &view, key_field))) return rc;
  q->route_note = "join → GROUP BY: key set of " + std::to_string(st->n_dim) + " qualifying dimension rows, sort-based GROUP BY of the fact key";
  q->join_state = st.release();
  *out = q.release();
  return LLKV_OK;
}

namespace {
// One ORDER BY key of a result row as something comparable: NULL flag + (integer | double).
struct OrderCell {
  bool null = false, is_f = false;
  __int128 i = 0;
  double f = 0;
};
int cmp_cells(const OrderCell &a, const OrderCell &b, bool nulls_first) {
  if (a.null || b.null) return a.null == b.null ? 0 : ((a.null == nulls_first) ? -1 : 1);
  if (a.is_f || b.is_f) {
    // arrow's sort orders floats by totalOrder: NaN above every number (llkv-executor/src/lib.rs:13847-13864 → lexsort_to_indices)
    const double x = a.is_f ? a.f : (double)a.i, y = b.is_f ? b.f : (double)b.i;
    const bool xn = x != x, yn = y != y;
    if (xn || yn) return xn == yn ? 0 : (xn ? 1 : -1);
    return x < y ? -1 : x > y ? 1 : 0;
  }
  return a.i < b.i ? -1 : a.i > b.i ? 1 : 0;
}
OrderCell cell_of_value(const llkv_value &v) {
  OrderCell c;
  c.null = v.is_null != 0;
  if (v.dtype == LLKV_DT_FLOAT64) { c.is_f = true; c.f = v.f64; }
  else if (v.dtype == LLKV_DT_DECIMAL128) c.i = (__int128)(((unsigned __int128)(uint64_t)v.i64_hi << 64) | (unsigned __int128)(uint64_t)v.i64);
  else c.i = v.i64;
  return c;
}
} // namespace

static int join_groupby_rows(Query *q, const uint32_t *payload_fields, uint32_t n_payload, const llkv_join_order_key *order, uint32_t n_order,
                             uint64_t limit, JoinRows **out) {
  int rc = ensure_device();
  if (rc) return rc;
  if (!q || !q->join_state || !q->sorted || !out) return set_error(LLKV_INVALID_ARGUMENT, "not a join → GROUP BY query");
  if (!q->lazy.active) return set_error(LLKV_INVALID_ARGUMENT, "the query has not finished an execution yet");
  if (n_payload > 4) return set_error(LLKV_UNSUPPORTED, "more than 4 payload columns");
  const JoinGroupState &st = *q->join_state;
  const LazyGroups &lz = q->lazy;
  const uint32_t n_aggs = q->n_user_aggs;
  for (uint32_t o = 0; o < n_order; ++o) {
    const llkv_join_order_key &k = order[o];
    if ((k.kind == LLKV_JOIN_ORDER_AGGREGATE && k.index >= n_aggs) || (k.kind == LLKV_JOIN_ORDER_PAYLOAD && k.index >= n_payload) ||
        (k.kind != LLKV_JOIN_ORDER_AGGREGATE && k.kind != LLKV_JOIN_ORDER_PAYLOAD && k.kind != LLKV_JOIN_ORDER_KEY))
      return set_error(LLKV_INVALID_ARGUMENT, "ORDER BY key out of range");
  }
  const uint64_t n = lz.n;
  std::unique_ptr<JoinRows> res(new JoinRows());
  res->n_aggs = n_aggs;
  res->n_payload = n_payload;
  res->total_groups = n;
  if (n == 0 || limit == 0) { *out = res.release(); return LLKV_OK; }
  hipStream_t s = g_ctx.stream;

  // ---- 3a. every group's dimension row: payload cells + position among the qualifying rows -----------------------------------
  JoinPayloadCols pc;
  std::memset(&pc, 0, sizeof pc);
  pc.n = n_payload;
  for (uint32_t c = 0; c < n_payload; ++c)
    if ((rc = key_col_of(st.td, payload_fields[c], &pc.col[c], nullptr, true))) return rc;
  std::vector<int64_t> h_payload((size_t)n_payload * n);
  std::vector<uint8_t> h_pvalid((size_t)n_payload * n);
  std::vector<uint32_t> h_pos(n);
  {
    Scratch d_keys, d_payload, d_valid, d_pos;
    if ((rc = d_keys.alloc(n * 8)) || (rc = d_payload.alloc((size_t)(n_payload ? n_payload : 1) * n * 8)) || (rc = d_valid.alloc((size_t)(n_payload ? n_payload : 1) * n)) ||
        (rc = d_pos.alloc(n * 4)))
      return rc;
    HIP_TRY(hipMemcpyAsync(d_keys.p, lz.key_vals, n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hj_launch_lookup_payload(st.keys.as<uint64_t>(), st.rows.as<uint64_t>(), st.pos.as<uint32_t>(), st.n_dim, d_keys.as<int64_t>(), st.key_base, n, pc,
                                     d_payload.as<int64_t>(), d_valid.as<uint8_t>(), d_pos.as<uint32_t>(), s));
    if (n_payload) {
      HIP_TRY(hipMemcpyAsync(h_payload.data(), d_payload.p, (size_t)n_payload * n * 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(h_pvalid.data(), d_valid.p, (size_t)n_payload * n, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipMemcpyAsync(h_pos.data(), d_pos.p, n * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  for (uint64_t g = 0; g < n; ++g)
    if (h_pos[g] == 0xFFFFFFFFu) return set_error(LLKV_INTERNAL, "join → GROUP BY: a result group's key is not among the qualifying dimension rows");

  // ---- 3b. ORDER BY … LIMIT over the finalized cells (ties: dimension row order, so every rank count agrees) -----------------
  // only the aggregates an ORDER BY key names are finalized for every group; the rest for the rows that are delivered
  std::vector<std::vector<OrderCell>> cells(n_order);
  std::string err;
  for (uint32_t o = 0; o < n_order; ++o) {
    cells[o].resize(n);
    const llkv_join_order_key &k = order[o];
    for (uint64_t g = 0; g < n; ++g) {
      OrderCell c;
      if (k.kind == LLKV_JOIN_ORDER_KEY) c.i = lz.key_vals[g];
      else if (k.kind == LLKV_JOIN_ORDER_PAYLOAD) { c.null = !h_pvalid[(size_t)k.index * n + g]; c.i = h_payload[(size_t)k.index * n + g]; }
      else {
        llkv_value v;
        if ((rc = finalize_value(lz.plan->aggs[k.index], lz.lanes + (size_t)g * lz.k, 2, &v, &err, false))) return set_error(rc, err);
        c = cell_of_value(v);
      }
      cells[o][g] = c;
    }
  }
  auto before = [&](uint64_t a, uint64_t b) {
    for (uint32_t o = 0; o < n_order; ++o) {
      int c = cmp_cells(cells[o][a], cells[o][b], order[o].nulls_first != 0);
      if (c) return order[o].descending ? c > 0 : c < 0;
    }
    return h_pos[a] < h_pos[b];
  };
  std::vector<uint64_t> idx(n);
  std::iota(idx.begin(), idx.end(), 0ull);
  const uint64_t take = limit > n ? n : limit; // (UINT64_MAX: every group)
  if (take < n) std::partial_sort(idx.begin(), idx.begin() + (long)take, idx.end(), before);
  else std::sort(idx.begin(), idx.end(), before);
  idx.resize(take);

  res->keys.resize(take);
  res->group_index.resize(take);
  res->payload.resize((size_t)take * n_payload);
  res->payload_null.resize((size_t)take * n_payload);
  res->values.resize((size_t)take * n_aggs);
  for (uint64_t r = 0; r < take; ++r) {
    const uint64_t g = idx[r];
    res->keys[r] = lz.key_vals[g];
    res->group_index[r] = h_pos[g];
    for (uint32_t c = 0; c < n_payload; ++c) {
      res->payload[(size_t)r * n_payload + c] = h_payload[(size_t)c * n + g];
      res->payload_null[(size_t)r * n_payload + c] = h_pvalid[(size_t)c * n + g] ? 0 : 1;
    }
    for (uint32_t a = 0; a < n_aggs; ++a)
      if ((rc = finalize_value(lz.plan->aggs[a], lz.lanes + (size_t)g * lz.k, 2, &res->values[(size_t)r * n_aggs + a], &err, false))) return set_error(rc, err);
  }
  *out = res.release();
  return LLKV_OK;
}

} // namespace llkv

using namespace llkv;

extern "C" {

llkv_status llkv_hip_join_groupby_prepare(const llkv_join_side *fact, const llkv_join_side *dim, uint32_t dim_fk_field, const llkv_join_side *dim2,
                                          const llkv_aggregate_spec *aggs, uint32_t n_aggs, llkv_hip_query **out) {
  Query *q = nullptr;
  const int rc = join_groupby_prepare(fact, dim, dim_fk_field, dim2, aggs, n_aggs, &q);
  if (rc) return (llkv_status)rc;
  *out = reinterpret_cast<llkv_hip_query *>(q);
  return LLKV_OK;
}

llkv_status llkv_hip_join_groupby_rows(llkv_hip_query *query, const uint32_t *payload_fields, uint32_t n_payload, const llkv_join_order_key *order,
                                       uint32_t n_order, uint64_t limit, llkv_hip_join_rows **out) {
  if (n_payload && !payload_fields) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL payload field list");
  if (n_order && !order) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL ORDER BY list");
  JoinRows *r = nullptr;
  const int rc = join_groupby_rows(reinterpret_cast<Query *>(query), payload_fields, n_payload, order, n_order, limit, &r);
  if (rc) return (llkv_status)rc;
  *out = reinterpret_cast<llkv_hip_join_rows *>(r);
  return LLKV_OK;
}

uint64_t llkv_hip_join_rows_len(const llkv_hip_join_rows *rows) { return rows ? reinterpret_cast<const JoinRows *>(rows)->n() : 0; }
uint64_t llkv_hip_join_rows_total_groups(const llkv_hip_join_rows *rows) { return rows ? reinterpret_cast<const JoinRows *>(rows)->total_groups : 0; }

llkv_status llkv_hip_join_rows_get(const llkv_hip_join_rows *rows, uint64_t i, int64_t *key, int64_t *payload, uint8_t *payload_is_null,
                                   uint64_t *group_index, const llkv_value **values) {
  const JoinRows *r = reinterpret_cast<const JoinRows *>(rows);
  if (!r || i >= r->n()) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "row index out of range");
  if (key) *key = r->keys[i];
  if (group_index) *group_index = r->group_index[i];
  for (uint32_t c = 0; c < r->n_payload; ++c) {
    if (payload) payload[c] = r->payload[(size_t)i * r->n_payload + c];
    if (payload_is_null) payload_is_null[c] = r->payload_null[(size_t)i * r->n_payload + c];
  }
  if (values) *values = r->values.data() + (size_t)i * r->n_aggs;
  return LLKV_OK;
}

void llkv_hip_join_rows_free(llkv_hip_join_rows *rows) { delete reinterpret_cast<JoinRows *>(rows); }

} // extern "C"
