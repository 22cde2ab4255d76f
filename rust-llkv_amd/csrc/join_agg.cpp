// join_agg.cpp — the TPC-H Q3 shape on the GPU (include/llkv_hip.h: llkv_hip_join_groupby_topk and the phased
// llkv_hip_join_agg_* form for a fact table sharded over ranks).
//   dim2 (customer) filter → key set            select + claim
//   dim  (orders)   filter ⋉ dim2 → hash table  select + semi flags + scan/compact + claim
//   fact (lineitem) filter ⋈ dim → (group, value) pairs in scan order   probe-emit (count/scan/write)
//   stable sort by group → per-group left-to-right f64 sums (the reference's order) → top-k
// Group id = position of the dim row in the qualifying-dim-row list (row order): the same on every rank that
// holds the replicated dimension tables, unlike hash-table slots, whose assignment depends on CAS timing.
//
// Sharded fact table: dims replicated, fact rows sharded by chunk.  A group whose rows all
// live on one rank is summed there exactly.  A group that straddles ranks must be summed in global row order
// to stay bit-exact with the reference's sequential `value += v`: ranks all-reduce the per-group row counts
// (int64, the one sizeable collective), find the straddlers (local count ≠ global count), all-gather only
// their raw (group, value) pairs — a handful for a fact table clustered by the key, as lineitem is — and fold
// them in rank order, i.e. global row order.  Each rank then reports the candidates of the groups it alone
// holds plus the straddlers it holds first; the union is merged on the host.
#include "comm.hpp"
#include "engine.hpp"
#include "join.hpp"

#include <algorithm>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace llkv {

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
    HIP_TRY(hj_launch_fill(owner.p, cap * 8, ~0ull, s));
    HIP_TRY(hj_launch_fill(flag.p, 4, 0, s));
    HIP_TRY(hj_launch_claim_list(key, d_rows, n, (unsigned long long *)owner.p, cap - 1, (uint32_t *)flag.p, s));
    uint32_t f = 0;
    HIP_TRY(hipMemcpyAsync(&f, flag.p, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *dup = f != 0;
    return LLKV_OK;
  }
};

// Direct-address form of a key set whose range the column statistics bound (join.hpp): bitmap + rank + group ids.
constexpr uint64_t kMaxDirectSpan = 1ull << 30; // 128 MiB of bits + 64 MiB of word ranks at most
struct DirectTable {
  DB bits, prefix, group, tmp;
  // device words behind the bitmap: [0] non-zero when a key occurred twice, [1] predicate error of a key-bits scan,
  // [2] non-zero when the list the bitmap was filled from (BitmapSink) is not in ascending key order
  uint32_t *flag_p = nullptr;
  bool from_sink = false;
  bool in_key_order = false; // known to the host (a selection from a strictly ascending column): rank = list index, no key twice
  uint64_t n_words = 0;
  int64_t kmin = 0;
  uint64_t span = 0;
  static bool usable(const ColumnInfo &ci) {
    return ci.has_stats && ci.max_i >= ci.min_i && (uint64_t)ci.max_i - (uint64_t)ci.min_i < kMaxDirectSpan;
  }
  // The bitmap (and the duplicate flag behind it); `fill`: the caller's zero fill takes it along, else zeroed here.
  int prepare_bits(const ColumnInfo &ci, hipStream_t s, FillRanges *fill = nullptr) {
    kmin = ci.min_i;
    span = (uint64_t)ci.max_i - (uint64_t)ci.min_i;
    n_words = (span / 64 + 1 + 511) / 512 * 512;
    // the duplicate flag and one error word live right behind the bitmap: one allocation, one fill
    int rc = bits.alloc(n_words * 8 + 32);
    if (rc) return rc;
    flag_p = reinterpret_cast<uint32_t *>(static_cast<char *>(bits.p) + n_words * 8);
    if (fill) fill->add(bits.p, n_words * 8 + 32);
    else HIP_TRY(hj_launch_fill(bits.p, n_words * 8 + 32, 0, s));
    return LLKV_OK;
  }
  // Launches only (no host synchronisation): *flag_p is non-zero afterwards when a key occurred twice.
  // `bits_done`: the bits were set already (BitmapSink of the selection's compaction).
  // `with_groups`: also the rank → list index table (the group ids of the join-aggregate pipeline).
  int build(const ColumnInfo &ci, const JoinKeyColumn &key, const uint64_t *d_rows, uint64_t n, bool with_groups, hipStream_t s, bool bits_done = false) {
    int rc;
    if (!bits.p && (rc = prepare_bits(ci, s))) return rc;
    from_sink = bits_done;
    if (!bits_done) HIP_TRY(hj_launch_bitmap_build(key, d_rows, n, kmin, (unsigned long long *)bits.p, flag_p, s));
    if (with_groups) {
      size_t tb = 0;
      // one entry more than words: prefix[n_words] = the number of set bits (the word behind the bitmap holds the flags; what
      // it counts would land in an entry that does not exist)
      if ((rc = prefix.alloc((n_words + 1) * 4)) || (!in_key_order && (rc = group.alloc((n ? n : 1) * 4)))) return rc;
      HIP_TRY(hj_exclusive_scan_popc(nullptr, &tb, (const uint64_t *)bits.p, (uint32_t *)prefix.p, n_words + 1, s));
      if ((rc = tmp.alloc(tb ? tb : 8))) return rc;
      HIP_TRY(hj_exclusive_scan_popc(tmp.p, &tb, (const uint64_t *)bits.p, (uint32_t *)prefix.p, n_words + 1, s));
      // a bitmap filled by the sink: duplicates show as missing bits, and a list in key order needs no rank → index table
      if (!in_key_order) HIP_TRY(hj_launch_bitmap_groups(key, d_rows, n, kmin, (const uint64_t *)bits.p, (const uint32_t *)prefix.p, n_words, from_sink ? flag_p + 2 : nullptr,
                                      from_sink ? flag_p : nullptr, (uint32_t *)group.p, s));
    }
    return LLKV_OK;
  }
};

// `tmp` belongs to the caller and must outlive the scan on the stream (no host synchronisation here)
int scan_exclusive(const uint64_t *in, uint64_t *out, uint64_t n, DB &tmp, hipStream_t s) {
  size_t tb = 0;
  HIP_TRY(hj_exclusive_scan_u64(nullptr, &tb, in, out, n, s));
  int rc = tmp.alloc(tb ? tb : 8);
  if (rc) return rc;
  HIP_TRY(hj_exclusive_scan_u64(tmp.p, &tb, in, out, n, s));
  return LLKV_OK;
}

// ORDER BY sum DESC, payload[0] ASC (arrow lexsort, llkv-executor/src/lib.rs:13847-13864); ties beyond that
// are left in an unspecified order by the reference — here: dim row order, so every rank count agrees.
bool row_before(const llkv_join_group_row &a, const llkv_join_group_row &b, uint32_t n_payload) {
  if (a.sum != b.sum) return a.sum > b.sum;
  if (n_payload && a.payload[0] != b.payload[0]) return a.payload[0] < b.payload[0];
  return a.group_index < b.group_index;
}
} // namespace

struct JoinAgg {
  const Table *tf = nullptr, *td = nullptr;
  uint32_t n_payload = 0;
  CandidateCols cc{};
  Selection seld;                    // qualifying dim rows before the dim2 semi join
  DB kept;                           // … and after it
  const uint64_t *d_dim_rows = nullptr;
  uint64_t n_dim = 0;                // groups = qualifying dim rows, in row order
  struct View { void *p = nullptr; };
  DB group_state;                    // one block, one memset:
  View sums, cnts, gcnts, report;    // per group: local f64 sum, local rows, exchanged rows (int64), rows this rank reports
  DB s_group, s_val;                 // local (group, value) pairs sorted by group, row order within a group
  DB e_group, e_val;                 // … as emitted (row order), until settle() has looked at the run flag
  uint64_t n_pairs = 0;
  // what prepare() reads back: queued behind its launches and, when deferred, delivered together with the top-k read-back
  Readback rb;
  bool pending = false;
  // … and their device sources, alive until then
  DirectTable set2_bits, dt;
  DB counts, offsets;
  DB st_slot, st_val;                // the probe's stripes: (group id | hash slot, value) pairs per (tile, wave), row order
  uint32_t n_slots = 0, stripe = 0;
  bool from_stripes = false;         // the sums were taken straight from the stripes: no pair count, no compacted pairs yet
  bool keybit_stripes = false;       // … and the stripes hold key-bit positions, not group ids (ProbePlan KEYBIT)
  bool sums_patched = false;         // finish_ranged has replaced boundary groups' sums / counts
  RankCols stripe_ranks{nullptr, nullptr, nullptr, 0};
  DB slot_group;                     // hash form: slot → group id
  bool direct_form = false;
  // ranked form: the dimension selection went straight into the key bitmap (no list of its rows): group id = rank of the
  // key among the set bits, the number of groups stays on the device (dt.prefix[dt.n_words]) until the final read-back
  bool ranked = false;
  uint32_t n_dim_dev = 0, dim_err = 0;
  // range form of a sharded fact table (llkv_hip.h: llkv_hip_join_agg_prepare_ranged): the dimension selection is
  // restricted to the key range of this rank's fact rows, group ids are local, the boundary runs are exchanged
  bool range_form = false;
  uint32_t desc_run = 0;
  uint64_t last_exchange_bytes = 0; // what finish_sharded moved between the ranks (all ranks' contributions)
  DB boundary_d;                        // the boundary kernel's output; its copy travels with prepare()'s read-back
  std::vector<uint64_t> boundary_raw;
  std::vector<uint64_t> boundary_block; // [0] bad, [1] pairs, [2] first key bit, [3] rows, [4] last key bit, [5] rows, [6] / [7] their local groups, 64 + 64 values
  static constexpr uint32_t kBoundaryCap = 64;
  int boundary();
  int finish_ranged(const uint64_t *blocks, const uint64_t *offsets, uint32_t world, uint32_t rank, uint32_t limit, llkv_join_group_row *out_rows,
                    uint32_t *out_n, uint64_t *out_groups);
  DB rank_base;  // ranked form: set bits before each chunk of 2^rank_shift bitmap words; [rank_chunks] = the number of groups
  uint32_t rank_shift = 0, rank_chunks = 0;
  const uint32_t *n_dim_ptr() const { return static_cast<const uint32_t *>(rank_base.p) + rank_chunks; }
  uint32_t pred_err = 0, zero_err = 0;
  DB zeros;                          // one zeroed block: [0] the run flag, [8..15] the top-k selection's state words
  uint32_t *multi_p() const { return static_cast<uint32_t *>(zeros.p); }
  uint64_t *topk_state() const { return static_cast<uint64_t *>(zeros.p) + 8; }
  uint64_t *total_pairs_p() const { return static_cast<uint64_t *>(zeros.p) + 4; } // range form, sums from the stripes: the number of pairs
  uint64_t *slice_best() const { return static_cast<uint64_t *>(zeros.p) + 32; } // [2 · kTopkSlices]: ~best key and groups of every slice (hj_launch_run_sums_stripes)
  uint32_t dup_keys = 0, key_err = 0, multi_run = 0;
  size_t state_bytes = 0;
  std::vector<uint32_t> st_groups;   // straddler pairs of this rank (host)
  std::vector<double> st_vals;
  bool have_straddlers = false;

  int prepare(const llkv_join_side *fact, const llkv_join_side *dim, uint32_t dim_fk_field, const llkv_join_side *dim2,
              const uint32_t *payload_fields, uint32_t n_payload_, const llkv_expr_token *sum_expr, uint32_t sum_expr_len, bool defer = false);
  int settle(bool delivered = false);
  // early exits after a key-bits scan has been launched: its predicate-error word is read before the call reports success
  int error_words(const uint32_t *a, const uint32_t *b) {
    uint32_t ea = 0, eb = 0;
    Readback r;
    int rc;
    if ((a && (rc = r.add(&ea, a, 4, g_ctx.stream))) || (b && (rc = r.add(&eb, b, 4, g_ctx.stream))) || ((a || b) && (rc = r.wait()))) return rc;
    return ea || eb ? set_error(LLKV_INTERNAL, "Arithmetic overflow: Overflow happened in a comparison") : LLKV_OK;
  }
  int compact_pairs(bool run_sums);
  int straddlers();
  int candidates(const uint32_t *f_groups, const double *f_sums, const uint64_t *f_counts, const uint32_t *f_first_rank, uint64_t n_folded,
                 uint32_t rank, uint32_t limit, llkv_join_group_row *out_rows, uint32_t *out_n, uint64_t *out_groups);
};

// ---- key images (engine.hpp: KeyImage) ------------------------------------------------------------------------------------
// The scans of the ranked form stream their key columns for every row.  An Int64 key whose statistics fit 32 bits is read from its
// 4-byte image instead: the plan is lowered over a resolver that presents the field as Int32, and the slot is bound to the image.
// Only a field that the filters and the summed expression do not name (there the column keeps its own type and buffer).
namespace {
bool tokens_name(const llkv_expr_token *t, uint32_t n, uint32_t field) {
  for (uint32_t i = 0; t && i < n; ++i) if (t[i].kind == LLKV_TOK_COLUMN && t[i].field_id == field) return true;
  return false;
}
bool filters_name(const llkv_filter *f, uint32_t n, uint32_t field) {
  for (uint32_t i = 0; f && i < n; ++i) {
    if (f[i].field_id == field) return true; // (an expression filter leaves it 0: field 0 then reads as named — the column is used as it is)
    if (tokens_name(f[i].cmp_left, f[i].cmp_left_len, field) || tokens_name(f[i].cmp_right, f[i].cmp_right_len, field)) return true;
    for (uint32_t k = 0; f[i].list_exprs && k < f[i].list_len; ++k) if (tokens_name(f[i].list_exprs[k], f[i].list_expr_lens[k], field)) return true;
  }
  return false;
}
struct ImageBinds {
  const Table *t = nullptr;
  std::vector<std::pair<uint32_t, const KeyImage *>> of;
  const ColumnInfo *resolve(uint32_t fid) const {
    for (const auto &b : of) if (b.first == fid) return &b.second->info;
    auto it = t->cols.find(fid);
    return it == t->cols.end() ? nullptr : &it->second.info;
  }
  const void *buffer(const LoweredPlan &lp, size_t slot) const {
    const uint8_t part = slot < lp.slot_is_valid.size() ? lp.slot_is_valid[slot] : 0;
    if (part == 0) for (const auto &b : of) if (b.first == lp.slot_fields[slot]) return b.second->d;
    return slot_buffer(t->cols, lp, slot);
  }
  int add(uint32_t field, const llkv_filter *f, uint32_t nf, const llkv_expr_token *e = nullptr, uint32_t ne = 0) {
    if (std::getenv("LLKV_HIP_JOIN_NO_KEY_IMAGE") || filters_name(f, nf, field) || tokens_name(e, ne, field)) return LLKV_OK;
    for (const auto &b : of) if (b.first == field) return LLKV_OK;
    const KeyImage *img = nullptr;
    uint64_t min_rows = 1u << 20; // below that the scans are a few µs whatever they read
    if (const char *e = std::getenv("LLKV_HIP_KEY_IMAGE_MIN_ROWS")) min_rows = (uint64_t)std::atoll(e);
    const int rc = get_key_image(*t, field, min_rows, &img);
    if (!rc && img) of.emplace_back(field, img);
    return rc;
  }
};
} // namespace

int JoinAgg::prepare(const llkv_join_side *fact, const llkv_join_side *dim, uint32_t dim_fk_field, const llkv_join_side *dim2,
                     const uint32_t *payload_fields, uint32_t n_payload_, const llkv_expr_token *sum_expr, uint32_t sum_expr_len, bool defer) {
  int rc = ensure_device();
  if (rc) return rc;
  const auto t_enter = std::chrono::steady_clock::now(); // (LLKV_HIP_TRACE=1: host time up to the last launch, then the wait)
  if (!fact || !dim || !fact->table || !dim->table) return set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  if (n_payload_ > 4) return set_error(LLKV_UNSUPPORTED, "more than 4 payload columns");
  tf = reinterpret_cast<const Table *>(fact->table);
  td = reinterpret_cast<const Table *>(dim->table);
  const Table *t2 = dim2 ? reinterpret_cast<const Table *>(dim2->table) : nullptr;
  if (td->world != 1 || (t2 && t2->world != 1))
    return set_error(LLKV_INVALID_ARGUMENT, "dimension tables are replicated: stage them whole (world = 1) on every rank; only the fact table is sharded");
  n_payload = n_payload_;
  hipStream_t s = g_ctx.stream;

  // ---- dim rows: filter [⋉ dim2] ------------------------------------------------------
  // dim2's key range bounded by its statistics → the semi join is one more conjunct of dim's selection (a bit
  // test); otherwise: hash set of dim2's keys, flags over dim's selection, compaction
  HashSet set2;
  JoinKeyColumn k2{}, fk{}, kd{};
  bool fused_semi = false;
  std::string err;
  uint32_t *key_err_flag = nullptr; // device word the key-bits scan raises on a predicate arithmetic error (lives in set2_bits)
  if (t2 && ((rc = int_key_column(t2, dim2->key_field, &k2)) || (rc = int_key_column(td, dim_fk_field, &fk)))) return rc;
  if ((rc = int_key_column(td, dim->key_field, &kd))) return rc;
  const ColumnInfo &kd_info = td->cols.find(dim->key_field)->second.info;
  const bool direct = DirectTable::usable(kd_info) && !std::getenv("LLKV_HIP_JOIN_HASH");
  if (t2) fused_semi = DirectTable::usable(t2->cols.find(dim2->key_field)->second.info) && !std::getenv("LLKV_HIP_JOIN_HASH");
  // dim's bitmap is filled while its selection is compacted (the keys of the selected rows are gathered once: that
  // sparse gather is most of what building the table costs).  A key column in strictly ascending row order (staging
  // statistic): the selection is in key order and has no key twice — a key's rank among the set bits is its group id;
  // any other finds out about its order during the compaction.
  const bool dim_sorted = direct && kd_info.ascending && !std::getenv("LLKV_HIP_JOIN_UNSORTED");
  const bool sink_bits = fused_semi && direct && !std::getenv("LLKV_HIP_JOIN_NO_SINK");
  int64_t key_range_lo = 0, key_range_hi = -1; // (range form of a sharded fact table: the keys this rank's fact rows can hold)
  const bool use_key_range = range_form;
  if (range_form) {
    auto fk_it = tf->cols.find(fact->key_field);
    if (fk_it == tf->cols.end()) return set_error(LLKV_NOT_FOUND, "field " + std::to_string(fact->key_field) + " not found");
    if (tf->local_rows && !fk_it->second.has_local_stats) return set_error(LLKV_UNSUPPORTED, "range form: no statistics of this rank's fact keys");
    if (tf->local_rows) { key_range_lo = fk_it->second.local_min; key_range_hi = fk_it->second.local_max; }
  }
  ranked = direct && dim_sorted && (!t2 || fused_semi) && kd_info.dtype == LLKV_DT_INT64 && td->local_rows && !std::getenv("LLKV_HIP_JOIN_NO_SINK") &&
           !std::getenv("LLKV_HIP_JOIN_LISTED");
  // ---- everything that starts from zero, in one fill: the key bitmaps, the probe's per-stripe counts, the flags --------
  const TileSet *ts = nullptr;
  uint32_t probe_tile = 8192; // rows of a probe workgroup (a wave owns a quarter: its stripe)
  if (const char *e = std::getenv("LLKV_HIP_PROBE_TILE")) { const long v = std::atol(e); if (v >= 1024 && v <= 65536 && v % 512 == 0) probe_tile = (uint32_t)v; }
  if ((rc = get_tileset(*tf, probe_tile, &ts))) return rc;
  n_slots = ts->n_tiles * (kBlock / 64);
  // The ranked form's dimension plan is lowered and looked up BEFORE the first launch: the fill and the dim2 key-set scan are a
  // few µs each, and a host that lowers this plan between them leaves the device idle until the dimension scan is queued
  // (17 µs of a 320 µs query once the customer scan took 5 µs instead of 13).
  LoweredPlan dim_kp;
  JitKernel dim_kk;
  const TileSet *dim_ts = nullptr;
  ImageBinds img_d, img_f;
  img_d.t = td;
  img_f.t = tf;
  if (ranked) {
    // (the dimension's key and its foreign key into dim2 from their 4-byte images when the statistics allow: 20 → 12 B per order)
    if ((rc = img_d.add(dim->key_field, dim->filters, dim->n_filters)) || (t2 && (rc = img_d.add(dim_fk_field, dim->filters, dim->n_filters)))) return rc;
    auto resolve_d = [&](uint32_t fid) -> const ColumnInfo * { return img_d.resolve(fid); };
    llkv_expr_token key_tok;
    std::memset(&key_tok, 0, sizeof key_tok);
    key_tok.kind = LLKV_TOK_COLUMN;
    key_tok.field_id = dim->key_field;
    if ((rc = lower_emit(resolve_d, dim->filters, dim->n_filters, nullptr, 0, &key_tok, 1, &dim_kp, &err, false, nullptr, t2 ? &dim_fk_field : nullptr, nullptr, true))) return set_error(rc, err);
    if ((rc = jit_compile(JitKind::KeyBits, dim_kp.type_string, &dim_kk, &err))) return set_error(rc, err);
    // 2 048-row tiles: a wave's quarter is exactly one batch of kSelUnroll steps — every load and key-set gather of the tile
    // goes out before the first use (SF10 orders: 81 µs; 4 096: 85, 8 192: 90, 16 384: 92; 1 024: 114 — half-empty batches)
    uint32_t dim_tile = 2048;
    if (const char *e = std::getenv("LLKV_HIP_KEYBITS_TILE")) { const long v = std::atol(e); if (v >= 512 && v <= 65536 && v % 512 == 0) dim_tile = (uint32_t)v; }
    if ((rc = get_tileset(*td, dim_tile, &dim_ts))) return rc;
  }
  // … and dim2's key-set scan with it (the fill below is 4 µs: whatever the host does between the two launches, the device waits for)
  LoweredPlan kp2;
  JitKernel kk2;
  const TileSet *ts2 = nullptr;
  bool have_kp2 = false;
  if (t2 && fused_semi && t2->cols.find(dim2->key_field)->second.info.dtype == LLKV_DT_INT64 && t2->local_rows) {
    auto resolve_2 = [&](uint32_t fid) -> const ColumnInfo * {
      auto it = t2->cols.find(fid);
      return it == t2->cols.end() ? nullptr : &it->second.info;
    };
    llkv_expr_token key_tok;
    std::memset(&key_tok, 0, sizeof key_tok);
    key_tok.kind = LLKV_TOK_COLUMN;
    key_tok.field_id = dim2->key_field;
    std::string ignore;
    if (lower_emit(resolve_2, dim2->filters, dim2->n_filters, nullptr, 0, &key_tok, 1, &kp2, &ignore) == LLKV_OK &&
        jit_compile(JitKind::KeyBits, kp2.type_string, &kk2, &ignore) == LLKV_OK) {
      if ((rc = get_tileset(*t2, t2->local_rows < (4u << 20) ? 2048 : 8192, &ts2))) return rc;
      have_kp2 = true;
    }
  }
  {
    FillRanges fr;
    if (fused_semi && (rc = set2_bits.prepare_bits(t2->cols.find(dim2->key_field)->second.info, s, &fr))) return rc;
    if (range_form && !ranked) return set_error(LLKV_UNSUPPORTED, "range form: the dimension key must be an Int64 column in ascending row order with a statistics-bounded range");
  if ((sink_bits || ranked) && (rc = dt.prepare_bits(kd_info, s, &fr))) return rc;
    dt.in_key_order = dim_sorted;
    if ((rc = counts.alloc((size_t)(n_slots + 1) * 8)) || (rc = offsets.alloc((size_t)(n_slots + 1) * 8)) || (rc = zeros.alloc(256 + 2 * kTopkSlices * 8))) return rc;
    fr.add(counts.p, (size_t)(n_slots + 1) * 8); // the extra trailing 0 makes offsets[n_slots] the total
    fr.add(zeros.p, 256 + 2 * kTopkSlices * 8);  // (… and the top-k selection's slice winners, which the run sums fill on the way)
    HIP_TRY(hj_launch_fill_zero_ranges(fr, s));
  }
  if (t2) {
    const ColumnInfo &k2_info = t2->cols.find(dim2->key_field)->second.info;
    bool bits_set = false;
    if (have_kp2) {
      // the key set straight from dim2's scan: rows that pass set their bit — no selection vector, no read-back
      if (!kp2.always_false) {
        ScanParams p2;
        std::memset(&p2, 0, sizeof p2);
        for (size_t i = 0; i < kp2.slot_fields.size(); ++i) p2.col[i] = slot_buffer(t2->cols, kp2, i);
        for (size_t i = 0; i < kp2.lit_i.size(); ++i) p2.lit_i[i] = kp2.lit_i[i];
        for (size_t i = 0; i < kp2.lit_f.size(); ++i) p2.lit_f[i] = kp2.lit_f[i];
        p2.tiles = ts2->d_tiles;
        p2.n_tiles = ts2->n_tiles;
        p2.aux_out = (uint64_t *)set2_bits.bits.p;
        p2.aux_out32 = set2_bits.flag_p + 1; // predicate-error word (read with the pair count)
        p2.kb_min = set2_bits.kmin;
        p2.kb_span = set2_bits.span;
        if ((rc = jit_launch_raw(kk2.fn, ts2->n_tiles, &p2, sizeof p2, s))) return rc;
      }
      bits_set = true;
      key_err_flag = set2_bits.flag_p + 1;
    }
    if (!bits_set) {
      Selection sel2;
      if ((rc = run_selection(t2, dim2->filters, dim2->n_filters, nullptr, 0, &sel2))) return rc;
      bool dup = false;
      if (fused_semi) { // a set: a key that occurs twice is no error
        if ((rc = set2_bits.build(k2_info, k2, sel2.d_dev, sel2.n, false, s))) return rc;
        HIP_TRY(hipStreamSynchronize(s)); // sel2 is released at the end of the block
      } else if ((rc = set2.build(k2, sel2.d_dev, sel2.n, &dup, s))) {
        return rc;
      }
    }
  }
  // dim's own table: when the statistics bound its key, the bitmap is filled while the selection is compacted — or, for
  // a key column in ascending row order, by the selection scan itself (the ranked form): rows that pass set their bit,
  // nothing is listed, counted or read back; a key's rank among the set bits is its group id and the rank scan's total
  // the number of groups
  bool dt_bits_done = false;
  uint32_t *dim_err_flag = nullptr;
  if (ranked) {
    const LoweredPlan &kp = dim_kp;
    const JitKernel &kk = dim_kk;
    const TileSet *tsd = dim_ts;
    if (!kp.always_false && td->local_rows) {
      ScanParams pd;
      std::memset(&pd, 0, sizeof pd);
      for (size_t i = 0; i < kp.slot_fields.size(); ++i) pd.col[i] = img_d.buffer(kp, i);
      for (size_t i = 0; i < kp.lit_i.size(); ++i) pd.lit_i[i] = kp.lit_i[i];
      for (size_t i = 0; i < kp.lit_f.size(); ++i) pd.lit_f[i] = kp.lit_f[i];
      pd.tiles = tsd->d_tiles;
      pd.n_tiles = tsd->n_tiles;
      pd.aux_out = (uint64_t *)dt.bits.p;
      pd.aux_out32 = dt.flag_p + 1; // predicate-error word
      pd.kb_min = dt.kmin;
      pd.kb_span = dt.span;
      if (t2) { pd.bm_bits = (const uint64_t *)set2_bits.bits.p; pd.bm_min = set2_bits.kmin; pd.bm_span = set2_bits.span; }
      if (use_key_range) { pd.kb_ranged = 1; pd.kb_lo = key_range_lo; pd.kb_hi = key_range_hi; } // (no fact rows: the empty range — no tile stays)
      if ((rc = jit_launch_raw(kk.fn, tsd->n_tiles, &pd, sizeof pd, s))) return rc;
    }
    dim_err_flag = dt.flag_p + 1;
    dt_bits_done = true;
    n_dim = std::min<uint64_t>(td->local_rows, dt.span + 1); // an upper bound: sizes the group state (the kernels read the count on the device)
  } else if (fused_semi) {
    auto resolve_d = [&](uint32_t fid) -> const ColumnInfo * {
      auto it = td->cols.find(fid);
      return it == td->cols.end() ? nullptr : &it->second.info;
    };
    LoweredPlan sel_plan;
    if ((rc = lower_selection_in_set(resolve_d, dim->filters, dim->n_filters, dim_fk_field, &sel_plan, &err))) return set_error(rc, err);
    const KeySetView view{(const uint64_t *)set2_bits.bits.p, set2_bits.kmin, set2_bits.span};
    BitmapSink sink{};
    if (sink_bits) {
      sink = BitmapSink{kd.values, kd.width, kd.is_signed, dt.kmin, (unsigned long long *)dt.bits.p, dt.flag_p + 2};
      dt_bits_done = true;
    }
    if ((rc = run_selection_lowered(td, sel_plan, &seld, &view, 1, dt_bits_done ? &sink : nullptr, false))) return rc; // the bit test gathers: evaluate it once
  } else if ((rc = run_selection(td, dim->filters, dim->n_filters, nullptr, 0, &seld))) {
    return rc;
  }
  if (!ranked) {
    n_dim = seld.n;
    d_dim_rows = seld.d_dev;
  }
  if (t2 && !fused_semi && seld.n) {
    DB flags, offs, scan_tmp;
    if ((rc = flags.alloc(seld.n * 8)) || (rc = offs.alloc((seld.n + 1) * 8))) return rc;
    HIP_TRY(hj_launch_semi_flags(fk, seld.d_dev, seld.n, k2, (const unsigned long long *)set2.owner.p, set2.cap - 1, (uint64_t *)flags.p, s));
    if ((rc = scan_exclusive((const uint64_t *)flags.p, (uint64_t *)offs.p, seld.n, scan_tmp, s))) return rc;
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
  if (n_dim >= (1ull << 32)) return set_error(LLKV_UNSUPPORTED, "dimension too large");
  if (n_dim == 0) return error_words(key_err_flag, nullptr);
  state_bytes = (n_dim * 8 + 4095) / 4096 * 4096; // whole pages: the memset is one fill kernel
  if ((rc = group_state.alloc(5 * state_bytes))) return rc; // (the fifth: key-bit positions by group, half used — RankCols::pos_out)
  // (cnts first: the ranked form zeroes the first one — or three — arrays up to the real group count inside the probe launch)
  cnts.p = group_state.p;
  sums.p = (char *)group_state.p + state_bytes;
  gcnts.p = (char *)group_state.p + 2 * state_bytes;
  report.p = (char *)group_state.p + 3 * state_bytes;
  // one rank: only the row counts start from zero (a sum is read only where the count is not zero)
  if (ranked) {
  } else if (tf->world == 1) HIP_TRY(hj_launch_fill(cnts.p, state_bytes, 0, s));
  else HIP_TRY(hj_launch_fill(group_state.p, 3 * state_bytes, 0, s));

  // ---- dim hash table, slot → group id -----------------------------------------------------
  std::memset(&cc, 0, sizeof cc);
  cc.key = kd;
  cc.n_payload = n_payload;
  for (uint32_t i = 0; i < n_payload; ++i) if ((rc = int_key_column(td, payload_fields[i], &cc.payload[i]))) return rc;
  // key range bounded by the column statistics → bitmap + rank (no hashing, and a clustered fact table probes it
  // almost sequentially); otherwise the open-addressing table
  HashSet ht;
  bool dup = false;
  direct_form = direct;
  // one rank, ranked form, sums straight from the stripes: the probe emits the key's bit position and the head of every run
  // looks the rank up
  // (… and one rank of a range form: its boundary runs are read off the stripes too, nothing of it needs the pairs compacted)
  const bool stripes_ok = direct && !std::getenv("LLKV_HIP_JOIN_COMPACT") && ((tf->world == 1 && defer) || (range_form && !std::getenv("LLKV_HIP_JOIN_RANGE_COMPACT")));
  keybit_stripes = ranked && stripes_ok && dt.span < (1ull << 32) && !std::getenv("LLKV_HIP_JOIN_PROBE_RANKS");
  bool rank_in_probe = false; // such a probe needs no rank itself: its first workgroups rank the bitmap on the way (select.hip.h: probe_rank_chunk)
  if (ranked) { // the word ranks, chunk by chunk, and the chunk bases (in their last entry: the number of groups) — one launch
    rank_shift = 10;
    while (((dt.n_words + (1ull << rank_shift) - 1) >> rank_shift) > 256) ++rank_shift; // about one workgroup per CU
    if (std::getenv("LLKV_HIP_JOIN_RANK_SCAN")) { // one chunk: the library scan's absolute ranks (two launches)
      rank_shift = 40;
      rank_chunks = 1;
      if ((rc = rank_base.alloc(8))) return rc;
      size_t tb = 0;
      if ((rc = dt.prefix.alloc((dt.n_words + 1) * 4))) return rc;
      HIP_TRY(hj_exclusive_scan_popc(nullptr, &tb, (const uint64_t *)dt.bits.p, (uint32_t *)dt.prefix.p, dt.n_words + 1, s));
      if ((rc = dt.tmp.alloc(tb ? tb : 8))) return rc;
      HIP_TRY(hj_exclusive_scan_popc(dt.tmp.p, &tb, (const uint64_t *)dt.bits.p, (uint32_t *)dt.prefix.p, dt.n_words + 1, s));
      HIP_TRY(hipMemsetAsync(rank_base.p, 0, 4, s));
      HIP_TRY(hipMemcpyAsync((uint32_t *)rank_base.p + 1, (uint32_t *)dt.prefix.p + dt.n_words, 4, hipMemcpyDeviceToDevice, s));
    } else {
      rank_chunks = (uint32_t)((dt.n_words + (1ull << rank_shift) - 1) >> rank_shift);
      if ((rc = dt.prefix.alloc(dt.n_words * 4)) || (rc = rank_base.alloc((size_t)(rank_chunks + 1) * 4))) return rc;
      rank_in_probe = keybit_stripes && rank_chunks <= kBlock && ts->n_tiles >= rank_chunks && tf->local_rows && !std::getenv("LLKV_HIP_JOIN_RANK_LAUNCH");
      if (!rank_in_probe) HIP_TRY(hj_launch_rank_words((const uint64_t *)dt.bits.p, dt.n_words, rank_shift, (uint32_t *)dt.prefix.p, (uint32_t *)rank_base.p, multi_p() + 2, s));
    }
    dt.from_sink = true;
    cc.rank_bits = (const uint64_t *)dt.bits.p;
    cc.rank_prefix = (const uint32_t *)dt.prefix.p;
    cc.rank_base = (const uint32_t *)rank_base.p;
    cc.rank_chunk_shift = rank_shift;
    cc.rank_chunks = rank_chunks;
    cc.rank_words = dt.n_words;
    cc.rank_kmin = dt.kmin;
    cc.rank_rows = td->local_rows;
  } else if (direct) { // launches only; the duplicate flag is read with the pair count below
    if ((rc = dt.build(kd_info, kd, d_dim_rows, n_dim, true, s, dt_bits_done && d_dim_rows == seld.d_dev))) return rc;
  } else {
    if ((rc = ht.build(kd, d_dim_rows, n_dim, &dup, s))) return rc;
  }
  if (dup) return set_error(LLKV_UNSUPPORTED, "dimension key is not unique: groups are not identified by the dim row");
  if (!direct) {
    if ((rc = slot_group.alloc(ht.cap * 4))) return rc;
    HIP_TRY(hj_launch_slot_groups(kd, d_dim_rows, n_dim, (const unsigned long long *)ht.owner.p, ht.cap - 1, (uint32_t *)slot_group.p, s));
  }

  // ---- fact probe-emit -------------------------------------------------------------------
  // (the fact key from its 4-byte image when the statistics allow: the Q3 probe streams 8 B of every lineitem row instead of 12)
  if (ranked && (rc = img_f.add(fact->key_field, fact->filters, fact->n_filters, sum_expr, sum_expr_len))) return rc;
  auto resolve = [&](uint32_t fid) -> const ColumnInfo * { return img_f.resolve(fid); };
  LoweredPlan plan;
  if ((rc = lower_probe(resolve, fact->filters, fact->n_filters, fact->key_field, sum_expr, sum_expr_len, &plan, &err, keybit_stripes))) return set_error(rc, err);
  if (plan.always_false || tf->local_rows == 0) {
    if (ranked) { // nothing probes: the group state still starts from zero, and the group count is wanted
      if (rank_in_probe) HIP_TRY(hj_launch_rank_words((const uint64_t *)dt.bits.p, dt.n_words, rank_shift, (uint32_t *)dt.prefix.p, (uint32_t *)rank_base.p, multi_p() + 2, s));
      HIP_TRY(hj_launch_fill(group_state.p, (tf->world == 1 ? 1 : 3) * state_bytes, 0, s));
      if ((rc = rb.add(&n_dim_dev, n_dim_ptr(), 4, s)) || (rc = rb.wait())) return rc;
      n_dim = n_dim_dev;
    }
    HIP_TRY(hipStreamSynchronize(s));
    return error_words(key_err_flag, dim_err_flag);
  }
  JitKernel k;
  if ((rc = jit_compile(JitKind::Probe, plan.type_string, &k, &err))) return set_error(rc, err);
  if (rank_in_probe) { // a helper may wait for the ranking workgroups: all of them and the helpers must fit on the device together
    int per_cu = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k.fn2, (int)kBlock, 0) != hipSuccess || (uint64_t)per_cu * g_ctx.cu_count < (uint64_t)rank_chunks + 64) {
      rank_in_probe = false;
      HIP_TRY(hj_launch_rank_words((const uint64_t *)dt.bits.p, dt.n_words, rank_shift, (uint32_t *)dt.prefix.p, (uint32_t *)rank_base.p, multi_p() + 2, s));
    }
  }
  ScanParams p;
  std::memset(&p, 0, sizeof p);
  for (size_t i = 0; i < plan.slot_fields.size(); ++i) p.col[i] = img_f.buffer(plan, i);
  for (size_t i = 0; i < plan.lit_i.size(); ++i) p.lit_i[i] = plan.lit_i[i];
  for (size_t i = 0; i < plan.lit_f.size(); ++i) p.lit_f[i] = plan.lit_f[i];
  p.tiles = ts->d_tiles;
  p.n_tiles = ts->n_tiles;
  p.sub_rows = probe_tile / (kBlock / 64);
  p.tile_partials = (uint64_t *)counts.p;
  if (direct) {
    p.bm_bits = (const uint64_t *)dt.bits.p;
    p.bm_prefix = (const uint32_t *)dt.prefix.p;
    p.bm_group = (const uint32_t *)dt.group.p;
    p.bm_unsorted = dt.from_sink && !ranked ? dt.flag_p + 2 : nullptr;
    p.bm_min = dt.kmin;
    p.bm_span = dt.span;
    if (ranked) {
      p.bm_group = nullptr; // the rank is the group id
      p.bm_base = rank_chunks > 1 ? (const uint32_t *)rank_base.p : nullptr;
      p.bm_chunk_shift = rank_shift;
      stripe_ranks = RankCols{(const uint64_t *)dt.bits.p, (const uint32_t *)dt.prefix.p, p.bm_base, rank_shift};
      if (keybit_stripes && tf->world == 1 && !std::getenv("LLKV_HIP_JOIN_NO_GROUP_POS")) { // one rank: the candidates are read off this run's own sums
        stripe_ranks.pos_out = (uint32_t *)((char *)group_state.p + 4 * state_bytes);
        cc.pos_by_group = stripe_ranks.pos_out;
      }
      p.zero_k = tf->world == 1 ? 1 : 3;
      p.zero_words = (uint64_t *)group_state.p;
      p.zero_stride = state_bytes / 8;
      p.zero_n = n_dim_ptr();
      if (rank_in_probe) {
        p.rk_bits = (const uint64_t *)dt.bits.p;
        p.rk_words = dt.n_words;
        p.rk_shift = rank_shift;
        p.rk_chunks = rank_chunks;
        p.rk_prefix = (uint32_t *)dt.prefix.p;
        p.rk_base = (uint32_t *)rank_base.p;
        p.rk_state = multi_p() + 40; // four of the zeroed block's free words
        // the helpers that zero the group state: workgroups of the second round of dispatch
        p.rk_helpers = std::min<uint32_t>(64, ts->n_tiles);
        p.rk_help_first = std::min<uint32_t>(5 * g_ctx.cu_count, ts->n_tiles - p.rk_helpers);
      }
    }
  } else {
    p.ht_owner = (const unsigned long long *)ht.owner.p;
    p.ht_mask = ht.cap - 1;
    p.ht_keys = kd.values;
    p.ht_key_width = kd.width;
    p.ht_key_signed = kd.is_signed;
  }
  // single pass over the fact columns: every (tile, wave) writes its pairs into its own stripe and reports its
  // count; only the emitted pairs (a few percent of the rows for Q3) are touched again by the compaction
  stripe = p.sub_rows;
  if ((rc = st_slot.alloc((size_t)n_slots * stripe * 4)) || (rc = st_val.alloc((size_t)n_slots * stripe * 8))) return rc;
  p.aux_in = nullptr;
  p.aux_out32 = (uint32_t *)st_slot.p;
  p.aux_out = (uint64_t *)st_val.p;
  if ((rc = jit_launch_raw(direct ? k.fn2 : k.fn, ts->n_tiles, &p, sizeof p, s))) return rc;
  // ---- per-group sums in scan order ------------------------------------------------------------
  // The pairs are in row order.  A fact table clustered by the join key leaves every group as ONE run of them: sum
  // the runs where they lie; only when some group turns out to have a second run, sort (stable) by group first.
  // One rank, direct table: straight from the stripes (no scan, no compaction, no pair count) — whoever needs the
  // pairs themselves later (the sort, a sharded fact table's straddlers) compacts them then.
  from_stripes = stripes_ok;
  const RankCols stripe_rank_cols = keybit_stripes ? stripe_ranks : RankCols{nullptr, nullptr, nullptr, 0};
  if (from_stripes) {
    HIP_TRY(hj_launch_run_sums_stripes((const uint32_t *)st_slot.p, (const uint64_t *)st_val.p, (const uint64_t *)counts.p, n_slots, stripe, (double *)sums.p,
                                       (uint64_t *)cnts.p, multi_p(), s, stripe_rank_cols, slice_best(), range_form ? total_pairs_p() : nullptr));
  } else if ((rc = compact_pairs(true))) {
    return rc;
  }
  if (range_form) { // the boundary runs of the pair stream, while the pair count is still on its way to the host
    boundary_raw.assign(8 + 2 * kBoundaryCap, 0);
    if ((rc = boundary_d.alloc(boundary_raw.size() * 8))) return rc;
    if (from_stripes)
      HIP_TRY(hj_launch_boundary_runs_stripes((const uint32_t *)st_slot.p, (const uint64_t *)st_val.p, (const uint64_t *)counts.p, n_slots, stripe, kBoundaryCap, cc, stripe_rank_cols,
                                              (uint64_t *)boundary_d.p, s));
    else
      HIP_TRY(hj_launch_boundary_runs((const uint32_t *)e_group.p, (const uint64_t *)e_val.p, 0, (const uint64_t *)offsets.p + n_slots, kBoundaryCap, cc, (uint64_t *)boundary_d.p, s));
    if ((rc = rb.add(boundary_raw.data(), boundary_d.p, boundary_raw.size() * 8, s))) return rc;
    if (from_stripes && (rc = rb.add(&n_pairs, total_pairs_p(), 8, s))) return rc;
  }
  if ((!from_stripes && (rc = rb.add(&n_pairs, (uint64_t *)offsets.p + n_slots, 8, s))) || (from_stripes && (rc = rb.add(&pred_err, multi_p() + 1, 4, s))) || (rank_in_probe && (rc = rb.add(&zero_err, multi_p() + 43, 4, s))) ||
      (direct && (rc = rb.add(&dup_keys, dt.flag_p, 4, s))) || (key_err_flag && (rc = rb.add(&key_err, key_err_flag, 4, s))) ||
      (dim_err_flag && (rc = rb.add(&dim_err, dim_err_flag, 4, s))) || (ranked && (rc = rb.add(&n_dim_dev, n_dim_ptr(), 4, s))) ||
      (range_form && (rc = rb.add(&desc_run, multi_p() + 3, 4, s))) ||
      (rc = rb.add(&multi_run, multi_p(), 4, s)))
    return rc;
  pending = true;
  if (defer) return LLKV_OK; // the sources are members: the selection's last workgroup carries the items
  const auto t_launched = std::chrono::steady_clock::now();
  rc = settle();
  if (std::getenv("LLKV_HIP_TRACE"))
    std::fprintf(stderr, "[llkv join_agg] host to the last launch %7.1f us, wait for the device %7.1f us\n",
                 std::chrono::duration<double, std::micro>(t_launched - t_enter).count(),
                 std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_launched).count());
  return rc;
}

// Stripes → contiguous pairs in row order (e_group / e_val, sized by the bound of one pair per local fact row; the
// pair count stays on the device: offsets[n_slots]), and optionally the run sums over them.  Launches only.
int JoinAgg::compact_pairs(bool run_sums) {
  hipStream_t s = g_ctx.stream;
  int rc;
  DB scan_tmp;
  if (n_slots <= 64 * 1024) HIP_TRY(launch_exclusive_scan((const uint64_t *)counts.p, (uint64_t *)offsets.p, n_slots, s)); // one workgroup, rounds of 8 192
  else {
    if ((rc = scan_exclusive((const uint64_t *)counts.p, (uint64_t *)offsets.p, n_slots + 1, scan_tmp, s))) return rc;
    HIP_TRY(hipStreamSynchronize(s)); // scan_tmp is released on return
  }
  const uint64_t max_pairs = tf->local_rows;
  if ((rc = e_group.alloc(max_pairs * 4)) || (rc = e_val.alloc(max_pairs * 8))) return rc;
  HIP_TRY(hj_launch_compact_stripes((const uint32_t *)st_slot.p, (const uint64_t *)st_val.p, (const uint64_t *)counts.p, (const uint64_t *)offsets.p, n_slots, stripe,
                                    direct_form ? nullptr : (const uint32_t *)slot_group.p, (uint32_t *)e_group.p, (uint64_t *)e_val.p, s, // slot / key bit → group id on the way
                                    keybit_stripes ? stripe_ranks : RankCols{nullptr, nullptr, nullptr, 0}));
  if (run_sums)
    HIP_TRY(hj_launch_run_sums_dev((const uint32_t *)e_group.p, (const uint64_t *)e_val.p, (const uint64_t *)offsets.p + n_slots, max_pairs, (double *)sums.p,
                                   (uint64_t *)cnts.p, multi_p(), s, range_form ? multi_p() + 3 : nullptr));
  return LLKV_OK;
}

// The read-back of prepare(): errors, and the sort-based sums when some group's pairs were not one run.
// `delivered`: the caller has waited on `rb` already.
int JoinAgg::settle(bool delivered) {
  if (!pending) return LLKV_OK;
  int rc;
  if (!delivered && (rc = rb.wait())) return rc;
  pending = false;
  hipStream_t s = g_ctx.stream;
  if (zero_err) { n_pairs = 0; return set_error(LLKV_INTERNAL, "join pipeline: the group state was not zeroed (a probe workgroup gave up waiting for the group count)"); }
  if (key_err || pred_err || dim_err) { n_pairs = 0; return set_error(LLKV_INTERNAL, "Arithmetic overflow: Overflow happened in a comparison"); }
  if (ranked) n_dim = n_dim_dev; // the bound that sized the group state → the number of groups
  if (dup_keys) { n_pairs = 0; return set_error(LLKV_UNSUPPORTED, "dimension key is not unique: groups are not identified by the dim row"); }
  if (from_stripes) {
    if (!multi_run && !std::getenv("LLKV_HIP_JOIN_SORT")) return LLKV_OK; // sums and counts are final; nobody asked for the pairs
    // the sort needs the pairs: compact them now, and their number
    Readback cnt;
    if ((rc = compact_pairs(false)) || (rc = cnt.add(&n_pairs, (uint64_t *)offsets.p + n_slots, 8, s)) || (rc = cnt.wait())) return rc;
    from_stripes = false;
  }
  if (n_pairs >= kPredErrorBit) { n_pairs = 0; return set_error(LLKV_INTERNAL, "Arithmetic overflow: Overflow happened in a comparison"); }
  if (n_pairs == 0) return LLKV_OK;
  if (range_form || (!multi_run && !std::getenv("LLKV_HIP_JOIN_SORT"))) {
    std::swap(s_group.p, e_group.p); // a group's pairs are contiguous and in row order: all the later phases need
    std::swap(s_val.p, e_val.p);
    if (tf->world != 1 && !range_form) HIP_TRY(hipMemcpyAsync(gcnts.p, cnts.p, n_dim * 8, hipMemcpyDeviceToDevice, s)); // the image the ranks all-reduce
    return LLKV_OK; // (range form with a second run of some group: finish_ranged refuses, on every rank)
  }
  uint32_t bits = 1;
  while ((1ull << bits) < n_dim) ++bits;
  DB tmp;
  size_t tb = 0;
  if ((rc = s_group.alloc(n_pairs * 4)) || (rc = s_val.alloc(n_pairs * 8))) return rc;
  HIP_TRY(hj_launch_fill(group_state.p, 2 * state_bytes, 0, s)); // sums and counts again
  HIP_TRY(hj_sort_u32_u64(nullptr, &tb, (const uint32_t *)e_group.p, (uint32_t *)s_group.p, (const uint64_t *)e_val.p, (uint64_t *)s_val.p, n_pairs, bits, s));
  if ((rc = tmp.alloc(tb))) return rc;
  HIP_TRY(hj_sort_u32_u64(tmp.p, &tb, (const uint32_t *)e_group.p, (uint32_t *)s_group.p, (const uint64_t *)e_val.p, (uint64_t *)s_val.p, n_pairs, bits, s));
  HIP_TRY(hj_launch_segment_sums((const uint32_t *)s_group.p, (const uint64_t *)s_val.p, n_pairs, (double *)sums.p, (uint64_t *)cnts.p, s));
  HIP_TRY(hipMemcpyAsync(gcnts.p, cnts.p, n_dim * 8, hipMemcpyDeviceToDevice, s));
  HIP_TRY(hipStreamSynchronize(s)); // tmp is released at the end of the block
  e_group.alloc(8);
  e_val.alloc(8);
  return LLKV_OK;
}

// After the all-reduce of gcnts: this rank's raw pairs of the groups that other ranks hold rows of too.
int JoinAgg::straddlers() {
  st_groups.clear();
  st_vals.clear();
  have_straddlers = true;
  if (n_pairs == 0 || n_dim == 0) return LLKV_OK;
  hipStream_t s = g_ctx.stream;
  DB flags, offs;
  int rc;
  if ((rc = flags.alloc((n_pairs + 1) * 8)) || (rc = offs.alloc((n_pairs + 1) * 8))) return rc;
  HIP_TRY(hj_launch_fill(flags.p, (n_pairs + 1) * 8, 0, s));
  HIP_TRY(hj_launch_straddler_flags((const uint32_t *)s_group.p, n_pairs, (const uint64_t *)cnts.p, (const int64_t *)gcnts.p, (uint64_t *)flags.p, s));
  DB scan_tmp;
  if ((rc = scan_exclusive((const uint64_t *)flags.p, (uint64_t *)offs.p, n_pairs + 1, scan_tmp, s))) return rc;
  uint64_t n = 0;
  Readback rb;
  if ((rc = rb.add(&n, (uint64_t *)offs.p + n_pairs, 8, s)) || (rc = rb.wait())) return rc;
  if (n == 0) return LLKV_OK;
  if (n > (64ull << 20)) return set_error(LLKV_UNSUPPORTED, "fact rows of groups that straddle ranks exceed 64 Mi: the fact table is not clustered by the join key");
  DB og, ov;
  if ((rc = og.alloc(n * 4)) || (rc = ov.alloc(n * 8))) return rc;
  HIP_TRY(hj_launch_compact_pairs((const uint32_t *)s_group.p, (const uint64_t *)s_val.p, (const uint64_t *)flags.p, (const uint64_t *)offs.p, n_pairs,
                                  (uint32_t *)og.p, (uint64_t *)ov.p, s));
  st_groups.resize(n);
  st_vals.resize(n);
  HIP_TRY(hipMemcpyAsync(st_groups.data(), og.p, n * 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(st_vals.data(), ov.p, n * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return LLKV_OK;
}

int JoinAgg::candidates(const uint32_t *f_groups, const double *f_sums, const uint64_t *f_counts, const uint32_t *f_first_rank, uint64_t n_folded,
                        uint32_t rank, uint32_t limit, llkv_join_group_row *out_rows, uint32_t *out_n, uint64_t *out_groups) {
  *out_n = 0;
  if (out_groups) *out_groups = 0;
  if (n_dim == 0) return LLKV_OK;
  hipStream_t s = g_ctx.stream;
  int rc;
  // groups this rank reports: the ones it alone holds … (one rank holds every group alone: its counts are the report)
  const bool alone = (tf->world == 1 || range_form) && n_folded == 0;
  const bool by_selection = !std::getenv("LLKV_HIP_TOPK_SORT") && limit <= kTopkSlices;
  // a deferred prepare(): its read-back rides with the selection's below; anything else needs it settled first
  if (pending && !(alone && by_selection) && (rc = settle())) return rc;
  if (!alone && tf->world == 1) HIP_TRY(hipMemcpyAsync(gcnts.p, cnts.p, n_dim * 8, hipMemcpyDeviceToDevice, s)); // nobody else holds rows
  if (!alone) HIP_TRY(hj_launch_report_counts((const uint64_t *)cnts.p, (const int64_t *)gcnts.p, n_dim, (uint64_t *)report.p, s));
  const uint64_t *report_p = alone ? (const uint64_t *)cnts.p : (const uint64_t *)report.p;
  // … plus the straddlers it holds first, with their exact sums and global counts
  std::vector<uint32_t> mg;
  std::vector<double> ms;
  std::vector<uint64_t> mc;
  for (uint64_t i = 0; i < n_folded; ++i) {
    if (f_groups[i] >= n_dim) return set_error(LLKV_INVALID_ARGUMENT, "folded straddler group out of range");
    if (f_first_rank[i] != rank) continue;
    mg.push_back(f_groups[i]); ms.push_back(f_sums[i]); mc.push_back(f_counts[i]);
  }
  DB dg, dsum, dcnt;
  if (!mg.empty()) {
    if ((rc = dg.alloc(mg.size() * 4)) || (rc = dsum.alloc(mg.size() * 8)) || (rc = dcnt.alloc(mg.size() * 8))) return rc;
    HIP_TRY(hipMemcpyAsync(dg.p, mg.data(), mg.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dsum.p, ms.data(), ms.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dcnt.p, mc.data(), mc.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hj_launch_patch_groups((const uint32_t *)dg.p, (const double *)dsum.p, (const uint64_t *)dcnt.p, mg.size(), (double *)sums.p, (uint64_t *)report.p, s));
    HIP_TRY(hipStreamSynchronize(s)); // the host vectors are pageable
  }
  const uint32_t np = n_payload;
  // ---- top-k by selection: the best keys of ≤ 1024 slices of the groups bound the LIMIT from below; the few groups
  // that reach the bound go to the host, which orders them exactly (sum DESC, payload[0], dim row).  More than kCap
  // of them (many equal sums) falls back to the sort below.
  if (by_selection) {
    constexpr uint32_t kCap = 960; // head + records fit the read-back slab
    DB best, groups_d;
    if ((rc = best.alloc(2 * kTopkSlices * 8)) || (rc = groups_d.alloc(kCap * 4))) return rc;
    // two launches; the second one's last workgroup writes the head and the records into the read-back slab itself and
    // carries the items a deferred prepare() left
    void *slab = nullptr;
    GatherItems extra;
    uint32_t *slab_base = nullptr;
    if ((rc = rb.reserve(nullptr, 64 + (size_t)kCap * 64, s, &slab)) || (rc = rb.take(&extra, &slab_base))) return rc;
    // (sums straight from the stripes, nothing patched in: the first launch's slice winners came with the run sums — if some
    // group turns out to have had two runs, everything is redone below)
    const bool winners_known = from_stripes && alone && mg.empty() && !sums_patched && (pending || !multi_run) && !std::getenv("LLKV_HIP_TOPK_TWO_LAUNCHES");
    HIP_TRY(hj_launch_topk_select2((const double *)sums.p, report_p, n_dim, std::max(1u, limit), kCap, d_dim_rows, cc, (uint64_t *)best.p, topk_state(),
                                   (uint32_t *)groups_d.p, (uint64_t *)slab, extra, slab_base, s, ranked && pending ? n_dim_ptr() : nullptr,
                                   winners_known ? slice_best() : nullptr));
    if ((rc = rb.wait())) return rc;
    if (pending) {
      if ((rc = settle(true))) return rc;
      if (multi_run || std::getenv("LLKV_HIP_JOIN_SORT")) // the sums were made again: select again
        return candidates(f_groups, f_sums, f_counts, f_first_rank, n_folded, rank, limit, out_rows, out_n, out_groups);
    }
    const uint64_t *head = static_cast<const uint64_t *>(slab);
    const uint32_t n_sel = (uint32_t)head[1];
    if (n_sel <= kCap) {
      std::vector<llkv_join_group_row> cand(n_sel);
      for (uint32_t i = 0; i < n_sel; ++i) {
        const uint64_t *c = head + 8 + (size_t)i * 8;
        llkv_join_group_row &g = cand[i];
        g.group_index = (uint32_t)c[0];
        g.key = (int64_t)c[1];
        std::memcpy(&g.sum, &c[2], 8);
        g.count = c[3];
        for (int k = 0; k < 4; ++k) g.payload[k] = (int64_t)c[4 + k];
      }
      std::sort(cand.begin(), cand.end(), [np](const llkv_join_group_row &a, const llkv_join_group_row &b) { return row_before(a, b, np); });
      const uint32_t n = std::min<uint32_t>(limit, n_sel);
      for (uint32_t i = 0; i < n; ++i) out_rows[i] = cand[i];
      *out_n = n;
      if (out_groups) *out_groups = head[2];
      return LLKV_OK;
    }
  }
  DB tk_keys, tk_groups, tk_keys_s, tk_groups_s, n_groups_d;
  if ((rc = tk_keys.alloc(n_dim * 8)) || (rc = tk_groups.alloc(n_dim * 4)) || (rc = tk_keys_s.alloc(n_dim * 8)) || (rc = tk_groups_s.alloc(n_dim * 4)) ||
      (rc = n_groups_d.alloc(8)))
    return rc;
  HIP_TRY(hipMemsetAsync(n_groups_d.p, 0, 8, s));
  HIP_TRY(hj_launch_topk_keys((const double *)sums.p, report_p, n_dim, (uint64_t *)tk_keys.p, (uint32_t *)tk_groups.p,
                              (unsigned long long *)n_groups_d.p, s));
  // ---- candidates → host: the first (limit + slack) groups by descending sum ------------------------------------
  // The top groups are decided by the high half of the order key almost always: sort on bits 32..63 first (half
  // the radix passes) and fall back to all 64 bits only when the cut falls inside a run of equal high halves.
  const uint32_t want = (uint32_t)std::min<uint64_t>(n_dim, (uint64_t)limit + 64);
  DB cand_d;
  if ((rc = cand_d.alloc((size_t)want * 64 + 8))) return rc;
  std::vector<uint64_t> hc((size_t)want * 8);
  std::vector<uint32_t> hg(want);
  uint64_t n_groups = 0;
  uint32_t n_cand = 0;
  DB hi, hi_s;
  if ((rc = hi.alloc(n_dim * 4)) || (rc = hi_s.alloc(n_dim * 4))) return rc;
  for (int pass = std::getenv("LLKV_HIP_TOPK_FULL") ? 1 : 0; pass < 2; ++pass) {
    DB tmp;
    size_t tb = 0;
    const uint32_t shift = pass == 0 ? 32 : 0;
    if (pass == 0) { // 32-bit keys = high halves of the order keys, (u32, u32) pairs: 4 radix passes instead of 8
      HIP_TRY(hj_launch_high_halves((const uint64_t *)tk_keys.p, n_dim, (uint32_t *)hi.p, s));
      HIP_TRY(hj_sort_by_slot(nullptr, &tb, (const uint32_t *)hi.p, (uint32_t *)hi_s.p, (const uint32_t *)tk_groups.p, (uint32_t *)tk_groups_s.p, (uint32_t)n_dim, 32, s));
      if ((rc = tmp.alloc(tb ? tb : 8))) return rc;
      HIP_TRY(hj_sort_by_slot(tmp.p, &tb, (const uint32_t *)hi.p, (uint32_t *)hi_s.p, (const uint32_t *)tk_groups.p, (uint32_t *)tk_groups_s.p, (uint32_t)n_dim, 32, s));
    } else {
      HIP_TRY(hj_sort_u64_u32(nullptr, &tb, (const uint64_t *)tk_keys.p, (uint64_t *)tk_keys_s.p, (const uint32_t *)tk_groups.p, (uint32_t *)tk_groups_s.p, n_dim, s));
      if ((rc = tmp.alloc(tb ? tb : 8))) return rc;
      HIP_TRY(hj_sort_u64_u32(tmp.p, &tb, (const uint64_t *)tk_keys.p, (uint64_t *)tk_keys_s.p, (const uint32_t *)tk_groups.p, (uint32_t *)tk_groups_s.p, n_dim, s));
    }
    HIP_TRY(hj_launch_gather_group_candidates(pass == 0 ? nullptr : (const uint64_t *)tk_keys_s.p, (const uint64_t *)tk_keys.p, (const uint32_t *)tk_groups_s.p, want,
                                              d_dim_rows, (const double *)sums.p, report_p, cc, (uint64_t *)cand_d.p, s));
    HIP_TRY(hipMemcpyAsync(hc.data(), cand_d.p, (size_t)want * 64, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hg.data(), tk_groups_s.p, (size_t)want * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&n_groups, n_groups_d.p, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    // candidates without a group (sentinel key) only trail the others on the full key; on the half key they may
    // be interleaved with a group: then — or when the cut falls inside a run of equal compared bits — decide on
    // the full key
    uint32_t first_sentinel = 0;
    while (first_sentinel < want && hc[(size_t)first_sentinel * 8] != ~0ull) ++first_sentinel;
    n_cand = first_sentinel;
    const bool cut_inside_tie = n_cand == want && (uint64_t)want < n_dim && limit > 0 && n_cand > limit &&
                                (hc[(size_t)(limit - 1) * 8] >> shift) == (hc[(size_t)(want - 1) * 8] >> shift);
    if (pass == 0 && (cut_inside_tie || n_cand < want)) continue;
    if (cut_inside_tie) return set_error(LLKV_UNSUPPORTED, "more than 64 groups tie on the LIMIT boundary");
    break;
  }
  std::vector<llkv_join_group_row> cand(n_cand);
  for (uint32_t i = 0; i < n_cand; ++i) {
    const uint64_t *c = &hc[(size_t)i * 8];
    llkv_join_group_row &g = cand[i];
    g.key = (int64_t)c[1];
    std::memcpy(&g.sum, &c[2], 8);
    g.count = c[3];
    for (int k = 0; k < 4; ++k) g.payload[k] = (int64_t)c[4 + k];
    g.group_index = hg[i];
  }
  std::sort(cand.begin(), cand.end(), [np](const llkv_join_group_row &a, const llkv_join_group_row &b) { return row_before(a, b, np); });
  const uint32_t n = std::min<uint32_t>(limit, n_cand);
  for (uint32_t i = 0; i < n; ++i) out_rows[i] = cand[i];
  *out_n = n;
  if (out_groups) *out_groups = n_groups;
  return LLKV_OK;
}

// ---- range form: boundary runs ------------------------------------------------------------------------------------------
int JoinAgg::boundary() {
  if (!range_form) return set_error(LLKV_INVALID_ARGUMENT, "not a ranged handle: take counts_buffer / straddlers / candidates");
  int rc;
  if (pending && (rc = settle())) return rc;
  boundary_block.assign(8 + 2 * kBoundaryCap, 0);
  boundary_block[0] = (multi_run || desc_run) ? 1 : 0;
  boundary_block[1] = n_pairs;
  if (n_pairs == 0 || n_dim == 0 || boundary_block[0]) return LLKV_OK;
  if (boundary_raw.size() != boundary_block.size()) return set_error(LLKV_INTERNAL, "boundary runs were not taken");
  const std::vector<uint64_t> &raw = boundary_raw;
  boundary_block[2] = raw[1]; boundary_block[3] = raw[2];
  boundary_block[4] = raw[4]; boundary_block[5] = raw[5];
  boundary_block[6] = raw[0]; boundary_block[7] = raw[3];
  std::copy(raw.begin() + 8, raw.end(), boundary_block.begin() + 8);
  return LLKV_OK;
}

int JoinAgg::finish_ranged(const uint64_t *blocks, const uint64_t *offsets, uint32_t world, uint32_t rank, uint32_t limit, llkv_join_group_row *out_rows,
                           uint32_t *out_n, uint64_t *out_groups) {
  if (!range_form) return set_error(LLKV_INVALID_ARGUMENT, "not a ranged handle");
  if (boundary_block.empty()) return set_error(LLKV_INVALID_ARGUMENT, "finish_ranged before boundary");
  const uint64_t words = 8 + 2 * kBoundaryCap;
  // every rank sees the same blocks and takes the same decisions
  struct Seg { uint64_t key; uint32_t rank; bool first; const uint64_t *vals; uint64_t n; };
  std::vector<Seg> segs;
  bool have_prev = false;
  uint64_t prev_last = 0;
  for (uint32_t r = 0; r < world; ++r) {
    if (offsets[r + 1] - offsets[r] != words * 8 || offsets[r] % 8) return set_error(LLKV_INVALID_ARGUMENT, "malformed boundary block");
    const uint64_t *b = blocks + offsets[r] / 8;
    if (b[0]) return set_error(LLKV_UNSUPPORTED, "range form: the pairs of rank " + std::to_string(r) + " are not in key order (the fact table is not clustered by the join key)");
    if (b[1] == 0) continue;
    if (b[3] > kBoundaryCap || b[5] > kBoundaryCap) return set_error(LLKV_UNSUPPORTED, "range form: a boundary group of rank " + std::to_string(r) + " has more than 64 rows");
    if (have_prev && b[2] < prev_last) return set_error(LLKV_UNSUPPORTED, "range form: the key ranges of the ranks' pairs overlap");
    segs.push_back({b[2], r, true, b + 8, b[3]});
    if (b[5]) segs.push_back({b[4], r, false, b + 8 + kBoundaryCap, b[5]});
    prev_last = b[5] ? b[4] : b[2];
    have_prev = true;
  }
  // a key that more than one rank holds: its exact sum in rank order = global row order (SumFloat64: 0.0, then +=)
  std::vector<uint32_t> pg;
  std::vector<double> ps;
  std::vector<uint64_t> pc;
  for (size_t i = 0; i < segs.size();) {
    size_t j = i + 1;
    while (j < segs.size() && segs[j].key == segs[i].key) ++j;
    if (j - i > 1) {
      double sum = 0.0;
      uint64_t cnt = 0;
      for (size_t k = i; k < j; ++k)
        for (uint64_t v = 0; v < segs[k].n; ++v) { double x; std::memcpy(&x, &segs[k].vals[v], 8); sum += x; ++cnt; }
      for (size_t k = i; k < j; ++k) {
        if (segs[k].rank != rank) continue;
        pg.push_back((uint32_t)(segs[k].first ? boundary_block[6] : boundary_block[7]));
        ps.push_back(sum);
        pc.push_back(k == i ? cnt : 0); // the first rank that holds the group reports it; the others drop it
      }
    }
    i = j;
  }
  int rc;
  hipStream_t s = g_ctx.stream;
  if (!pg.empty()) {
    DB dg, dsum, dcnt;
    if ((rc = dg.alloc(pg.size() * 4)) || (rc = dsum.alloc(pg.size() * 8)) || (rc = dcnt.alloc(pg.size() * 8))) return rc;
    HIP_TRY(hipMemcpyAsync(dg.p, pg.data(), pg.size() * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dsum.p, ps.data(), ps.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dcnt.p, pc.data(), pc.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hj_launch_patch_groups((const uint32_t *)dg.p, (const double *)dsum.p, (const uint64_t *)dcnt.p, pg.size(), (double *)sums.p, (uint64_t *)cnts.p, s));
    HIP_TRY(hipStreamSynchronize(s)); // the host vectors are pageable
    sums_patched = true; // (a group another rank reports has lost its count here: the slice winners the run sums left may name it)
  }
  *out_n = 0;
  if (out_groups) *out_groups = 0;
  if (n_dim == 0 || n_pairs == 0) return LLKV_OK;
  if ((rc = candidates(nullptr, nullptr, nullptr, nullptr, 0, rank, limit, out_rows, out_n, out_groups))) return rc;
  for (uint32_t i = 0; i < *out_n; ++i) out_rows[i].group_index = (uint64_t)(out_rows[i].key - dt.kmin); // the tie-break every rank agrees on: key order = dim row order
  return LLKV_OK;
}

} // namespace llkv

using namespace llkv;

static llkv_status finish_candidates(JoinAgg *j, const llkv_join_group_row *cand, uint32_t n_cand, uint64_t reported, uint32_t world, uint32_t limit,
                                     llkv_join_group_row *out_rows, uint32_t *out_n, uint64_t *out_total_groups);

extern "C" {

llkv_status llkv_hip_join_agg_prepare(const llkv_join_side *fact, const llkv_join_side *dim, uint32_t dim_fk_field, const llkv_join_side *dim2,
                                      const uint32_t *payload_fields, uint32_t n_payload, const llkv_expr_token *sum_expr, uint32_t sum_expr_len,
                                      llkv_hip_join_agg **out) {
  if (!out) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "out is NULL");
  auto *j = new JoinAgg();
  const int rc = j->prepare(fact, dim, dim_fk_field, dim2, payload_fields, n_payload, sum_expr, sum_expr_len);
  if (rc) { delete j; return (llkv_status)rc; }
  *out = reinterpret_cast<llkv_hip_join_agg *>(j);
  return LLKV_OK;
}

llkv_status llkv_hip_join_agg_prepare_ranged(const llkv_join_side *fact, const llkv_join_side *dim, uint32_t dim_fk_field, const llkv_join_side *dim2,
                                             const uint32_t *payload_fields, uint32_t n_payload, const llkv_expr_token *sum_expr, uint32_t sum_expr_len,
                                             llkv_hip_join_agg **out) {
  if (!out) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "out is NULL");
  if (std::getenv("LLKV_HIP_JOIN_NO_RANGE")) return (llkv_status)set_error(LLKV_UNSUPPORTED, "range form switched off (LLKV_HIP_JOIN_NO_RANGE)");
  auto *j = new JoinAgg();
  j->range_form = true;
  const int rc = j->prepare(fact, dim, dim_fk_field, dim2, payload_fields, n_payload, sum_expr, sum_expr_len);
  if (rc) { delete j; return (llkv_status)rc; }
  *out = reinterpret_cast<llkv_hip_join_agg *>(j);
  return LLKV_OK;
}

llkv_status llkv_hip_join_agg_boundary(llkv_hip_join_agg *h, const void **block, uint64_t *bytes) {
  auto *j = reinterpret_cast<JoinAgg *>(h);
  if (!j || !block || !bytes) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  const int rc = j->boundary();
  if (rc) return (llkv_status)rc;
  *block = j->boundary_block.data();
  *bytes = j->boundary_block.size() * 8;
  return LLKV_OK;
}

llkv_status llkv_hip_join_agg_finish_ranged(llkv_hip_join_agg *h, const void *blocks, const uint64_t *offsets, uint32_t world, uint32_t rank, uint32_t limit,
                                            llkv_join_group_row *out_rows, uint32_t *out_n, uint64_t *out_groups) {
  auto *j = reinterpret_cast<JoinAgg *>(h);
  if (!j || !blocks || !offsets || !out_rows || !out_n || world == 0 || rank >= world) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  if ((uintptr_t)blocks % 8) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "boundary blocks must be 8-byte aligned");
  return (llkv_status)j->finish_ranged(static_cast<const uint64_t *>(blocks), offsets, world, rank, limit, out_rows, out_n, out_groups);
}

void llkv_hip_join_agg_free(llkv_hip_join_agg *h) { delete reinterpret_cast<JoinAgg *>(h); }

llkv_status llkv_hip_join_agg_counts_buffer(llkv_hip_join_agg *h, void **device_ptr, uint64_t *len_i64) {
  auto *j = reinterpret_cast<JoinAgg *>(h);
  if (!j || !device_ptr || !len_i64) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  if (j->range_form) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "a ranged handle exchanges boundary runs, not counts: boundary / finish_ranged");
  *device_ptr = j->gcnts.p;
  *len_i64 = j->n_dim;
  return LLKV_OK;
}

llkv_status llkv_hip_join_agg_straddlers(llkv_hip_join_agg *h, const uint32_t **groups, const double **values, uint64_t *n) {
  auto *j = reinterpret_cast<JoinAgg *>(h);
  if (!j || !groups || !values || !n) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  if (j->range_form) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "a ranged handle exchanges boundary runs: boundary / finish_ranged");
  const int rc = j->straddlers();
  if (rc) return (llkv_status)rc;
  *groups = j->st_groups.data();
  *values = j->st_vals.data();
  *n = j->st_groups.size();
  return LLKV_OK;
}

// Host only: fold the straddler pairs of all ranks, concatenated in rank order (= global row order within
// each group), into exact per-group sums: SumFloat64 starts at 0.0 and adds in arrival order
// (llkv-aggregate/src/lib.rs:870-888).  Outputs are in order of first appearance; *n_out: capacity in, count out.
llkv_status llkv_hip_join_agg_fold_straddlers(const uint32_t *groups, const double *values, const uint64_t *rank_offsets, uint32_t world,
                                              uint32_t *out_groups, double *out_sums, uint64_t *out_counts, uint32_t *out_first_rank,
                                              uint64_t *n_out) {
  if (!rank_offsets || !n_out || world == 0) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  const uint64_t cap = *n_out;
  std::unordered_map<uint32_t, uint64_t> index;
  uint64_t n = 0;
  for (uint32_t r = 0; r < world; ++r) {
    if (rank_offsets[r + 1] < rank_offsets[r]) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "rank offsets must ascend");
    for (uint64_t i = rank_offsets[r]; i < rank_offsets[r + 1]; ++i) {
      auto it = index.find(groups[i]);
      if (it == index.end()) {
        if (n == cap) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "output capacity too small");
        it = index.emplace(groups[i], n).first;
        out_groups[n] = groups[i];
        out_sums[n] = 0.0;
        out_counts[n] = 0;
        out_first_rank[n] = r;
        ++n;
      }
      out_sums[it->second] += values[i];
      out_counts[it->second] += 1;
    }
  }
  *n_out = n;
  return LLKV_OK;
}

llkv_status llkv_hip_join_agg_candidates(llkv_hip_join_agg *h, const uint32_t *folded_groups, const double *folded_sums,
                                         const uint64_t *folded_counts, const uint32_t *folded_first_rank, uint64_t n_folded, uint32_t rank,
                                         uint32_t limit, llkv_join_group_row *out_rows, uint32_t *out_n, uint64_t *out_groups) {
  auto *j = reinterpret_cast<JoinAgg *>(h);
  if (!j || !out_rows || !out_n) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  if (n_folded && (!folded_groups || !folded_sums || !folded_counts || !folded_first_rank)) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL folded arrays");
  if (j->range_form) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "a ranged handle reports its candidates through finish_ranged");
  return (llkv_status)j->candidates(folded_groups, folded_sums, folded_counts, folded_first_rank, n_folded, rank, limit, out_rows, out_n, out_groups);
}

// Host only: ORDER BY sum DESC, payload[0] ASC, LIMIT over the ranks' candidates.
llkv_status llkv_hip_join_agg_merge(const llkv_join_group_row *rows, uint32_t n, uint32_t n_payload, uint32_t limit,
                                    llkv_join_group_row *out_rows, uint32_t *out_n) {
  if ((n && !rows) || !out_rows || !out_n) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  std::vector<llkv_join_group_row> v(rows, rows + n);
  std::sort(v.begin(), v.end(), [n_payload](const llkv_join_group_row &a, const llkv_join_group_row &b) { return row_before(a, b, n_payload); });
  const uint32_t m = std::min<uint32_t>(limit, n);
  for (uint32_t i = 0; i < m; ++i) out_rows[i] = v[i];
  *out_n = m;
  return LLKV_OK;
}

// Sharded fact table, collectives included (steps 2–6 of the phased form): the per-group row counts are all-reduced
// where they lie (HBM, RCCL), the straddler pairs and the ranks' candidates travel as small all-gathers.  Every rank
// returns the same rows.
llkv_status llkv_hip_join_agg_finish_sharded(llkv_hip_join_agg *h, uint32_t limit, llkv_join_group_row *out_rows, uint32_t *out_n,
                                             uint64_t *out_total_groups) {
  auto *j = reinterpret_cast<JoinAgg *>(h);
  if (!j || !out_rows || !out_n) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  if (!comm_ready()) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "no communicator: call llkv_hip_comm_init first");
  const uint32_t world = comm_world(), rank = comm_rank();
  if (j->tf->world != world || j->tf->rank != rank) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "the fact table's (rank, world) is not the communicator's");
  int rc;
  std::vector<uint64_t> off;
  std::vector<llkv_join_group_row> cand(std::max(1u, limit));
  uint32_t n_cand = 0;
  uint64_t reported = 0;
  if (j->range_form) { // one small all-gather of the boundary runs instead of the per-group counts and the straddler pairs
    if ((rc = j->boundary())) return (llkv_status)rc;
    std::vector<uint8_t> all;
    if ((rc = comm_allgather_v(j->boundary_block.data(), j->boundary_block.size() * 8, &all, &off))) return (llkv_status)rc;
    std::vector<uint64_t> aligned((all.size() + 7) / 8);
    std::memcpy(aligned.data(), all.data(), all.size());
    if ((rc = j->finish_ranged(aligned.data(), off.data(), world, rank, limit, cand.data(), &n_cand, &reported))) return (llkv_status)rc;
    j->last_exchange_bytes = all.size();
    return finish_candidates(j, cand.data(), n_cand, reported, world, limit, out_rows, out_n, out_total_groups);
  }
  if (j->n_dim && (rc = comm_allreduce_i64_device(static_cast<int64_t *>(j->gcnts.p), j->n_dim, g_ctx.stream))) return (llkv_status)rc;
  if ((rc = j->straddlers())) return (llkv_status)rc; // on the same stream: ordered behind the all-reduce
  // straddler pairs of every rank, rank order = global row order: [n][values f64 × n][groups u32 × n]
  const uint64_t n = j->st_groups.size();
  std::vector<uint8_t> mine(8 + n * 8 + (n * 4 + 7) / 8 * 8, 0), all;
  std::memcpy(mine.data(), &n, 8);
  if (n) {
    std::memcpy(mine.data() + 8, j->st_vals.data(), n * 8);
    std::memcpy(mine.data() + 8 + n * 8, j->st_groups.data(), n * 4);
  }
  if ((rc = comm_allgather_v(mine.data(), mine.size(), &all, &off))) return (llkv_status)rc;
  j->last_exchange_bytes = j->n_dim * 8 * 2 + all.size(); // the all-reduce moves the counts out and back
  std::vector<uint32_t> groups;
  std::vector<double> values;
  std::vector<uint64_t> rank_off(world + 1, 0);
  for (uint32_t r = 0; r < world; ++r) {
    uint64_t m = 0;
    std::memcpy(&m, all.data() + off[r], 8);
    if (off[r + 1] - off[r] != 8 + m * 8 + (m * 4 + 7) / 8 * 8) return (llkv_status)set_error(LLKV_INTERNAL, "malformed straddler block");
    const double *v = reinterpret_cast<const double *>(all.data() + off[r] + 8);
    const uint32_t *g = reinterpret_cast<const uint32_t *>(all.data() + off[r] + 8 + m * 8);
    values.insert(values.end(), v, v + m);
    groups.insert(groups.end(), g, g + m);
    rank_off[r + 1] = rank_off[r] + m;
  }
  uint64_t n_folded = std::max<uint64_t>(1, groups.size());
  std::vector<uint32_t> fg(n_folded), ffirst(n_folded);
  std::vector<double> fs(n_folded);
  std::vector<uint64_t> fc(n_folded);
  if ((rc = llkv_hip_join_agg_fold_straddlers(groups.data(), values.data(), rank_off.data(), world, fg.data(), fs.data(), fc.data(), ffirst.data(), &n_folded)))
    return (llkv_status)rc;
  if ((rc = j->candidates(fg.data(), fs.data(), fc.data(), ffirst.data(), n_folded, rank, limit, cand.data(), &n_cand, &reported))) return (llkv_status)rc;
  return finish_candidates(j, cand.data(), n_cand, reported, world, limit, out_rows, out_n, out_total_groups);
}

} // extern "C"

// candidates of every rank: [reported groups][n][rows] → all-gather → ORDER BY / LIMIT
static llkv_status finish_candidates(JoinAgg *j, const llkv_join_group_row *cand, uint32_t n_cand, uint64_t reported, uint32_t world, uint32_t limit,
                                     llkv_join_group_row *out_rows, uint32_t *out_n, uint64_t *out_total_groups) {
  int rc;
  std::vector<uint64_t> off;
  static_assert(sizeof(llkv_join_group_row) % 8 == 0, "rows travel as 8-byte words");
  std::vector<uint8_t> cmine(16 + (size_t)n_cand * sizeof(llkv_join_group_row)), call;
  const uint64_t head[2] = {reported, n_cand};
  std::memcpy(cmine.data(), head, 16);
  if (n_cand) std::memcpy(cmine.data() + 16, cand, (size_t)n_cand * sizeof(llkv_join_group_row));
  if ((rc = comm_allgather_v(cmine.data(), cmine.size(), &call, &off))) return (llkv_status)rc;
  j->last_exchange_bytes += call.size();
  std::vector<llkv_join_group_row> rows;
  uint64_t total = 0;
  for (uint32_t r = 0; r < world; ++r) {
    uint64_t hd[2];
    std::memcpy(hd, call.data() + off[r], 16);
    if (off[r + 1] - off[r] != 16 + hd[1] * sizeof(llkv_join_group_row)) return (llkv_status)set_error(LLKV_INTERNAL, "malformed candidate block");
    total += hd[0];
    const llkv_join_group_row *p = reinterpret_cast<const llkv_join_group_row *>(call.data() + off[r] + 16);
    rows.insert(rows.end(), p, p + hd[1]);
  }
  if (out_total_groups) *out_total_groups = total;
  return llkv_hip_join_agg_merge(rows.data(), (uint32_t)rows.size(), j->n_payload, limit, out_rows, out_n);
}

extern "C" {

/* bytes the collectives of the last llkv_hip_join_agg_finish_sharded call moved (every rank's contributions; bench only) */
uint64_t llkv_hip_join_agg_exchange_bytes(const llkv_hip_join_agg *h) { return h ? reinterpret_cast<const JoinAgg *>(h)->last_exchange_bytes : 0; }

// Single-rank form: prepare → (nothing to exchange) → candidates.
llkv_status llkv_hip_join_groupby_topk(const llkv_join_side *fact, const llkv_join_side *dim, uint32_t dim_fk_field,
                                       const llkv_join_side *dim2, const uint32_t *payload_fields, uint32_t n_payload,
                                       const llkv_expr_token *sum_expr, uint32_t sum_expr_len, uint32_t limit,
                                       llkv_join_group_row *out_rows, uint32_t *out_n, uint64_t *out_total_groups) {
  if (!out_rows || !out_n) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  if (fact && fact->table && reinterpret_cast<const Table *>(fact->table)->world != 1)
    return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "a sharded fact table needs the phased llkv_hip_join_agg_* calls (counts all-reduce, straddler exchange)");
  JoinAgg j;
  int rc = j.prepare(fact, dim, dim_fk_field, dim2, payload_fields, n_payload, sum_expr, sum_expr_len, true);
  if (rc) return (llkv_status)rc;
  return (llkv_status)j.candidates(nullptr, nullptr, nullptr, nullptr, 0, 0, limit, out_rows, out_n, out_total_groups);
}

} // extern "C"
