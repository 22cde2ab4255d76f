// join.cpp — TableJoinExt::join_stream (llkv-join/src/lib.rs:240-282) over two HBM table
// images: validation as the reference's (`validate_join_options`), build on the RIGHT table
// (hash_join.rs:209-215), probe the LEFT in scan order, pairs delivered in probe order × build
// insertion order in batches that follow the reference's flush rule (a batch ends after the probe
// row that brings it to ≥ batch_size pairs, and at the end of every probe window,
// hash_join.rs:1181-1213).
#include "engine.hpp"
#include "join.hpp"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

namespace llkv {

namespace {
struct DBuf {
  void *p = nullptr;
  size_t cap = 0;
  ~DBuf() { scratch_free(p); }
  int ensure(size_t bytes) {
    if (bytes <= cap && p) return LLKV_OK;
    scratch_free(p);
    p = scratch_alloc(bytes ? bytes : 8);
    if (!p) return set_error(LLKV_INTERNAL, "device scratch allocation failed");
    cap = bytes;
    return LLKV_OK;
  }
};
struct HBuf { // pinned, recycled (engine.cpp: pinned_acquire)
  void *p = nullptr;
  size_t cap = 0;
  ~HBuf() { if (p) pinned_release(p, cap); }
  int ensure(size_t bytes) {
    if (bytes <= cap) return LLKV_OK;
    if (p) pinned_release(p, cap);
    cap = bytes ? bytes : 8;
    p = pinned_acquire(&cap);
    if (!p) { cap = 0; return set_error(LLKV_INTERNAL, "pinned host allocation failed"); }
    return LLKV_OK;
  }
};

constexpr uint32_t kJoinTileRows = 8192;
// probe rows per device step: 32 reference scan batches.  (One batch per step — 8 workgroups, two synchronisations,
// three small copies — ran at 0.13 G probe rows/s.)
constexpr uint32_t kWindowTiles = 256;

bool fast_key_type(int32_t dt) { return dt == LLKV_DT_INT32 || dt == LLKV_DT_INT64 || dt == LLKV_DT_UINT32 || dt == LLKV_DT_UINT64; }
// the types extract_key_value (hash_join.rs:405-505) turns into a KeyValue; any other key type fails there
bool generic_key_type(int32_t dt) { return fast_key_type(dt) || dt == LLKV_DT_FLOAT32 || dt == LLKV_DT_FLOAT64 || dt == LLKV_DT_UTF8; }

int key_part(const Table *t, uint32_t field, const DeviceColumn **col, JoinKeyPart *out) {
  auto it = t->cols.find(field);
  if (it == t->cols.end()) return set_error(LLKV_NOT_FOUND, "join key field " + std::to_string(field) + " not found");
  const DeviceColumn &c = it->second;
  if (c.info.wide128) return set_error(LLKV_UNSUPPORTED, "join key over Decimal128 values beyond 64 bits (field " + std::to_string(field) + ")");
  *col = &c;
  std::memset(out, 0, sizeof *out);
  out->values = c.d_values;
  out->valid = c.info.nullable ? c.d_valid : nullptr;
  out->width = c.info.dtype == LLKV_DT_UTF8 ? 1u : dtype_width(c.info.dtype);
  out->is_signed = c.info.dtype == LLKV_DT_INT32 ? 1u : 0u;
  return LLKV_OK;
}

long long fast_null_sentinel(int32_t dt) { // hash_join.rs:1429-1465
  switch (dt) {
  case LLKV_DT_INT64: return INT64_MIN;
  case LLKV_DT_UINT64: return (long long)UINT64_MAX;
  case LLKV_DT_INT32: return INT32_MIN;
  default: return (long long)UINT32_MAX;
  }
}

// What both deliveries of a join (index pairs: run_join; record batches: run_join_batches) share: the validated
// options, the canonical key parts of the two sides and the build side's hash table.
struct JoinPlan {
  bool executor = false, fast = false, left_only = false;
  uint64_t batch_size = 8192;
  int jt = LLKV_JOIN_INNER;
  JoinKeySet lk, rk;
  DBuf translate;
  uint64_t n_build = 0, cap = 0;
  DBuf owner, slot_of, dev_of, log_of, seg_start, seg_count, idx_in, slot_sorted, idx_sorted, tile_base, tmp;
  const TileSet *tr = nullptr, *tl = nullptr;
  JoinPlan() { std::memset(&lk, 0, sizeof lk); std::memset(&rk, 0, sizeof rk); }
};

// validate_join_options llkv-join/src/lib.rs:284-310, hash_join.rs:328-332; the executor's rules :12387-12391
static int join_options(const llkv_join_options *options, uint32_t n_keys, JoinPlan *jp) {
  jp->executor = options && options->key_rules == LLKV_JOIN_KEYS_EXECUTOR;
  if (options && options->key_rules != LLKV_JOIN_KEYS_TABLE && !jp->executor) return set_error(LLKV_INVALID_ARGUMENT, "unknown join key rules");
  // executor rules: no batch cuts (hash_join_table_batches materialises one batch)
  jp->batch_size = jp->executor ? UINT64_MAX : options ? options->batch_size : 8192;
  const int jt = jp->jt = options ? options->join_type : LLKV_JOIN_INNER;
  if (jp->executor && jt != LLKV_JOIN_INNER && jt != LLKV_JOIN_LEFT)
    return set_error(LLKV_INTERNAL, "join type not supported in hash_join_table_batches; use llkv-join"); // llkv-executor/src/lib.rs:12387-12391
  if (jp->executor && n_keys == 0) return set_error(LLKV_INVALID_ARGUMENT, "executor join rules need at least one key pair");
  if (jp->batch_size == 0) return set_error(LLKV_INVALID_ARGUMENT, "join batch_size must be greater than zero");
  if (jt == LLKV_JOIN_RIGHT || jt == LLKV_JOIN_FULL) return set_error(LLKV_INVALID_ARGUMENT, "Right and Full joins are not yet implemented");
  if (jt != LLKV_JOIN_INNER && jt != LLKV_JOIN_LEFT && jt != LLKV_JOIN_SEMI && jt != LLKV_JOIN_ANTI) return set_error(LLKV_INVALID_ARGUMENT, "unknown join type");
  jp->left_only = jt == LLKV_JOIN_SEMI || jt == LLKV_JOIN_ANTI;
  return LLKV_OK;
}

// The canonical key parts of both sides (JoinKeyPart, join.hpp) under the rules the key list selects
static int join_key_setup(const Table *left, const Table *right, const llkv_join_key *keys, uint32_t n_keys, JoinPlan *jp) {
  int rc;
  const bool executor = jp->executor;
  // "build side replicated, probe side sharded" (BASELINE.json configs[4]): every rank probes its own rows of the left
  // table against the whole right table; the binding concatenates the ranks' batches in rank order
  if (right->world != 1) return set_error(LLKV_INVALID_ARGUMENT, "the build (right) table of a join is replicated: stage it whole (world = 1) on every rank");
  if (n_keys > kMaxJoinKeys) return set_error(LLKV_UNSUPPORTED, "GPU join path takes at most " + std::to_string(kMaxJoinKeys) + " key pairs");
  if (!keys) return set_error(LLKV_INVALID_ARGUMENT, "join keys is NULL");
  JoinKeySet &lk = jp->lk, &rk = jp->rk;
  DBuf &translate = jp->translate;
  lk.n = rk.n = n_keys;
  const DeviceColumn *lc[kMaxJoinKeys], *rc_[kMaxJoinKeys];
  for (uint32_t i = 0; i < n_keys; ++i)
    if ((rc = key_part(left, keys[i].left_field, &lc[i], &lk.k[i])) || (rc = key_part(right, keys[i].right_field, &rc_[i], &rk.k[i]))) return rc;
  // one key of one fast integer type on both sides → the integer fast path; anything else → the generic
  // typed-key path (hash_join.rs:171-200), with its own NULL rule and its own batching
  const bool fast = jp->fast = !executor && n_keys == 1 && lc[0]->info.dtype == rc_[0]->info.dtype && fast_key_type(lc[0]->info.dtype);
  if (executor) {
    // normalize_join_column + arrow-row bytes: equal only inside one class; NULL parts never match
    std::vector<uint16_t> tables((size_t)n_keys * 256, 0xFFFFu);
    bool any_table = false;
    auto klass = [](int32_t dt) {
      switch (dt) {
      case LLKV_DT_BOOLEAN: case LLKV_DT_INT32: case LLKV_DT_UINT32: case LLKV_DT_INT64: case LLKV_DT_UINT64: return 1;
      case LLKV_DT_FLOAT32: case LLKV_DT_FLOAT64: return 2;
      case LLKV_DT_UTF8: return 3;
      case LLKV_DT_DATE32: return 4;
      case LLKV_DT_DECIMAL128: return 5;
      default: return 0;
      }
    };
    for (uint32_t i = 0; i < n_keys; ++i) {
      const int32_t dts[2] = {lc[i]->info.dtype, rc_[i]->info.dtype};
      JoinKeyPart *parts[2] = {&lk.k[i], &rk.k[i]};
      const int kl = klass(dts[0]), kr = klass(dts[1]);
      if (kl == 0 || kr == 0) return set_error(LLKV_UNSUPPORTED, std::string("join key of type ") + dtype_name(kl == 0 ? dts[0] : dts[1]));
      for (int side = 0; side < 2; ++side) {
        JoinKeyPart &k = *parts[side];
        k.values_never_match = kl != kr;
        k.is_signed = dts[side] == LLKV_DT_INT32 || dts[side] == LLKV_DT_DATE32;
        k.f32_as_f64 = dts[side] == LLKV_DT_FLOAT32;
        k.u64_high_is_null = dts[side] == LLKV_DT_UINT64;
      }
      if (kl == 3 && kr == 3) {
        uint16_t *tab = tables.data() + (size_t)i * 256;
        const std::vector<std::string> &ldict = lc[i]->info.dictionary, &rdict = rc_[i]->info.dictionary;
        for (size_t c = 0; c < ldict.size() && c < 256; ++c)
          for (size_t d = 0; d < rdict.size(); ++d)
            if (rdict[d] == ldict[c]) tab[c] = (uint16_t)d;
        any_table = true;
        lk.k[i].translate = reinterpret_cast<const uint16_t *>(1); // patched below
      }
    }
    if (any_table) {
      if ((rc = translate.ensure(tables.size() * 2))) return rc;
      HIP_TRY(hipMemcpyAsync(translate.p, tables.data(), tables.size() * 2, hipMemcpyHostToDevice, g_ctx.stream));
      HIP_TRY(hipStreamSynchronize(g_ctx.stream)); // `tables` is pageable host memory
      for (uint32_t i = 0; i < n_keys; ++i)
        if (lk.k[i].translate) lk.k[i].translate = (const uint16_t *)translate.p + (size_t)i * 256;
    }
  } else if (fast) {
    for (JoinKeyPart *k : {&lk.k[0], &rk.k[0]}) {
      k->null_equals_null = keys[0].null_equals_null != 0;
      k->null_is_value = 1;
      k->null_value = fast_null_sentinel(lc[0]->info.dtype);
    }
  } else {
    std::vector<uint16_t> tables((size_t)n_keys * 256, 0xFFFFu);
    bool any_table = false;
    for (uint32_t i = 0; i < n_keys; ++i) {
      const int32_t ldt = lc[i]->info.dtype, rdt = rc_[i]->info.dtype;
      JoinKeyPart &l = lk.k[i], &r = rk.k[i];
      const bool null_eq = keys[i].null_equals_null != 0;
      l.null_equals_null = r.null_equals_null = null_eq;
      l.unusable = !generic_key_type(ldt);
      r.unusable = !generic_key_type(rdt);
      if (ldt != LLKV_DT_UTF8 && rdt != LLKV_DT_UTF8) {
        l.values_never_match = r.values_never_match = ldt != rdt; // NULLs still meet through the `nulls` bit
        continue;
      }
      // at least one Utf8 side: keys live in the build column's code space; the NULL marker is the string
      // "<NULL>" (hash_join.rs:391-396) — the build dictionary's code for it, or 256 when it has none
      const std::vector<std::string> none;
      const std::vector<std::string> &rdict = rdt == LLKV_DT_UTF8 ? rc_[i]->info.dictionary : none;
      long long null_code = 256;
      for (size_t c = 0; c < rdict.size(); ++c)
        if (rdict[c] == "<NULL>") null_code = (long long)c;
      l.null_is_value = r.null_is_value = 1;
      l.null_value = r.null_value = null_code;
      if (rdt != LLKV_DT_UTF8) r.values_never_match = 1;
      if (ldt != LLKV_DT_UTF8) { l.values_never_match = 1; continue; }
      uint16_t *tab = tables.data() + (size_t)i * 256;
      const std::vector<std::string> &ldict = lc[i]->info.dictionary;
      for (size_t c = 0; c < ldict.size() && c < 256; ++c) {
        for (size_t d = 0; d < rdict.size(); ++d)
          if (rdict[d] == ldict[c]) tab[c] = (uint16_t)d;
        if (tab[c] == 0xFFFFu && null_eq && ldict[c] == "<NULL>") tab[c] = (uint16_t)null_code;
      }
      any_table = true;
      l.translate = reinterpret_cast<const uint16_t *>(1); // patched below
    }
    if (any_table) {
      if ((rc = translate.ensure(tables.size() * 2))) return rc;
      HIP_TRY(hipMemcpyAsync(translate.p, tables.data(), tables.size() * 2, hipMemcpyHostToDevice, g_ctx.stream));
      HIP_TRY(hipStreamSynchronize(g_ctx.stream)); // `tables` is pageable host memory
      for (uint32_t i = 0; i < n_keys; ++i)
        if (lk.k[i].translate) lk.k[i].translate = (const uint16_t *)translate.p + (size_t)i * 256;
    }
  }

  return LLKV_OK;
}

// Build (right): claim-insert, stable slot sort (insertion order inside a key), segments
static int join_build(const Table *left, const Table *right, JoinPlan *jp, hipStream_t s) {
  int rc;
  if ((rc = get_tileset(*right, kJoinTileRows, &jp->tr)) || (rc = get_tileset(*left, kJoinTileRows, &jp->tl))) return rc;
  const TileSet *tr = jp->tr;
  const uint64_t n_build = jp->n_build = right->local_rows;
  if (n_build >= (1ull << 31)) return set_error(LLKV_UNSUPPORTED, "build side larger than 2^31 rows");
  uint64_t cap = 1024;
  uint32_t bits = 10;
  while (cap < 2 * n_build) { cap <<= 1; ++bits; }
  jp->cap = cap;
  DBuf &owner = jp->owner, &slot_of = jp->slot_of, &dev_of = jp->dev_of, &log_of = jp->log_of, &seg_start = jp->seg_start, &seg_count = jp->seg_count,
       &idx_in = jp->idx_in, &slot_sorted = jp->slot_sorted, &idx_sorted = jp->idx_sorted, &tile_base = jp->tile_base, &tmp = jp->tmp;
  if ((rc = owner.ensure(cap * 8)) || (rc = seg_start.ensure((cap + 1) * 4)) || (rc = seg_count.ensure((cap + 1) * 4)) ||
      (rc = slot_of.ensure(n_build * 4)) || (rc = dev_of.ensure(n_build * 8)) || (rc = log_of.ensure(n_build * 8)) ||
      (rc = idx_in.ensure(n_build * 4)) || (rc = slot_sorted.ensure(n_build * 4)) || (rc = idx_sorted.ensure(n_build * 4)) ||
      (rc = tile_base.ensure((size_t)(tr->n_tiles + 1) * 8)))
    return rc;
  HIP_TRY(hipMemsetAsync(owner.p, 0xFF, cap * 8, s));
  HIP_TRY(hipMemsetAsync(seg_count.p, 0, (cap + 1) * 4, s)); // + the slot that parks NULL build keys
  if (n_build) {
    std::vector<TileDesc> tiles;
    uint32_t otb[kOctantsHost + 1];
    build_tiles_host(*right, kJoinTileRows, tiles, otb);
    std::vector<uint64_t> base(tiles.size() + 1, 0);
    for (size_t i = 0; i < tiles.size(); ++i) base[i + 1] = base[i] + tiles[i].rows;
    HIP_TRY(hipMemcpyAsync(tile_base.p, base.data(), base.size() * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s)); // `base` is pageable host memory
    HIP_TRY(hj_launch_claim(jp->rk, tr->d_tiles, tr->n_tiles, kJoinTileRows, (unsigned long long *)owner.p, cap - 1, (uint32_t *)slot_of.p,
                            (uint64_t *)dev_of.p, (uint64_t *)log_of.p, (const uint64_t *)tile_base.p, s));
    HIP_TRY(hj_launch_iota((uint32_t *)idx_in.p, (uint32_t)n_build, s));
    size_t tmp_bytes = 0;
    HIP_TRY(hj_sort_by_slot(nullptr, &tmp_bytes, (const uint32_t *)slot_of.p, (uint32_t *)slot_sorted.p, (const uint32_t *)idx_in.p,
                            (uint32_t *)idx_sorted.p, (uint32_t)n_build, bits + 1, s));
    if ((rc = tmp.ensure(tmp_bytes))) return rc;
    HIP_TRY(hj_sort_by_slot(tmp.p, &tmp_bytes, (const uint32_t *)slot_of.p, (uint32_t *)slot_sorted.p, (const uint32_t *)idx_in.p,
                            (uint32_t *)idx_sorted.p, (uint32_t)n_build, bits + 1, s));
    HIP_TRY(hj_launch_segments((const uint32_t *)slot_sorted.p, (uint32_t)n_build, (uint32_t *)seg_start.p, (uint32_t *)seg_count.p, s));
  }
  return LLKV_OK;
}
} // namespace

int run_join(const Table *left, const Table *right, const llkv_join_key *keys, uint32_t n_keys,
             const llkv_join_options *options, llkv_on_join_batch on_batch, void *user) {
  JoinPlan jp;
  int rc = join_options(options, n_keys, &jp);
  if (rc) return rc;
  const bool executor = jp.executor;
  const uint64_t batch_size = jp.batch_size;
  const int jt = jp.jt;
  if ((rc = ensure_device())) return rc;
  if (!left || !right || !on_batch) return set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  if (left->d_row_ids || right->d_row_ids)
    return set_error(LLKV_UNSUPPORTED, "index pairs over a table whose row ids are not its positions (llkv_hip_join_stream_batches takes it)");
  if (n_keys == 0) {
    // Empty join keys = Cartesian product (cross_product_stream, llkv-join/src/hash_join.rs:1500-1599): every
    // 65 536-row scan window of the left against every window of the right, left-major inside a pair of windows;
    // LEFT with an empty right side pads; SEMI / ANTI trip the reference's schema check.
    if (jt == LLKV_JOIN_SEMI || jt == LLKV_JOIN_ANTI) return set_error(LLKV_INTERNAL, "cross join schema mismatch: semi / anti joins deliver left columns only");
    const uint64_t nl = left->local_rows, nr = right->local_rows;
    if (left->world != 1 || right->world != 1) return set_error(LLKV_UNSUPPORTED, "cross product over sharded tables");
    if (nr == 0 && jt == LLKV_JOIN_INNER) return LLKV_OK;
    constexpr uint64_t kWin = 65536;
    hipStream_t s = g_ctx.stream;
    DBuf d_l, d_r;
    HBuf h_l, h_r;
    for (uint64_t l0 = 0; l0 < nl; l0 += kWin) {
      const uint64_t ln = std::min(kWin, nl - l0);
      if (nr == 0) { // LEFT: NULL-padded right side (synthesize_left_join_nulls)
        if ((rc = h_l.ensure(ln * 8)) || (rc = h_r.ensure(ln * 8))) return rc;
        uint64_t *hl = (uint64_t *)h_l.p, *hr = (uint64_t *)h_r.p;
        for (uint64_t i = 0; i < ln; ++i) { hl[i] = left->local_logical_start + l0 + i; hr[i] = ~0ull; }
        on_batch(hl, hr, ln, user);
        continue;
      }
      for (uint64_t r0 = 0; r0 < nr; r0 += kWin) {
        const uint64_t rn = std::min(kWin, nr - r0), np = ln * rn;
        if (np > (1ull << 28)) return set_error(LLKV_UNSUPPORTED, "cross product batch of more than 2^28 pairs");
        if ((rc = d_l.ensure(np * 8)) || (rc = d_r.ensure(np * 8)) || (rc = h_l.ensure(np * 8)) || (rc = h_r.ensure(np * 8))) return rc;
        HIP_TRY(hj_launch_cross_pairs(l0, ln, r0, rn, (uint64_t *)d_l.p, (uint64_t *)d_r.p, s));
        HIP_TRY(hipMemcpyAsync(h_l.p, d_l.p, np * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(h_r.p, d_r.p, np * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        on_batch((const uint64_t *)h_l.p, (const uint64_t *)h_r.p, np, user);
      }
    }
    return LLKV_OK;
  }
  if ((rc = join_key_setup(left, right, keys, n_keys, &jp))) return rc;
  const bool fast = jp.fast;
  const JoinKeySet &lk = jp.lk, &rk = jp.rk;

  hipStream_t s = g_ctx.stream;
  // LLKV_HIP_TRACE=1: phase times on stderr
  const bool trace = std::getenv("LLKV_HIP_TRACE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  double t_acc[4] = {0, 0, 0, 0}; // probe steps: count+scan, write+copy, cuts+callbacks
  auto lap = [&]() {
    const auto now = std::chrono::steady_clock::now();
    const double ms = std::chrono::duration<double, std::milli>(now - t_last).count();
    t_last = now;
    return ms;
  };
  // ---- build (right) ----
  if ((rc = join_build(left, right, &jp, s))) return rc;
  const TileSet *tl = jp.tl;
  const uint64_t n_build = jp.n_build, cap = jp.cap;
  DBuf &owner = jp.owner, &log_of = jp.log_of, &seg_start = jp.seg_start, &seg_count = jp.seg_count, &idx_sorted = jp.idx_sorted;

  if (trace) { (void)hipStreamSynchronize(s); std::fprintf(stderr, "[llkv join] build %9.3f ms (%llu rows)\n", lap(), (unsigned long long)n_build); }
  // ---- probe (left), window by window ----
  const uint32_t win_pos = std::min(kWindowTiles, std::max(1u, tl->n_tiles)) * kJoinTileRows;
  DBuf counts, mslot, offsets, scan_tmp;
  if ((rc = counts.ensure((size_t)(win_pos + 1) * 8)) || (rc = mslot.ensure((size_t)win_pos * 4)) || (rc = offsets.ensure((size_t)(win_pos + 1) * 8))) return rc;
  const bool left_only = jp.left_only;
  // Batches.  The reference probes one scan batch (65 536 rows of the left table) at a time and flushes after the
  // probe row that brings the pending pairs to ≥ batch_size, and at the end of the scan batch (fast path,
  // hash_join.rs:1141-1213); the generic path first cuts every scan batch into slices of batch_size probe rows
  // and applies the same rule inside each slice (:228-246,509-565).  A device step covers many scan batches; the
  // host finds the cuts in the left-row column of the pairs (ascending): a forced cut before the first pair of a
  // row ≥ the boundary, a size cut after the last pair of the row that holds the batch_size-th pair.  Pairs after the
  // last cut of a step wait in `pend_*` for the next one.
  constexpr uint64_t kRefWindow = 65536;
  std::vector<TileDesc> ltiles;
  {
    uint32_t otb[kOctantsHost + 1];
    build_tiles_host(*left, kJoinTileRows, ltiles, otb);
  }
  const uint64_t left_end = left->local_logical_start + left->local_rows;
  std::vector<uint64_t> pend_l, pend_r;
  // Two pair buffers: while the pairs of step i cross PCIe on the copy stream and the host cuts step i − 1 into
  // batches, the compute stream already counts step i + 1.
  struct Step {
    DBuf out_l, out_r;
    HBuf h_l, h_r;
    hipEvent_t written = nullptr, copied = nullptr;
    uint64_t total = 0, L0 = 0, L1 = 0;
    bool live = false;
    ~Step() { // the buffers go back to their pools: nothing may still be writing them (error paths leave early)
      if (written) { (void)hipEventSynchronize(written); (void)hipEventDestroy(written); }
      if (copied) { (void)hipEventSynchronize(copied); (void)hipEventDestroy(copied); }
    }
  } steps[2];
  for (Step &st : steps) {
    HIP_TRY(hipEventCreateWithFlags(&st.written, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&st.copied, hipEventDisableTiming));
  }
  struct CopyStream {
    hipStream_t s = nullptr;
    ~CopyStream() { if (s) (void)hipStreamDestroy(s); }
  } copy;
  HIP_TRY(hipStreamCreateWithFlags(&copy.s, hipStreamNonBlocking));

  // cuts one finished step into the reference's batches
  auto emit = [&](Step &st) -> int {
    if (!st.live) return LLKV_OK;
    st.live = false;
    HIP_TRY(hipEventSynchronize(st.copied));
    const uint64_t total = st.total, L0 = st.L0, L1 = st.L1;
    const uint64_t *hl = (const uint64_t *)st.h_l.p, *hr = left_only ? nullptr : (const uint64_t *)st.h_r.p;
    uint64_t start = 0; // pairs of this step already delivered
    auto deliver = [&](uint64_t end) {
      if (pend_l.empty()) {
        on_batch(hl + start, hr ? hr + start : nullptr, end - start, user);
      } else {
        pend_l.insert(pend_l.end(), hl + start, hl + end);
        if (hr) pend_r.insert(pend_r.end(), hr + start, hr + end);
        on_batch(pend_l.data(), hr ? pend_r.data() : nullptr, pend_l.size(), user);
        pend_l.clear();
        pend_r.clear();
      }
      start = end;
    };
    if (executor) { // no batch structure to reproduce: one callback per device step
      if (total) deliver(total);
      return LLKV_OK;
    }
    for (uint64_t row = L0; row < L1;) {
      // the next forced cut: end of the reference scan batch, of the slice (generic path), of the table
      const uint64_t in_win = row % kRefWindow;
      uint64_t b = row - in_win + kRefWindow;
      if (!fast) b = std::min(b, row - in_win + (in_win / batch_size + 1) * batch_size);
      b = std::min(b, left_end);
      const uint64_t seg_end = (uint64_t)(std::lower_bound(hl + start, hl + total, b) - hl); // first pair of a row ≥ b
      while (pend_l.size() + (seg_end - start) >= batch_size) {
        const uint64_t j = start + (batch_size - pend_l.size()) - 1; // the pair that fills the batch …
        deliver((uint64_t)(std::upper_bound(hl + j, hl + seg_end, hl[j]) - hl)); // … and the rest of its probe row
      }
      if (b <= L1 && pend_l.size() + (seg_end - start) > 0) deliver(seg_end);
      row = b;
    }
    if (total > start) { // the reference batch goes on in the next step
      pend_l.insert(pend_l.end(), hl + start, hl + total);
      if (hr) pend_r.insert(pend_r.end(), hr + start, hr + total);
    }
    return LLKV_OK;
  };

  int cur = 0;
  // a many-to-many key can turn one step into billions of pairs: steps shrink until their pairs fit kMaxStepPairs
  constexpr uint64_t kMaxStepPairs = 64ull << 20; // 1 GiB of row-id pairs per buffer
  uint32_t step_tiles = kWindowTiles;
  for (uint32_t t0 = 0, nt = 0; t0 < tl->n_tiles; t0 += nt) {
    nt = std::min(step_tiles, tl->n_tiles - t0);
    const uint32_t npos = nt * kJoinTileRows;
    ProbeParams p;
    std::memset(&p, 0, sizeof p);
    p.lkey = lk; p.rkey = rk;
    p.tiles = tl->d_tiles + t0; p.n_tiles = nt; p.tile_rows = kJoinTileRows;
    p.slot_owner = (const unsigned long long *)owner.p; p.cap_mask = cap - 1;
    p.seg_start = (const uint32_t *)seg_start.p; p.seg_count = (const uint32_t *)seg_count.p;
    p.sorted_idx = (const uint32_t *)idx_sorted.p; p.build_logical = (const uint64_t *)log_of.p;
    p.join_type = jt;
    p.counts = (uint64_t *)counts.p; p.match_slot = (uint32_t *)mslot.p;
    HIP_TRY(hj_launch_probe_count(p, s));
    // counts[npos] = 0, so offsets[npos] is the total (the one-workgroup scan of the selection kernels is too slow
    // for millions of positions)
    HIP_TRY(hipMemsetAsync((uint64_t *)counts.p + npos, 0, 8, s));
    {
      size_t tb = 0;
      HIP_TRY(hj_exclusive_scan_u64(nullptr, &tb, (const uint64_t *)counts.p, (uint64_t *)offsets.p, (uint64_t)npos + 1, s));
      if ((rc = scan_tmp.ensure(tb))) return rc;
      HIP_TRY(hj_exclusive_scan_u64(scan_tmp.p, &tb, (const uint64_t *)counts.p, (uint64_t *)offsets.p, (uint64_t)npos + 1, s));
    }
    uint64_t total = 0;
    Readback rb;
    if ((rc = rb.add(&total, (uint64_t *)offsets.p + npos, 8, s)) || (rc = rb.wait())) return rc;
    if (trace) t_acc[0] += lap();
    if (total > kMaxStepPairs && nt > 1) { // count again over fewer tiles
      step_tiles = std::max(1u, nt / 2);
      nt = 0;
      continue;
    }
    Step &st = steps[cur];
    if ((rc = emit(st))) return rc; // its buffers are about to be reused (normally already emitted below)
    uint64_t wrows = 0;
    for (uint32_t t = 0; t < nt; ++t) wrows += ltiles[t0 + t].rows;
    st.total = total;
    st.L0 = ltiles[t0].logical_row; // the rows of a rank are contiguous
    st.L1 = st.L0 + wrows;
    st.live = true;
    if (total) {
      if ((rc = st.out_l.ensure(total * 8)) || (rc = st.out_r.ensure(total * 8)) || (rc = st.h_l.ensure(total * 8)) || (rc = st.h_r.ensure(total * 8))) return rc;
      p.offsets = (const uint64_t *)offsets.p;
      p.out_left = (uint64_t *)st.out_l.p; p.out_right = (uint64_t *)st.out_r.p;
      HIP_TRY(hj_launch_probe_write(p, s));
      HIP_TRY(hipEventRecord(st.written, s));
      HIP_TRY(hipStreamWaitEvent(copy.s, st.written, 0));
      HIP_TRY(hipMemcpyAsync(st.h_l.p, st.out_l.p, total * 8, hipMemcpyDeviceToHost, copy.s));
      if (!left_only) HIP_TRY(hipMemcpyAsync(st.h_r.p, st.out_r.p, total * 8, hipMemcpyDeviceToHost, copy.s));
    }
    HIP_TRY(hipEventRecord(st.copied, copy.s));
    if (trace) t_acc[1] += lap();
    if ((rc = emit(steps[cur ^ 1]))) return rc; // the previous step, while this one's pairs are on their way
    if (trace) t_acc[2] += lap();
    cur ^= 1;
  }
  if ((rc = emit(steps[cur])) || (rc = emit(steps[cur ^ 1]))) return rc; // oldest first
  if (!pend_l.empty()) { // pairs carried over a device step that ended inside a reference batch: the last batch
    on_batch(pend_l.data(), left_only ? nullptr : pend_r.data(), pend_l.size(), user);
    pend_l.clear();
    pend_r.clear();
  }
  if (trace) std::fprintf(stderr, "[llkv join] probe: count+scan %9.3f ms, write+copy %9.3f ms, cuts+callbacks %9.3f ms\n", t_acc[0], t_acc[1], t_acc[2]);
  return LLKV_OK;
}

// ====================================================================================================================
// run_join_batches — the same joins, delivering the reference's RecordBatches: the matched rows' columns are gathered on
// the device (emit_joined_batch / emit_left_joined_batch / emit_semi_batch hash_join.rs:715-772, cross_join_pair
// cartesian.rs:22-110, synthesize_left_join_nulls hash_join.rs:1468-1497) and cross PCIe once, as column values.
//
// A probe step (a few dozen scan batches of the left table) runs count → scan as above; the batch cuts are then found on
// the device from the scan alone (hj_launch_batch_cuts), the host lays the step's batches out so that each starts on a
// multiple of 64 rows (one validity word never spans two batches, every batch's values are 256-byte aligned), the write
// pass emits DEVICE row indices into that layout and one gather launch per side and group of columns (the projection
// kernel of scan_stream, `ProjPlan`; PAD form for the NULL-padded side of a LEFT join) fills the output columns.  Outputs
// are double-buffered: step i crosses PCIe on the copy stream while step i + 1 is counted and the host hands out step
// i − 1's batches as views into the pinned buffers.
namespace {
constexpr uint32_t kGatherGroup = 5; // columns per gather launch: 5 × (values + validity mask + high halves) ≤ kMaxCols slots
constexpr uint64_t kRefWindow = 65536;

// The projected columns of one side
struct SideOut {
  const Table *t = nullptr;
  bool pad = false;
  std::vector<uint32_t> fields;
  std::vector<LoweredPlan> plans; // one per group of ≤ kGatherGroup columns
  std::vector<JitKernel> kernels;
  std::vector<int32_t> dtypes;    // per column
  std::vector<uint8_t> nullable;
  std::vector<std::vector<const char *>> dicts;
  std::vector<int32_t> precision, scale;
  bool all_nullable = true;       // every projected column has NULL cells: the scan may drop rows (DropNulls)

  int prepare(const Table *table, const llkv_join_column *cols, uint32_t n, bool pad_rows) {
    t = table;
    pad = pad_rows;
    auto resolve = [&](uint32_t fid) -> const ColumnInfo * {
      auto it = t->cols.find(fid);
      return it == t->cols.end() ? nullptr : &it->second.info;
    };
    for (uint32_t i = 0; i < n; ++i) {
      const ColumnInfo *ci = resolve(cols[i].field_id);
      if (!ci) return set_error(LLKV_NOT_FOUND, "join output field " + std::to_string(cols[i].field_id) + " not found");
      fields.push_back(cols[i].field_id);
      all_nullable = all_nullable && ci->nullable;
    }
    for (uint32_t g0 = 0; g0 < n; g0 += kGatherGroup) {
      const uint32_t gn = std::min(kGatherGroup, n - g0);
      llkv_projection pr[kGatherGroup];
      std::memset(pr, 0, sizeof pr);
      for (uint32_t i = 0; i < gn; ++i) pr[i].field_id = fields[g0 + i];
      LoweredPlan lp;
      std::string err;
      int rc = lower_projection(resolve, pr, gn, &lp, &err, pad);
      if (rc) return set_error(rc, err);
      JitKernel k;
      if ((rc = jit_compile(JitKind::Project, lp.type_string, &k, &err))) return set_error(rc, err);
      for (uint32_t i = 0; i < gn; ++i) {
        const ColumnInfo &ci = t->cols.at(fields[g0 + i]).info;
        dtypes.push_back(lp.out_dtypes[i]);
        nullable.push_back(lp.out_nullable[i]);
        dicts.emplace_back();
        if (lp.out_dtypes[i] == LLKV_DT_UTF8) for (auto &str : ci.dictionary) dicts.back().push_back(str.c_str());
        precision.push_back(lp.out_dtypes[i] == LLKV_DT_DECIMAL128 ? ci.precision : 0);
        scale.push_back(lp.out_dtypes[i] == LLKV_DT_DECIMAL128 ? ci.scale : 0);
      }
      plans.push_back(std::move(lp));
      kernels.push_back(k);
    }
    return LLKV_OK;
  }
  uint32_t n() const { return (uint32_t)fields.size(); }
};

// live[dev row] of a side whose user columns all have NULL cells (else no row can be dropped: *live stays empty)
int side_live_mask(const SideOut &so, DBuf *live, uint64_t *dead, hipStream_t s, bool *synthetic) {
  *dead = 0;
  *synthetic = false;
  if (!so.all_nullable || so.n() == 0 || so.t->local_rows == 0) return LLKV_OK;
  if (so.n() > 32) return set_error(LLKV_UNSUPPORTED, "more than 32 output columns that all have NULL cells");
  const TileSet *ts = nullptr;
  int rc = get_tileset(*so.t, kJoinTileRows, &ts);
  if (rc) return rc;
  LiveMaskCols lc;
  std::memset(&lc, 0, sizeof lc);
  for (uint32_t f : so.fields) lc.valid[lc.n++] = so.t->cols.at(f).d_valid;
  DBuf d_dead;
  if ((rc = live->ensure(so.t->dev_rows + 8)) || (rc = d_dead.ensure(8))) return rc;
  HIP_TRY(hipMemsetAsync(d_dead.p, 0, 8, s));
  HIP_TRY(hj_launch_live_mask(lc, ts->d_tiles, ts->n_tiles, (uint8_t *)live->p, (unsigned long long *)d_dead.p, s));
  Readback rb;
  if ((rc = rb.add(dead, d_dead.p, 8, s)) || (rc = rb.wait())) return rc;
  // a table whose every row is dropped comes out of the reference's scan as ONE synthetic batch of total_rows NULL rows
  // (llkv-scan/src/execute.rs:355-372, llkv-compute/src/projection.rs:36-66): the rows are back — every cell NULL, as it was —
  // in one batch instead of one per 65 536-row window
  if (*dead == so.t->local_rows) {
    if (so.t->world != 1) return set_error(LLKV_UNSUPPORTED, "a sharded join side whose every row is NULL in all of its user columns");
    *dead = 0;
    *synthetic = true;
  }
  return LLKV_OK;
}

// Output buffers of one device step and what the host needs to hand its batches out
struct OutStep {
  std::vector<DBuf> d, d_valid;
  std::vector<HBuf> h, h_valid;
  hipEvent_t written = nullptr, copied = nullptr;
  bool live = false;
  uint64_t carry_in = 0;              // rows of the running batch that earlier steps hold (the device saw the same number)
  std::vector<uint64_t> pos, len;     // the step's batches: first row in the layout, rows
  std::vector<uint8_t> closed;        // the batch ends in this step
  uint32_t error = 0;
  ~OutStep() {
    if (written) { (void)hipEventSynchronize(written); (void)hipEventDestroy(written); }
    if (copied) { (void)hipEventSynchronize(copied); (void)hipEventDestroy(copied); }
  }
};

struct JoinEmitter {
  SideOut L, R;
  bool right_out = false;            // the batches carry right columns (not SEMI / ANTI)
  std::vector<std::string> names;
  std::vector<const char *> name_ptrs;
  std::vector<uint32_t> widths;      // per output column
  std::vector<uint8_t> nullable;
  llkv_on_join_record_batch on_batch = nullptr;
  void *user = nullptr;
  OutStep steps[2];
  hipStream_t copy_stream = nullptr;
  DBuf d_err;
  // rows of a batch that began in an earlier step
  uint64_t pend_rows = 0;
  std::vector<std::vector<uint8_t>> pend_vals;
  std::vector<std::vector<uint64_t>> pend_valid;

  // (the steps' pinned buffers go back to their pool after this: no copy of an abandoned step may still be writing them)
  ~JoinEmitter() { if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); } }
  uint32_t n_out() const { return (uint32_t)widths.size(); }

  int init() {
    for (const SideOut *so : {&L, &R}) {
      if (so == &R && !right_out) break;
      for (uint32_t i = 0; i < so->n(); ++i) { widths.push_back(dtype_out_width(so->dtypes[i])); nullable.push_back(so->nullable[i]); }
    }
    pend_vals.resize(n_out());
    pend_valid.resize(n_out());
    for (OutStep &st : steps) {
      st.d.resize(n_out()); st.d_valid.resize(n_out()); st.h.resize(n_out()); st.h_valid.resize(n_out());
      HIP_TRY(hipEventCreateWithFlags(&st.written, hipEventDisableTiming));
      HIP_TRY(hipEventCreateWithFlags(&st.copied, hipEventDisableTiming));
    }
    HIP_TRY(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    return d_err.ensure(8);
  }

  // Gathers the `rows` layout positions of a step (d_lrows / d_rrows: device row per position) into its output buffers
  // and queues their copies.  `right_all_null`: the build side has no batch to gather from (all NULL)
  int gather(OutStep &st, const uint64_t *d_lrows, const uint64_t *d_rrows, uint64_t rows, bool right_all_null, hipStream_t s) {
    int rc;
    if (rows >= (1ull << 32)) return set_error(LLKV_UNSUPPORTED, "join step of 2^32 output rows");
    for (uint32_t o = 0; o < n_out(); ++o) {
      if ((rc = st.d[o].ensure(rows * widths[o])) || (rc = st.h[o].ensure(rows * widths[o]))) return rc;
      if (nullable[o] && ((rc = st.d_valid[o].ensure(rows / 8 + 8)) || (rc = st.h_valid[o].ensure(rows / 8 + 8)))) return rc;
    }
    HIP_TRY(hipMemsetAsync(d_err.p, 0, 8, s));
    uint32_t o0 = 0;
    for (const SideOut *so : {&L, &R}) {
      if (so == &R && !right_out) break;
      const uint64_t *rows_of = so == &L ? d_lrows : d_rrows;
      uint32_t c0 = 0;
      for (size_t g = 0; g < so->plans.size(); ++g) {
        const LoweredPlan &lp = so->plans[g];
        const uint32_t gn = (uint32_t)lp.out_dtypes.size();
        if (so == &R && right_all_null) {
          for (uint32_t i = 0; i < gn; ++i) {
            HIP_TRY(hipMemsetAsync(st.d[o0 + c0 + i].p, 0, rows * widths[o0 + c0 + i], s));
            HIP_TRY(hipMemsetAsync(st.d_valid[o0 + c0 + i].p, 0, rows / 8 + 8, s));
          }
        } else {
          ProjParams q;
          std::memset(&q, 0, sizeof q);
          for (size_t sl = 0; sl < lp.slot_fields.size(); ++sl) q.col[sl] = slot_buffer(so->t->cols, lp, sl);
          q.dev_rows = rows_of;
          q.n = (uint32_t)rows;
          for (uint32_t i = 0; i < gn; ++i) { q.out[i] = st.d[o0 + c0 + i].p; q.out_valid[i] = (uint64_t *)st.d_valid[o0 + c0 + i].p; }
          q.error_flag = (uint32_t *)d_err.p;
          if ((rc = jit_launch_raw(so->kernels[g].fn, (uint32_t)((rows + kBlock - 1) / kBlock), &q, sizeof q, s))) return rc;
        }
        c0 += gn;
      }
      o0 += so->n();
    }
    HIP_TRY(hipEventRecord(st.written, s));
    HIP_TRY(hipStreamWaitEvent(copy_stream, st.written, 0));
    for (uint32_t o = 0; o < n_out(); ++o) {
      HIP_TRY(hipMemcpyAsync(st.h[o].p, st.d[o].p, rows * widths[o], hipMemcpyDeviceToHost, copy_stream));
      if (nullable[o]) HIP_TRY(hipMemcpyAsync(st.h_valid[o].p, st.d_valid[o].p, (rows + 63) / 64 * 8, hipMemcpyDeviceToHost, copy_stream));
    }
    HIP_TRY(hipEventRecord(st.copied, copy_stream));
    st.live = true;
    return LLKV_OK;
  }

  void deliver(const void *const *vals, const uint8_t *const *valid, uint64_t rows) {
    llkv_column_view cols[64];
    std::vector<llkv_column_view> many;
    llkv_column_view *cv = cols;
    if (n_out() > 64) { many.resize(n_out()); cv = many.data(); }
    uint32_t o = 0;
    for (const SideOut *so : {&L, &R}) {
      if (so == &R && !right_out) break;
      for (uint32_t i = 0; i < so->n(); ++i, ++o) {
        cv[o].dtype = so->dtypes[i];
        cv[o].values = vals[o];
        cv[o].validity = valid[o];
        cv[o].dictionary = so->dicts[i].empty() ? nullptr : so->dicts[i].data();
        cv[o].precision = so->precision[i];
        cv[o].scale = so->scale[i];
      }
    }
    llkv_batch_view b;
    b.num_rows = rows;
    b.num_columns = n_out();
    b.columns = cv;
    b.row_ids = nullptr; // the row-id column is not a user column (build_output_schema)
    on_batch(&b, name_ptrs.data(), user);
  }

  // rows [pos, pos + len) of a step's buffers join the running batch; pos ≡ pend_rows (mod 64), so validity words line up
  void append(const OutStep &st, uint64_t pos, uint64_t len) {
    const uint64_t sh = pend_rows % 64;
    for (uint32_t o = 0; o < n_out(); ++o) {
      const uint8_t *src = (const uint8_t *)st.h[o].p + pos * widths[o];
      pend_vals[o].insert(pend_vals[o].end(), src, src + len * widths[o]);
      if (!nullable[o]) continue;
      const uint64_t *w = (const uint64_t *)st.h_valid[o].p + pos / 64;
      const uint64_t n_words = (sh + len + 63) / 64;
      uint64_t k = 0;
      if (sh) { // the first word continues the last word of the running batch
        const uint64_t low = (1ull << sh) - 1;
        pend_valid[o].back() = (pend_valid[o].back() & low) | (w[0] & ~low);
        k = 1;
      }
      pend_valid[o].insert(pend_valid[o].end(), w + k, w + n_words);
    }
    pend_rows += len;
  }
  void flush_pending() {
    if (!pend_rows) return;
    std::vector<const void *> vals(n_out());
    std::vector<const uint8_t *> valid(n_out());
    for (uint32_t o = 0; o < n_out(); ++o) {
      vals[o] = pend_vals[o].data();
      valid[o] = nullable[o] ? (const uint8_t *)pend_valid[o].data() : nullptr;
    }
    deliver(vals.data(), valid.data(), pend_rows);
    for (uint32_t o = 0; o < n_out(); ++o) { pend_vals[o].clear(); pend_valid[o].clear(); }
    pend_rows = 0;
  }

  // hands out the batches of a finished step
  int emit(OutStep &st) {
    if (!st.live) return LLKV_OK;
    st.live = false;
    HIP_TRY(hipEventSynchronize(st.copied));
    if (st.carry_in != pend_rows) return set_error(LLKV_INTERNAL, "join batch carry out of step");
    std::vector<const void *> vals(n_out());
    std::vector<const uint8_t *> valid(n_out());
    for (size_t b = 0; b < st.pos.size(); ++b) {
      const uint64_t pos = st.pos[b], len = st.len[b];
      if (!st.closed[b]) { if (len) append(st, pos, len); continue; }
      if (pend_rows) { append(st, pos, len); flush_pending(); continue; }
      if (!len) continue;
      for (uint32_t o = 0; o < n_out(); ++o) {
        vals[o] = (const uint8_t *)st.h[o].p + pos * widths[o];
        valid[o] = nullable[o] ? (const uint8_t *)st.h_valid[o].p + pos / 8 : nullptr;
      }
      deliver(vals.data(), valid.data(), len);
    }
    return LLKV_OK;
  }
};

// build_output_schema hash_join.rs:877-943 (names only; the executor keeps them as given, lib.rs:12237-12244)
void join_output_names(const llkv_join_output *out, bool left_only, bool executor, std::vector<std::string> *names) {
  names->clear();
  for (uint32_t i = 0; i < out->n_left; ++i) names->push_back(out->left_columns[i].name ? out->left_columns[i].name : "");
  if (left_only) return;
  for (uint32_t i = 0; i < out->n_right; ++i) {
    std::string nm = out->right_columns[i].name ? out->right_columns[i].name : "";
    if (!executor && std::find(names->begin(), names->end(), nm) != names->end()) nm += "_1";
    names->push_back(nm);
  }
}
} // namespace

int join_output_names_c(const llkv_join_output *output, int32_t join_type, int32_t key_rules, char **names, uint32_t *n_names) {
  if (!output || !names || !n_names || (output->n_left && !output->left_columns) || (output->n_right && !output->right_columns))
    return set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  std::vector<std::string> v;
  join_output_names(output, join_type == LLKV_JOIN_SEMI || join_type == LLKV_JOIN_ANTI, key_rules == LLKV_JOIN_KEYS_EXECUTOR, &v);
  for (size_t i = 0; i < v.size(); ++i) {
    names[i] = (char *)std::malloc(v[i].size() + 1);
    if (!names[i]) return set_error(LLKV_INTERNAL, "out of memory");
    std::memcpy(names[i], v[i].c_str(), v[i].size() + 1);
  }
  *n_names = (uint32_t)v.size();
  return LLKV_OK;
}

int run_join_batches(const Table *left, const Table *right, const llkv_join_key *keys, uint32_t n_keys,
                     const llkv_join_options *options, const llkv_join_output *output, llkv_on_join_record_batch on_batch, void *user) {
  JoinPlan jp;
  int rc = join_options(options, n_keys, &jp);
  if (rc) return rc;
  if ((rc = ensure_device())) return rc;
  if (!left || !right || !on_batch || !output) return set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  if ((output->n_left && !output->left_columns) || (output->n_right && !output->right_columns)) return set_error(LLKV_INVALID_ARGUMENT, "NULL column list");
  const int jt = jp.jt;
  hipStream_t s = g_ctx.stream;

  JoinEmitter em;
  em.on_batch = on_batch;
  em.user = user;
  em.right_out = !jp.left_only;
  // the NULL-padded side of a LEFT join: every right column carries a validity bitmap
  if ((rc = em.L.prepare(left, output->left_columns, output->n_left, false)) ||
      (rc = em.R.prepare(right, output->right_columns, output->n_right, jt == LLKV_JOIN_LEFT)))
    return rc;
  join_output_names(output, jp.left_only, jp.executor, &em.names);
  for (auto &nm : em.names) em.name_ptrs.push_back(nm.c_str());
  if ((rc = em.init())) return rc;

  // ---- no key pair: the Cartesian product (cross_product_stream hash_join.rs:1500-1599) ---------------------------
  if (n_keys == 0) {
    if (left->world != 1 || right->world != 1) return set_error(LLKV_UNSUPPORTED, "cross product over sharded tables");
    // the two scans: all user columns, DropNulls; batches = the surviving rows of every 65 536-row-id window
    Selection rsel, lsel;
    // (a side whose every row is NULL in all of its user columns: ONE synthetic batch of all its rows, llkv-scan/src/execute.rs:355-372)
    bool l_one = false, r_one = false;
    if (em.R.n() && (rc = run_selection(right, nullptr, 0, nullptr, 0, &rsel, em.R.fields.data(), em.R.n()))) return rc;
    if (em.R.n() && right->local_rows && rsel.n == 0) {
      if ((rc = run_selection(right, nullptr, 0, nullptr, 0, &rsel))) return rc;
      r_one = true;
    }
    const bool right_empty = rsel.n == 0;
    if (right_empty && jt == LLKV_JOIN_INNER) return LLKV_OK;
    if (em.L.n() == 0) return LLKV_OK;
    if ((rc = run_selection(left, nullptr, 0, nullptr, 0, &lsel, em.L.fields.data(), em.L.n()))) return rc;
    if (left->local_rows && lsel.n == 0) {
      if ((rc = run_selection(left, nullptr, 0, nullptr, 0, &lsel))) return rc;
      l_one = true;
    }
    auto windows = [&](const Selection &sel, std::vector<uint64_t> *start) -> int { // index of every window's first surviving row (+ end)
      std::vector<uint64_t> ids(sel.n);
      if (sel.n) { int r = fetch_to_host(ids.data(), sel.d_ids, sel.n * 8); if (r) return r; }
      start->clear();
      for (uint64_t i = 0; i < sel.n; ++i)
        if (i == 0 || ids[i] / kRefWindow != ids[i - 1] / kRefWindow) start->push_back(i);
      start->push_back(sel.n);
      return LLKV_OK;
    };
    std::vector<uint64_t> lw, rw;
    if ((rc = windows(lsel, &lw)) || (rc = windows(rsel, &rw))) return rc;
    if (l_one) lw = {0, lsel.n};
    if (r_one) rw = {0, rsel.n};
    DBuf d_l, d_r;
    int cur = 0;
    for (size_t li = 0; li + 1 < lw.size(); ++li) {
      const uint64_t ln = lw[li + 1] - lw[li];
      if (right_empty) { // LEFT: synthesize_left_join_nulls — the left batch and NULL arrays
        if (jt != LLKV_JOIN_LEFT) continue; // SEMI / ANTI: no right batch to pair with
        OutStep &st = em.steps[cur];
        if ((rc = em.emit(st))) return rc;
        st.carry_in = 0; st.pos = {0}; st.len = {ln}; st.closed = {1};
        if ((rc = em.gather(st, lsel.d_dev + lw[li], nullptr, ln, true, s)) || (rc = em.emit(em.steps[cur ^ 1]))) return rc;
        cur ^= 1;
        continue;
      }
      for (size_t ri = 0; ri + 1 < rw.size(); ++ri) {
        // SEMI / ANTI: the schema holds the left columns only and cross_join_pair refuses the pair (cartesian.rs:36-44)
        if (jp.left_only) return set_error(LLKV_INTERNAL, "cross join schema mismatch: semi / anti joins deliver left columns only");
        const uint64_t rn = rw[ri + 1] - rw[ri], np = ln * rn;
        if (np > (1ull << 26)) return set_error(LLKV_UNSUPPORTED, "cross product batch of more than 2^26 rows");
        OutStep &st = em.steps[cur];
        if ((rc = em.emit(st))) return rc; // its buffers are about to be reused
        if ((rc = d_l.ensure(np * 8)) || (rc = d_r.ensure(np * 8))) return rc;
        HIP_TRY(hj_launch_cross_rows(lsel.d_dev + lw[li], ln, rsel.d_dev + rw[ri], rn, (uint64_t *)d_l.p, (uint64_t *)d_r.p, s));
        st.carry_in = 0; st.pos = {0}; st.len = {np}; st.closed = {1};
        if ((rc = em.gather(st, (const uint64_t *)d_l.p, (const uint64_t *)d_r.p, np, false, s)) || (rc = em.emit(em.steps[cur ^ 1]))) return rc;
        cur ^= 1;
      }
    }
    if ((rc = em.emit(em.steps[cur])) || (rc = em.emit(em.steps[cur ^ 1]))) return rc;
    HIP_TRY(hipStreamSynchronize(s));
    return LLKV_OK;
  }

  // ---- hash join --------------------------------------------------------------------------------------------------
  if ((rc = join_key_setup(left, right, keys, n_keys, &jp))) return rc;
  // no left columns: the reference scans nothing (hash_join.rs:226)
  if (em.L.n() == 0) return LLKV_OK;
  // rows the two scans drop (NULL in every user column); the executor's join reads its tables with their NULL rows
  DBuf l_live, r_live;
  uint64_t l_dead = 0, r_dead = 0;
  bool l_synthetic = false, r_synthetic = false; // (the build side's batches do not show in a hash join's output)
  if (!jp.executor && ((rc = side_live_mask(em.L, &l_live, &l_dead, s, &l_synthetic)) || (rc = side_live_mask(em.R, &r_live, &r_dead, s, &r_synthetic)))) return rc;
  if (l_dead) jp.lk.live = (const uint8_t *)l_live.p;
  if (r_dead) jp.rk.live = (const uint8_t *)r_live.p;
  if (l_dead && !jp.fast && !jp.executor)
    return set_error(LLKV_UNSUPPORTED, "generic join path over a probe side with rows that are NULL in every user column (the slices count surviving rows)");
  if ((rc = join_build(left, right, &jp, s))) return rc;
  const TileSet *tl = jp.tl;
  // a build side without a batch (no right columns, no rows): INNER / SEMI find nothing, ANTI everything; a LEFT join's
  // gather_optional_indices_from_batches returns no arrays and RecordBatch::try_new fails on the column count — the
  // fast path logs and drops that error per probe batch (hash_join.rs:1058-1060), the generic path returns it (:313-317)
  const bool build_empty = em.R.n() == 0 || jp.n_build == 0;
  if (em.R.n() == 0 && jp.n_build) { // (no right columns: the reference builds nothing, :211-215 — keys of the right table match nothing)
    HIP_TRY(hipMemsetAsync(jp.owner.p, 0xFF, jp.cap * 8, s));
    HIP_TRY(hipMemsetAsync(jp.seg_count.p, 0, (jp.cap + 1) * 4, s));
  }
  if (build_empty && jt == LLKV_JOIN_LEFT && !jp.executor) {
    if (jp.fast || left->local_rows == l_dead) return LLKV_OK;
    return set_error(LLKV_INTERNAL, "Invalid argument error: number of columns(" + std::to_string(em.L.n()) + ") must match number of fields(" +
                                        std::to_string(em.L.n() + em.R.n()) + ") in schema");
  }

  const uint32_t win_pos = std::min(kWindowTiles, std::max(1u, tl->n_tiles)) * kJoinTileRows;
  DBuf counts, mslot, offsets, scan_tmp, seg_pos_d, seg_cuts, seg_base, cuts_d, carry_d, shift_d, d_lrows, d_rrows;
  HBuf seg_pos_h, cuts_h, shift_h;
  if ((rc = counts.ensure((size_t)(win_pos + 1) * 8)) || (rc = mslot.ensure((size_t)win_pos * 4)) || (rc = offsets.ensure((size_t)(win_pos + 1) * 8)) || (rc = carry_d.ensure(8))) return rc;
  std::vector<TileDesc> ltiles;
  {
    uint32_t otb[kOctantsHost + 1];
    build_tiles_host(*left, kJoinTileRows, ltiles, otb);
  }
  const uint64_t left_end = left->local_logical_start + left->local_rows;
  const uint64_t batch_size = jp.batch_size;
  constexpr uint64_t kMaxStepRows = 16ull << 20; // output rows of one step (× the row width in HBM and in pinned memory, twice)
  uint32_t step_tiles = kWindowTiles;
  uint64_t carry = 0; // rows of the running batch in steps already launched
  int cur = 0;
  for (uint32_t t0 = 0, nt = 0; t0 < tl->n_tiles; t0 += nt) {
    nt = std::min(step_tiles, tl->n_tiles - t0);
    const uint32_t npos = nt * kJoinTileRows;
    ProbeParams p;
    std::memset(&p, 0, sizeof p);
    p.lkey = jp.lk; p.rkey = jp.rk;
    p.tiles = tl->d_tiles + t0; p.n_tiles = nt; p.tile_rows = kJoinTileRows;
    p.slot_owner = (const unsigned long long *)jp.owner.p; p.cap_mask = jp.cap - 1;
    p.seg_start = (const uint32_t *)jp.seg_start.p; p.seg_count = (const uint32_t *)jp.seg_count.p;
    p.sorted_idx = (const uint32_t *)jp.idx_sorted.p; p.build_logical = (const uint64_t *)jp.log_of.p;
    p.build_dev = (const uint64_t *)jp.dev_of.p;
    p.join_type = jt;
    p.counts = (uint64_t *)counts.p; p.match_slot = (uint32_t *)mslot.p;
    HIP_TRY(hj_launch_probe_count(p, s));
    HIP_TRY(hipMemsetAsync((uint64_t *)counts.p + npos, 0, 8, s));
    {
      size_t tb = 0;
      HIP_TRY(hj_exclusive_scan_u64(nullptr, &tb, (const uint64_t *)counts.p, (uint64_t *)offsets.p, (uint64_t)npos + 1, s));
      if ((rc = scan_tmp.ensure(tb))) return rc;
      HIP_TRY(hj_exclusive_scan_u64(scan_tmp.p, &tb, (const uint64_t *)counts.p, (uint64_t *)offsets.p, (uint64_t)npos + 1, s));
    }
    // the step's segments in position space: a forced cut at the end of every reference scan batch (65 536 row ids), of
    // every slice of batch_size rows of it (generic path), and of the table
    uint64_t wrows = 0;
    for (uint32_t t = 0; t < nt; ++t) wrows += ltiles[t0 + t].rows;
    const uint64_t L0 = ltiles[t0].logical_row, L1 = L0 + wrows; // the rows of a rank are contiguous
    std::vector<uint32_t> seg_pos;
    bool last_open = false;
    if (jp.executor) { // one batch per device step
      seg_pos = {0u, npos};
    } else {
      uint32_t t = t0;
      auto pos_of = [&](uint64_t row) { // row in [L0, L1]
        while (t + 1 < t0 + nt && row >= ltiles[t + 1].logical_row) ++t;
        return (uint32_t)((t - t0) * kJoinTileRows + std::min<uint64_t>(row - ltiles[t].logical_row, ltiles[t].rows));
      };
      seg_pos.push_back(0);
      for (uint64_t row = L0; row < L1;) {
        const uint64_t in_win = l_synthetic ? row : row % kRefWindow; // (the synthetic batch of an all-NULL side: one window)
        uint64_t b = l_synthetic ? left_end : row - in_win + kRefWindow;
        if (!jp.fast) b = std::min(b, row - in_win + (in_win / batch_size + 1) * batch_size);
        b = std::min(b, left_end);
        if (b > L1) { last_open = true; b = L1; }
        seg_pos.push_back(b == L1 ? npos : pos_of(b));
        row = b;
      }
      if (seg_pos.size() == 1) seg_pos.push_back(npos); // (a step without rows)
    }
    const uint32_t n_seg = (uint32_t)seg_pos.size() - 1;
    const uint64_t cut_cap = (uint64_t)npos + n_seg + 1;
    if ((rc = seg_pos_h.ensure(seg_pos.size() * 4)) || (rc = seg_pos_d.ensure(seg_pos.size() * 4)) || (rc = seg_cuts.ensure((size_t)(n_seg + 1) * 8)) ||
        (rc = seg_base.ensure((size_t)(n_seg + 1) * 8)) || (rc = cuts_d.ensure(cut_cap * 8)))
      return rc;
    std::memcpy(seg_pos_h.p, seg_pos.data(), seg_pos.size() * 4);
    HIP_TRY(hipMemcpyAsync(seg_pos_d.p, seg_pos_h.p, seg_pos.size() * 4, hipMemcpyHostToDevice, s));
    CutParams cp;
    std::memset(&cp, 0, sizeof cp);
    cp.offsets = (const uint64_t *)offsets.p; cp.seg_pos = (const uint32_t *)seg_pos_d.p; cp.n_seg = n_seg; cp.last_open = last_open;
    cp.batch_size = batch_size; cp.carry_in = carry; cp.carry_out = (uint64_t *)carry_d.p;
    cp.seg_cuts = (uint64_t *)seg_cuts.p;
    HIP_TRY(hj_launch_batch_cuts(cp, s));
    {
      size_t tb = 0;
      HIP_TRY(hj_exclusive_scan_u64(nullptr, &tb, (const uint64_t *)seg_cuts.p, (uint64_t *)seg_base.p, (uint64_t)n_seg + 1, s));
      if ((rc = scan_tmp.ensure(tb))) return rc;
      HIP_TRY(hj_exclusive_scan_u64(scan_tmp.p, &tb, (const uint64_t *)seg_cuts.p, (uint64_t *)seg_base.p, (uint64_t)n_seg + 1, s));
    }
    cp.seg_cut_base = (const uint64_t *)seg_base.p; cp.cuts = (uint64_t *)cuts_d.p;
    HIP_TRY(hj_launch_batch_cuts(cp, s));
    uint64_t total = 0, n_cuts = 0, carry_out = 0;
    constexpr uint64_t kEagerCuts = 4096; // the first cuts travel with the counts
    if ((rc = cuts_h.ensure(std::max<uint64_t>(cut_cap, kEagerCuts) * 8))) return rc;
    {
      Readback rb;
      if ((rc = rb.add(&total, (uint64_t *)offsets.p + npos, 8, s)) || (rc = rb.add(&n_cuts, (uint64_t *)seg_base.p + n_seg, 8, s)) ||
          (rc = rb.add(&carry_out, carry_d.p, 8, s)) || (rc = rb.add(cuts_h.p, cuts_d.p, std::min<uint64_t>(cut_cap, kEagerCuts) * 8, s)) || (rc = rb.wait()))
        return rc;
    }
    if (n_cuts > kEagerCuts) {
      HIP_TRY(hipMemcpyAsync(cuts_h.p, cuts_d.p, n_cuts * 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
    }
    // the layout: batch b of the step (b = n_cuts: the rows after the last cut, which the next step continues) at a
    // multiple of 64 — the first one at carry % 64, where the running batch's validity words go on
    const uint64_t *cuts = (const uint64_t *)cuts_h.p;
    std::vector<uint64_t> pos(n_cuts + 1), len(n_cuts + 1);
    uint64_t at = carry % 64, start = 0;
    for (uint64_t b = 0; b <= n_cuts; ++b) {
      const uint64_t end = b < n_cuts ? cuts[b] : total;
      pos[b] = at; len[b] = end - start;
      at = (at + len[b] + 63) / 64 * 64;
      start = end;
    }
    const uint64_t rows = at;
    if (rows > kMaxStepRows && nt > 1) { // a many-to-many key: count again over fewer tiles
      step_tiles = std::max(1u, nt / 2);
      nt = 0;
      continue;
    }
    OutStep &st = em.steps[cur];
    if ((rc = em.emit(st))) return rc; // its buffers are about to be reused (normally already emitted below)
    st.carry_in = carry;
    st.pos = pos; st.len = len;
    st.closed.assign(n_cuts + 1, 1);
    st.closed[n_cuts] = 0;
    carry = carry_out;
    if (total) {
      if ((rc = shift_h.ensure((n_cuts + 1) * 8)) || (rc = shift_d.ensure((n_cuts + 1) * 8)) || (rc = d_lrows.ensure(rows * 8)) || (rc = d_rrows.ensure(rows * 8))) return rc;
      int64_t *sh = (int64_t *)shift_h.p;
      start = 0;
      for (uint64_t b = 0; b <= n_cuts; ++b) { sh[b] = (int64_t)pos[b] - (int64_t)start; start = b < n_cuts ? cuts[b] : total; }
      HIP_TRY(hipMemcpyAsync(shift_d.p, shift_h.p, (n_cuts + 1) * 8, hipMemcpyHostToDevice, s));
      // layout positions no pair lands on gather row 0 (left) / nothing (right)
      HIP_TRY(hipMemsetAsync(d_lrows.p, 0, rows * 8, s));
      if (em.right_out) HIP_TRY(hipMemsetAsync(d_rrows.p, jt == LLKV_JOIN_LEFT ? 0xFF : 0, rows * 8, s));
      p.offsets = (const uint64_t *)offsets.p;
      p.cuts = (const uint64_t *)cuts_d.p; p.n_cuts = (uint32_t)n_cuts; p.batch_shift = (const int64_t *)shift_d.p;
      p.out_left = (uint64_t *)d_lrows.p; p.out_right = em.right_out ? (uint64_t *)d_rrows.p : nullptr;
      HIP_TRY(hj_launch_probe_write_rows(p, s));
      if ((rc = em.gather(st, (const uint64_t *)d_lrows.p, (const uint64_t *)d_rrows.p, rows, build_empty, s))) return rc;
    } else {
      st.live = true; // nothing to copy; batches of length 0 only
      HIP_TRY(hipEventRecord(st.copied, em.copy_stream));
    }
    if ((rc = em.emit(em.steps[cur ^ 1]))) return rc; // the previous step, while this one's columns are on their way
    cur ^= 1;
  }
  if ((rc = em.emit(em.steps[cur])) || (rc = em.emit(em.steps[cur ^ 1]))) return rc; // oldest first
  em.flush_pending(); // (rows carried over a step that ended inside a reference batch: the last batch)
  HIP_TRY(hipStreamSynchronize(s));
  return LLKV_OK;
}

} // namespace llkv
