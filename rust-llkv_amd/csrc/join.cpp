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
} // namespace

int run_join(const Table *left, const Table *right, const llkv_join_key *keys, uint32_t n_keys,
             const llkv_join_options *options, llkv_on_join_batch on_batch, void *user) {
  const bool executor = options && options->key_rules == LLKV_JOIN_KEYS_EXECUTOR;
  if (options && options->key_rules != LLKV_JOIN_KEYS_TABLE && !executor) return set_error(LLKV_INVALID_ARGUMENT, "unknown join key rules");
  // executor rules: no batch cuts (hash_join_table_batches materialises one batch)
  const uint64_t batch_size = executor ? UINT64_MAX : options ? options->batch_size : 8192;
  const int jt = options ? options->join_type : LLKV_JOIN_INNER;
  if (executor && jt != LLKV_JOIN_INNER && jt != LLKV_JOIN_LEFT)
    return set_error(LLKV_INTERNAL, "join type not supported in hash_join_table_batches; use llkv-join"); // llkv-executor/src/lib.rs:12387-12391
  if (executor && n_keys == 0) return set_error(LLKV_INVALID_ARGUMENT, "executor join rules need at least one key pair");
  // validate_join_options llkv-join/src/lib.rs:284-310, hash_join.rs:328-332
  if (batch_size == 0) return set_error(LLKV_INVALID_ARGUMENT, "join batch_size must be greater than zero");
  if (jt == LLKV_JOIN_RIGHT || jt == LLKV_JOIN_FULL) return set_error(LLKV_INVALID_ARGUMENT, "Right and Full joins are not yet implemented");
  if (jt != LLKV_JOIN_INNER && jt != LLKV_JOIN_LEFT && jt != LLKV_JOIN_SEMI && jt != LLKV_JOIN_ANTI) return set_error(LLKV_INVALID_ARGUMENT, "unknown join type");
  int rc = ensure_device();
  if (rc) return rc;
  if (!left || !right || !on_batch) return set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
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
  // "build side replicated, probe side sharded" (BASELINE.json configs[4]): every rank probes its own rows of the left
  // table against the whole right table; the binding concatenates the ranks' batches in rank order
  if (right->world != 1) return set_error(LLKV_INVALID_ARGUMENT, "the build (right) table of a join is replicated: stage it whole (world = 1) on every rank");
  if (n_keys > kMaxJoinKeys) return set_error(LLKV_UNSUPPORTED, "GPU join path takes at most " + std::to_string(kMaxJoinKeys) + " key pairs");
  if (!keys) return set_error(LLKV_INVALID_ARGUMENT, "join keys is NULL");
  JoinKeySet lk, rk;
  std::memset(&lk, 0, sizeof lk);
  std::memset(&rk, 0, sizeof rk);
  lk.n = rk.n = n_keys;
  const DeviceColumn *lc[kMaxJoinKeys], *rc_[kMaxJoinKeys];
  for (uint32_t i = 0; i < n_keys; ++i)
    if ((rc = key_part(left, keys[i].left_field, &lc[i], &lk.k[i])) || (rc = key_part(right, keys[i].right_field, &rc_[i], &rk.k[i]))) return rc;
  // one key of one fast integer type on both sides → the integer fast path; anything else → the generic
  // typed-key path (hash_join.rs:171-200), with its own NULL rule and its own batching
  const bool fast = !executor && n_keys == 1 && lc[0]->info.dtype == rc_[0]->info.dtype && fast_key_type(lc[0]->info.dtype);
  DBuf translate;
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
  const TileSet *tr = nullptr, *tl = nullptr;
  if ((rc = get_tileset(*right, kJoinTileRows, &tr)) || (rc = get_tileset(*left, kJoinTileRows, &tl))) return rc;

  // ---- build (right) ----
  const uint64_t n_build = right->local_rows;
  if (n_build >= (1ull << 31)) return set_error(LLKV_UNSUPPORTED, "build side larger than 2^31 rows");
  uint64_t cap = 1024;
  uint32_t bits = 10;
  while (cap < 2 * n_build) { cap <<= 1; ++bits; }
  DBuf owner, slot_of, dev_of, log_of, seg_start, seg_count, idx_in, slot_sorted, idx_sorted, tile_base, tmp;
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
    HIP_TRY(hj_launch_claim(rk, tr->d_tiles, tr->n_tiles, kJoinTileRows, (unsigned long long *)owner.p, cap - 1, (uint32_t *)slot_of.p,
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

  if (trace) { (void)hipStreamSynchronize(s); std::fprintf(stderr, "[llkv join] build %9.3f ms (%llu rows)\n", lap(), (unsigned long long)n_build); }
  // ---- probe (left), window by window ----
  const uint32_t win_pos = std::min(kWindowTiles, std::max(1u, tl->n_tiles)) * kJoinTileRows;
  DBuf counts, mslot, offsets, scan_tmp;
  if ((rc = counts.ensure((size_t)(win_pos + 1) * 8)) || (rc = mslot.ensure((size_t)win_pos * 4)) || (rc = offsets.ensure((size_t)(win_pos + 1) * 8))) return rc;
  const bool left_only = jt == LLKV_JOIN_SEMI || jt == LLKV_JOIN_ANTI;
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

} // namespace llkv
