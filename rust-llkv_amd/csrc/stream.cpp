// stream.cpp — StorageTable::scan_stream / filter_row_ids
// (llkv-executor/src/types/storage.rs:20-50 → execute_scan llkv-scan/src/execute.rs:47-295):
// predicate → selection vector (ballot + popcount prefix, ascending row ids) → 65 536-row
// windows (execute.rs:31) → gather + computed projections on the device → pinned host
// window → on_batch on the calling thread, in row-id order, never an empty batch.
#include "engine.hpp"
#include "join.hpp"

#include <cstring>
#include <memory>

namespace llkv {

static constexpr uint32_t kRowStreamChunk = 65536; // ROW_STREAM_CHUNK_SIZE, llkv-scan/src/execute.rs:31
static constexpr uint32_t kSelectTileRows = 8192;

using DeviceBuf = Scratch;
// Pinned host buffers are recycled: pinning memory costs far more than a selective scan (hundreds of µs per
// buffer), so freed blocks wait in a small cache for the next stream.
struct PinnedBuf {
  void *p = nullptr;
  size_t bytes = 0;
  ~PinnedBuf() { if (p) pinned_release(p, bytes); }
  int alloc(size_t n) {
    bytes = n ? n : 8;
    p = pinned_acquire(&bytes);
    return p ? LLKV_OK : set_error(LLKV_INTERNAL, "pinned host allocation of " + std::to_string(bytes) + " bytes failed");
  }
};

int run_selection(const Table *t, const llkv_filter *filters, uint32_t n_filters, const llkv_eval_op *ops,
                  uint32_t n_ops, Selection *sel, const uint32_t *drop_null_fields, uint32_t n_drop_null_fields) {
  int rc = ensure_device();
  if (rc) return rc;
  if (!t) return set_error(LLKV_INVALID_ARGUMENT, "table is NULL");
  auto resolve = [&](uint32_t fid) -> const ColumnInfo * {
    auto it = t->cols.find(fid);
    return it == t->cols.end() ? nullptr : &it->second.info;
  };
  LoweredPlan plan;
  std::string err;
  if ((rc = lower_selection(resolve, filters, n_filters, ops, n_ops, drop_null_fields, n_drop_null_fields, &plan, &err))) return set_error(rc, err);
  return run_selection_lowered(t, plan, sel);
}

// The selection kernels of an already lowered predicate (prepared statements keep the plan).
int run_selection_lowered(const Table *t, const LoweredPlan &plan, Selection *sel, const KeySetView *key_set, int single_pass_mode, const BitmapSink *sink, bool sync) {
  if (sink && single_pass_mode != 1) return set_error(LLKV_INTERNAL, "a bitmap sink belongs to the single-pass selection");
  int rc;
  std::string err;
  scratch_free(sel->d_ids); sel->d_ids = nullptr;
  scratch_free(sel->d_dev); sel->d_dev = nullptr;
  sel->n = 0;
  if (plan.always_false || t->local_rows == 0) return LLKV_OK;
  const TileSet *ts = nullptr;
  if ((rc = get_tileset(*t, kSelectTileRows, &ts))) return rc;
  JitKernel k;
  if ((rc = jit_compile(JitKind::Select, plan.type_string, &k, &err))) return set_error(rc, err);

  hipStream_t stream = g_ctx.stream;
  const uint32_t n_slots = ts->n_tiles * (kBlock / 64);
  DeviceBuf counts, offsets;
  if ((rc = counts.alloc((size_t)n_slots * 8)) || (rc = offsets.alloc((size_t)(n_slots + 1) * 8))) return rc;
  ScanParams p;
  std::memset(&p, 0, sizeof p);
  for (size_t s = 0; s < plan.slot_fields.size(); ++s) p.col[s] = slot_buffer(t->cols, plan, s);
  for (size_t i = 0; i < plan.lit_i.size(); ++i) p.lit_i[i] = plan.lit_i[i];
  for (size_t i = 0; i < plan.lit_f.size(); ++i) p.lit_f[i] = plan.lit_f[i];
  p.tiles = ts->d_tiles;
  p.n_tiles = ts->n_tiles;
  p.sub_rows = kSelectTileRows / (kBlock / 64);
  p.tile_partials = (uint64_t *)counts.p;
  if (key_set) {
    p.bm_bits = key_set->bits;
    p.bm_min = key_set->kmin;
    p.bm_span = key_set->span;
  }
  bool single_pass = single_pass_mode == 1;
  if (single_pass_mode < 0 && ts->n_sample >= 8 && !std::getenv("LLKV_HIP_SELECT_TWO_PASS")) {
    // selectivity from every 64th tile (1.6 % of the predicate columns, one more small launch)
    const uint32_t sample_slots = ts->n_sample * (kBlock / 64);
    DeviceBuf sample_counts;
    if ((rc = sample_counts.alloc((size_t)sample_slots * 8))) return rc;
    ScanParams ps = p;
    ps.tiles = ts->d_sample;
    ps.n_tiles = ts->n_sample;
    ps.tile_partials = (uint64_t *)sample_counts.p;
    if ((rc = jit_launch_raw(k.fn, ts->n_sample, &ps, sizeof ps, stream))) return rc;
    std::vector<uint64_t> hc(sample_slots);
    Readback rb;
    if (sample_slots * 8 <= Readback::kBytes) {
      if ((rc = rb.add(hc.data(), sample_counts.p, (size_t)sample_slots * 8, stream)) || (rc = rb.wait())) return rc;
      uint64_t hit = 0;
      bool err_bit = false;
      for (uint64_t c : hc) { err_bit |= c >= kPredErrorBit; hit += c; }
      // one pass: P + 48 s bytes per row; two passes: 2 P + 16 s  →  one pass when 32 s < P
      if (!err_bit && ts->sample_rows) single_pass = 32.0 * (double)hit / (double)ts->sample_rows < (double)plan.bytes_per_row;
    }
  }
  DeviceBuf stripe_ids, stripe_dev;
  if (single_pass) {
    const size_t stripe_bytes = (size_t)n_slots * p.sub_rows * 8;
    if ((rc = stripe_ids.alloc(stripe_bytes)) || (rc = stripe_dev.alloc(stripe_bytes))) return rc;
    p.aux_in = nullptr;
    p.aux_out = (uint64_t *)stripe_ids.p;
    p.aux_out2 = (uint64_t *)stripe_dev.p;
    if ((rc = jit_launch_raw(k.fn2, ts->n_tiles, &p, sizeof p, stream))) return rc;
  } else if ((rc = jit_launch_raw(k.fn, ts->n_tiles, &p, sizeof p, stream))) {
    return rc;
  }
  HIP_TRY(launch_exclusive_scan((const uint64_t *)counts.p, (uint64_t *)offsets.p, n_slots, stream));
  uint64_t total = 0;
  Readback rb;
  if ((rc = rb.add(&total, (uint64_t *)offsets.p + n_slots, 8, stream)) || (rc = rb.wait())) return rc;
  if (total >= kPredErrorBit) return set_error(LLKV_INTERNAL, "Arithmetic overflow: Overflow happened in a comparison");
  sel->n = total;
  if (total == 0) return LLKV_OK;
  sel->d_ids = (uint64_t *)scratch_alloc(total * 8);
  sel->d_dev = (uint64_t *)scratch_alloc(total * 8);
  if (!sel->d_ids || !sel->d_dev) return set_error(LLKV_INTERNAL, "device scratch allocation failed");
  if (single_pass && sink) {
    HIP_TRY(hj_launch_compact_stripes2_bits((const uint64_t *)stripe_ids.p, (const uint64_t *)stripe_dev.p, (const uint64_t *)counts.p, (const uint64_t *)offsets.p,
                                            n_slots, p.sub_rows, sel->d_ids, sel->d_dev, sink->key_values, sink->key_width, sink->key_signed, sink->kmin, sink->bits,
                                            sink->unsorted_flag, stream));
  } else if (single_pass) {
    HIP_TRY(hj_launch_compact_stripes2((const uint64_t *)stripe_ids.p, (const uint64_t *)stripe_dev.p, (const uint64_t *)counts.p, (const uint64_t *)offsets.p,
                                       n_slots, p.sub_rows, sel->d_ids, sel->d_dev, stream));
  } else {
    p.aux_in = (const uint64_t *)offsets.p;
    p.aux_out = sel->d_ids;
    p.aux_out2 = sel->d_dev;
    if ((rc = jit_launch_raw(k.fn2, ts->n_tiles, &p, sizeof p, stream))) return rc;
  }
  // (d_ids are POSITIONS — what every consumer inside the library works on: window cuts, first-appearance order, sort bits;
  // a table with its own row ids has them translated where ids are reported: selection_report_ids)
  if (sync) HIP_TRY(hipStreamSynchronize(stream)); // (the scratch blocks released on return are only handed to work on this same stream)
  return LLKV_OK;
}

// A table whose row ids are not its positions (llkv_hip_table_set_row_ids): the ids the callers REPORT — filter_row_ids, the row-id
// column of scan windows — are the table's; translated in place from the device row indices, after any sort of the selection.
static int selection_report_ids(const Table *t, Selection *sel) {
  if (!t->d_row_ids || sel->n == 0) return LLKV_OK;
  HIP_TRY(hj_launch_gather_u64_by_row(t->d_row_ids, sel->d_dev, sel->n, sel->d_ids, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  return LLKV_OK;
}

// ScanStreamOptions.order: sort the selection by one column (sort_row_ids_with_order, llkv-scan/src/ordering.rs:16-140
// → arrow sort_to_indices with {descending, nulls_first}); stable, so equal keys keep row-id order.
static int sort_selection(const Table *t, const llkv_scan_options *o, Selection *sel) {
  auto it = t->cols.find(o->order_field);
  if (it == t->cols.end()) return set_error(LLKV_NOT_FOUND, "ORDER BY field " + std::to_string(o->order_field) + " not found");
  const DeviceColumn &c = it->second;
  switch (o->order_transform) {
  case LLKV_ORDER_IDENTITY_INT64: if (c.info.dtype != LLKV_DT_INT64) return set_error(LLKV_INVALID_ARGUMENT, "ORDER BY expected INT64 column for IdentityInt64 transform"); break;
  case LLKV_ORDER_IDENTITY_INT32: if (c.info.dtype != LLKV_DT_INT32) return set_error(LLKV_INVALID_ARGUMENT, "ORDER BY expected INT32 column for IdentityInt32 transform"); break;
  case LLKV_ORDER_IDENTITY_UTF8: if (c.info.dtype != LLKV_DT_UTF8) return set_error(LLKV_INVALID_ARGUMENT, "ORDER BY expected UTF8 column for IdentityUtf8 transform"); break;
  case LLKV_ORDER_CAST_UTF8_TO_INTEGER:
    if (c.info.dtype != LLKV_DT_UTF8) return set_error(LLKV_INVALID_ARGUMENT, "ORDER BY CAST expects a UTF8 column");
    return set_error(LLKV_UNSUPPORTED, "ORDER BY CAST(utf8 AS INTEGER) is not on the GPU path");
  default: return set_error(LLKV_INVALID_ARGUMENT, "unknown ORDER BY transform");
  }
  const uint64_t n = sel->n;
  if (n < 2) return LLKV_OK;
  if (n >= (1ull << 32)) return set_error(LLKV_UNSUPPORTED, "more than 2^32 selected rows in an ordered scan");
  hipStream_t s = g_ctx.stream;
  JoinKeyColumn kc;
  std::memset(&kc, 0, sizeof kc);
  kc.values = c.d_values;
  kc.valid = c.info.nullable ? c.d_valid : nullptr;
  long long base = 0;
  Scratch rank_d;
  const uint8_t *code_rank = nullptr;
  if (c.info.dtype == LLKV_DT_INT64) { kc.width = 8; kc.is_signed = 1; base = INT64_MIN; }
  else if (c.info.dtype == LLKV_DT_INT32) { kc.width = 4; kc.is_signed = 1; base = INT32_MIN; }
  else { // dictionary codes sort as their strings do (str::cmp)
    kc.width = 1;
    std::vector<uint32_t> idx(c.info.dictionary.size());
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = (uint32_t)i;
    std::sort(idx.begin(), idx.end(), [&](uint32_t a, uint32_t b) { return c.info.dictionary[a] < c.info.dictionary[b]; });
    uint8_t rank[256] = {0};
    for (size_t r = 0; r < idx.size(); ++r) rank[idx[r]] = (uint8_t)r;
    int rc = rank_d.alloc(256);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(rank_d.p, rank, 256, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    code_rank = rank_d.as<uint8_t>();
  }
  Scratch perm_a, perm_b, keys_a, keys_b, tmp;
  int rc;
  if ((rc = perm_a.alloc(n * 4)) || (rc = perm_b.alloc(n * 4)) || (rc = keys_a.alloc(n * 8)) || (rc = keys_b.alloc(n * 8))) return rc;
  HIP_TRY(hj_launch_iota(perm_a.as<uint32_t>(), (uint32_t)n, s));
  uint32_t *perm = perm_a.as<uint32_t>(), *other = perm_b.as<uint32_t>();
  HIP_TRY(hj_launch_gather_sort_keys(kc, base, code_rank, sel->d_dev, perm, n, keys_a.as<uint64_t>(), s));
  if (o->order_descending) HIP_TRY(hj_launch_xor_u64(keys_a.as<uint64_t>(), n, ~0ull, s)); // descending = ascending on the complement, still stable
  size_t tb = 0;
  HIP_TRY(hj_sort_u64_u32(nullptr, &tb, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), perm, other, n, s));
  if ((rc = tmp.alloc(tb ? tb : 8))) return rc;
  HIP_TRY(hj_sort_u64_u32(tmp.p, &tb, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), perm, other, n, s));
  std::swap(perm, other);
  HIP_TRY(hipStreamSynchronize(s));
  if (kc.valid) { // NULL cells: one more stable 1-bit pass puts them first or last
    Scratch va, vb;
    if ((rc = va.alloc(n * 4)) || (rc = vb.alloc(n * 4))) return rc;
    HIP_TRY(hj_launch_gather_valid(kc, sel->d_dev, perm, n, va.as<uint32_t>(), s));
    if (!o->order_nulls_first) HIP_TRY(hj_launch_xor_u32(va.as<uint32_t>(), n, 1u, s));
    size_t vb_bytes = 0;
    HIP_TRY(hj_sort_by_slot(nullptr, &vb_bytes, va.as<uint32_t>(), vb.as<uint32_t>(), perm, other, (uint32_t)n, 1, s));
    if ((rc = tmp.alloc(vb_bytes ? vb_bytes : 8))) return rc;
    HIP_TRY(hj_sort_by_slot(tmp.p, &vb_bytes, va.as<uint32_t>(), vb.as<uint32_t>(), perm, other, (uint32_t)n, 1, s));
    std::swap(perm, other);
    HIP_TRY(hipStreamSynchronize(s));
  }
  uint64_t *ids2 = (uint64_t *)scratch_alloc(n * 8), *dev2 = (uint64_t *)scratch_alloc(n * 8);
  if (!ids2 || !dev2) { scratch_free(ids2); scratch_free(dev2); return set_error(LLKV_INTERNAL, "device scratch allocation failed"); }
  HIP_TRY(hj_launch_gather_u64(sel->d_ids, perm, n, ids2, s));
  HIP_TRY(hj_launch_gather_u64(sel->d_dev, perm, n, dev2, s));
  HIP_TRY(hipStreamSynchronize(s));
  scratch_free(sel->d_ids);
  scratch_free(sel->d_dev);
  sel->d_ids = ids2;
  sel->d_dev = dev2;
  return LLKV_OK;
}

Selection::~Selection() {
  scratch_free(d_ids);
  scratch_free(d_dev);
}

} // namespace llkv

using namespace llkv;

extern "C" {

llkv_status llkv_hip_filter_row_ids(const llkv_hip_table *table, const llkv_filter *filters, uint32_t n_filters,
                                    const llkv_eval_op *ops, uint32_t n_ops, uint64_t **out_row_ids, uint64_t *out_len) {
  if (!out_row_ids || !out_len) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL output");
  Selection sel;
  int rc = run_selection(reinterpret_cast<const Table *>(table), filters, n_filters, ops, n_ops, &sel);
  if (rc) return (llkv_status)rc;
  if ((rc = selection_report_ids(reinterpret_cast<const Table *>(table), &sel))) return (llkv_status)rc;
  bool large = sel.n * 8 >= (1u << 20);
  uint64_t *ids = large ? (uint64_t *)result_acquire(sel.n * 8) : nullptr;
  if (!ids) { // small, or no pinned memory to be had: pageable memory through the staging lanes
    large = false;
    ids = (uint64_t *)std::malloc(sel.n ? sel.n * 8 : 8);
  }
  if (!ids) return (llkv_status)set_error(LLKV_INTERNAL, "out of memory");
  // run_selection has synchronised: the ids are complete on the device
  if (large) {
    if (hipMemcpy(ids, sel.d_ids, sel.n * 8, hipMemcpyDeviceToHost) != hipSuccess) {
      result_release(ids);
      return (llkv_status)set_error(LLKV_INTERNAL, "copy of row ids failed");
    }
  } else if ((rc = fetch_to_host(ids, sel.d_ids, sel.n * 8))) {
    std::free(ids);
    return (llkv_status)rc;
  }
  *out_row_ids = ids;
  *out_len = sel.n;
  return LLKV_OK;
}

llkv_status llkv_hip_scan_stream(const llkv_hip_table *table, const llkv_projection *projections, uint32_t n_projections,
                                 const llkv_filter *filters, uint32_t n_filters, const llkv_eval_op *ops, uint32_t n_ops,
                                 const llkv_scan_options *options, llkv_on_batch on_batch, void *user) {
  const Table *t = reinterpret_cast<const Table *>(table);
  int rc = ensure_device();
  if (rc) return (llkv_status)rc;
  if (!t || !on_batch) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  auto resolve = [&](uint32_t fid) -> const ColumnInfo * {
    auto it = t->cols.find(fid);
    return it == t->cols.end() ? nullptr : &it->second.info;
  };
  // projections are validated before any row is touched (execute_scan builds the output schema first,
  // llkv-scan/src/execute.rs:66-186)
  LoweredPlan proj;
  std::string err;
  if ((rc = lower_projection(resolve, projections, n_projections, &proj, &err))) return (llkv_status)set_error(rc, err);
  Selection sel;
  // include_nulls = false → GatherNullPolicy::DropNulls over the gathered fields (projected columns and the
  // inputs of computed projections; llkv-scan/src/row_stream.rs:451-623)
  std::vector<uint32_t> gathered;
  if (!(options && options->include_nulls)) {
    for (uint32_t i = 0; i < n_projections; ++i) {
      if (!projections[i].computed) gathered.push_back(projections[i].field_id);
      else for (uint32_t k = 0; k < projections[i].expr_len; ++k)
        if (projections[i].expr[k].kind == LLKV_TOK_COLUMN) gathered.push_back(projections[i].expr[k].field_id);
    }
  }
  if ((rc = run_selection(t, filters, n_filters, ops, n_ops, &sel, gathered.data(), (uint32_t)gathered.size()))) return (llkv_status)rc;
  if (sel.n == 0) return LLKV_OK; // a filter that matches nothing yields no batch
  if (options && options->order_enabled && (rc = sort_selection(t, options, &sel))) return (llkv_status)rc;
  if (options && options->include_row_ids && (rc = selection_report_ids(t, &sel))) return (llkv_status)rc;
  JitKernel k;
  if ((rc = jit_compile(JitKind::Project, proj.type_string, &k, &err))) return (llkv_status)set_error(rc, err);

  hipStream_t stream = g_ctx.stream;
  const uint32_t n_out = (uint32_t)proj.out_dtypes.size();
  const bool with_ids = options && options->include_row_ids;
  // Two buffers of up to kSuper reference windows each: one gather launch and one copy per output fill a buffer
  // (65 536-row launches and 512 KB copies leave the PCIe link half idle), the host then hands out its windows
  // one by one — same batches, same order — while the device fills the other buffer.
  constexpr uint64_t kSuper = 16;
  const uint64_t buf_rows = std::min<uint64_t>(kSuper * kRowStreamChunk, (sel.n + kRowStreamChunk - 1) / kRowStreamChunk * kRowStreamChunk);
  struct Win {
    DeviceBuf d[kMaxOuts], d_valid[kMaxOuts], d_err; // d_err: one arithmetic-error cell per reference window of the buffer
    PinnedBuf h[kMaxOuts], h_valid[kMaxOuts], h_ids, h_err;
    hipEvent_t done = nullptr;
    uint32_t n = 0;
    ~Win() { if (done) (void)hipEventDestroy(done); }
  } win[2];
  // declared after the windows, so it runs before their buffers go back to the pools: on an early return the gather
  // kernels and copies that are still in flight finish first
  struct Drain {
    hipStream_t s;
    bool armed = true;
    ~Drain() { if (armed) (void)hipStreamSynchronize(s); }
  } drain{stream};
  for (auto &w : win) {
    if ((rc = w.d_err.alloc(kSuper * 4)) || (rc = w.h_err.alloc(kSuper * 4))) return (llkv_status)rc;
    for (uint32_t o = 0; o < n_out; ++o) {
      const size_t bytes = (size_t)buf_rows * dtype_out_width(proj.out_dtypes[o]);
      if ((rc = w.d[o].alloc(bytes)) || (rc = w.h[o].alloc(bytes))) return (llkv_status)rc;
      if (proj.out_nullable[o] && ((rc = w.d_valid[o].alloc(buf_rows / 8)) || (rc = w.h_valid[o].alloc(buf_rows / 8)))) return (llkv_status)rc;
    }
    if (with_ids && (rc = w.h_ids.alloc((size_t)buf_rows * 8))) return (llkv_status)rc;
    if (hipEventCreateWithFlags(&w.done, hipEventDisableTiming) != hipSuccess) return (llkv_status)set_error(LLKV_INTERNAL, "event create failed");
  }
  ProjParams pp;
  std::memset(&pp, 0, sizeof pp);
  for (size_t s = 0; s < proj.slot_fields.size(); ++s) pp.col[s] = slot_buffer(t->cols, proj, s);
  for (size_t i = 0; i < proj.lit_i.size(); ++i) pp.lit_i[i] = proj.lit_i[i];
  for (size_t i = 0; i < proj.lit_f.size(); ++i) pp.lit_f[i] = proj.lit_f[i];
  pp.error_stride = (uint32_t)kRowStreamChunk;

  auto enqueue = [&](uint64_t w0, Win &w) -> int {
    w.n = (uint32_t)std::min<uint64_t>(buf_rows, sel.n - w0);
    ProjParams q = pp;
    q.dev_rows = sel.d_dev + w0;
    q.n = w.n;
    for (uint32_t o = 0; o < n_out; ++o) { q.out[o] = w.d[o].p; q.out_valid[o] = (uint64_t *)w.d_valid[o].p; }
    q.error_flag = (uint32_t *)w.d_err.p;
    HIP_TRY(hipMemsetAsync(w.d_err.p, 0, kSuper * 4, stream));
    int r = jit_launch_raw(k.fn, (w.n + kBlock - 1) / kBlock, &q, sizeof q, stream);
    if (r) return r;
    for (uint32_t o = 0; o < n_out; ++o)
    {
      HIP_TRY(hipMemcpyAsync(w.h[o].p, w.d[o].p, (size_t)w.n * dtype_out_width(proj.out_dtypes[o]), hipMemcpyDeviceToHost, stream));
      if (proj.out_nullable[o]) HIP_TRY(hipMemcpyAsync(w.h_valid[o].p, w.d_valid[o].p, (size_t)((w.n + 63) / 64) * 8, hipMemcpyDeviceToHost, stream));
    }
    if (with_ids) HIP_TRY(hipMemcpyAsync(w.h_ids.p, sel.d_ids + w0, (size_t)w.n * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(w.h_err.p, w.d_err.p, kSuper * 4, hipMemcpyDeviceToHost, stream)); // travels with the window
    HIP_TRY(hipEventRecord(w.done, stream));
    return LLKV_OK;
  };

  // dictionaries of passthrough Utf8 columns
  std::vector<std::vector<const char *>> dicts(n_out);
  for (uint32_t o = 0; o < n_out; ++o)
    if (proj.out_dtypes[o] == LLKV_DT_UTF8 && proj.out_fields[o] >= 0)
      for (auto &s : t->cols.at((uint32_t)proj.out_fields[o]).info.dictionary) dicts[o].push_back(s.c_str());

  int cur = 0;
  if ((rc = enqueue(0, win[0]))) return (llkv_status)rc;
  for (uint64_t w0 = 0; w0 < sel.n; w0 += buf_rows) {
    const uint64_t next = w0 + buf_rows;
    if (next < sel.n && (rc = enqueue(next, win[cur ^ 1]))) return (llkv_status)rc;
    Win &w = win[cur];
    if (hipEventSynchronize(w.done) != hipSuccess) return (llkv_status)set_error(LLKV_INTERNAL, "window copy failed");
    for (uint64_t r0 = 0; r0 < w.n; r0 += kRowStreamChunk) { // the reference's windows, one callback each
      // a computed projection that failed in this window: the windows before it have been delivered, this one is not
      // (the reference's arrow kernel fails the batch it is evaluating)
      if (const uint32_t e = static_cast<const uint32_t *>(w.h_err.p)[r0 / kRowStreamChunk]) return (llkv_status)set_error(LLKV_INTERNAL, arith_error_message(e));
      llkv_column_view cols[kMaxOuts];
      for (uint32_t o = 0; o < n_out; ++o) {
        cols[o].dtype = proj.out_dtypes[o];
        cols[o].values = (const char *)w.h[o].p + r0 * dtype_out_width(proj.out_dtypes[o]);
        cols[o].validity = proj.out_nullable[o] ? (const uint8_t *)w.h_valid[o].p + r0 / 8 : nullptr;
        cols[o].dictionary = dicts[o].empty() ? nullptr : dicts[o].data();
        cols[o].precision = cols[o].scale = 0;
        if (proj.out_dtypes[o] == LLKV_DT_DECIMAL128 && proj.out_fields[o] >= 0) {
          const ColumnInfo &ci = t->cols.at((uint32_t)proj.out_fields[o]).info;
          cols[o].precision = ci.precision;
          cols[o].scale = ci.scale;
        }
      }
      llkv_batch_view b;
      b.num_rows = (uint32_t)std::min<uint64_t>(kRowStreamChunk, w.n - r0);
      b.num_columns = n_out;
      b.columns = cols;
      b.row_ids = with_ids ? (const uint64_t *)w.h_ids.p + r0 : nullptr;
      on_batch(&b, user); // calling thread, ascending row-id order
    }
    cur ^= 1;
  }
  drain.armed = false; // every window has been waited for
  return LLKV_OK;
}

llkv_status llkv_hip_join_stream(const llkv_hip_table *left, const llkv_hip_table *right, const llkv_join_key *keys,
                                 uint32_t n_keys, const llkv_join_options *options, llkv_on_join_batch on_batch, void *user) {
  return (llkv_status)run_join(reinterpret_cast<const Table *>(left), reinterpret_cast<const Table *>(right), keys, n_keys, options, on_batch, user);
}

llkv_status llkv_hip_join_stream_batches(const llkv_hip_table *left, const llkv_hip_table *right, const llkv_join_key *keys, uint32_t n_keys,
                                         const llkv_join_options *options, const llkv_join_output *output, llkv_on_join_record_batch on_batch,
                                         void *user) {
  return (llkv_status)run_join_batches(reinterpret_cast<const Table *>(left), reinterpret_cast<const Table *>(right), keys, n_keys, options, output, on_batch, user);
}

llkv_status llkv_hip_join_output_names(const llkv_join_output *output, int32_t join_type, int32_t key_rules, char **names, uint32_t *n_names) {
  return (llkv_status)join_output_names_c(output, join_type, key_rules, names, n_names);
}

} // extern "C"
