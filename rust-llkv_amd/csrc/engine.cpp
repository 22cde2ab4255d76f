// engine.cpp — host side of the MI355X execution path behind include/llkv_hip.h: device context and prepared
// queries (plan lowering → kernel selection → launch → canonical fold → finalize).  Table images: table.cpp;
// memory pools and staging lanes: memory.cpp; selection-vector route: stream.cpp; joins: join.cpp, join_agg.cpp;
// sort-based GROUP BY: group_sort.cpp.
//
// The host logic mirrors the reference's operator interfaces:
//   StorageTable::scan_stream / filter_row_ids   llkv-executor/src/types/storage.rs:20-50
//   execute_aggregates / compute_aggregate_values llkv-executor/src/lib.rs:5357-5682,6087-6665
//   execute_group_by_with_aggregates              llkv-executor/src/lib.rs:5028-5355
//   AggregateAccumulator::finalize                llkv-aggregate/src/lib.rs:1488-1939
// There is no CPU fallback in here: without a HIP device every data-path call fails
// with LLKV_NO_DEVICE.
#include "engine.hpp"
#include "join.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unordered_map>
#include <unordered_set>

namespace llkv {

thread_local std::string g_last_error;

int set_error(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}
Context g_ctx;

int ensure_device() {
  if (!g_ctx.ready) return set_error(LLKV_NO_DEVICE, "llkv_hip_init() has not bound a HIP device (no GPU path without one)");
  // the current HIP device is per host thread: a call arriving on another thread of the process (the traits
  // are Send + Sync, llkv-executor/src/types/storage.rs:20-50) must land on the GPU this process is bound to
  static thread_local int bound = -1;
  if (bound != g_ctx.device) {
    if (hipSetDevice(g_ctx.device) != hipSuccess) return set_error(LLKV_NO_DEVICE, "hipSetDevice failed on this thread");
    bound = g_ctx.device;
  }
  return LLKV_OK;
}

// ---------------------------------------------------------------------------------
// Query
// ---------------------------------------------------------------------------------
Query::~Query() {
  if (sorted) sorted_groupby_free(sorted);
  if (join_state) join_group_state_free(join_state); // (behind the GROUP BY that tests its bitmap)
  // the buffers go back to the pools (hipFree / hipHostFree cost 0.2 ms per statement); executions that were
  // launched and never collected may still be running on their streams
  if (n_launched != n_collected) {
    if (pending_stream) (void)hipStreamSynchronize(pending_stream);
    for (hipStream_t st : slot_stream) if (st) (void)hipStreamSynchronize(st);
    (void)hipStreamSynchronize(g_ctx.stream);
  }
  scratch_free(d_tile_partials);
  scratch_free(d_dict_num);
  if (d_exchange && !host_mapped) scratch_free(d_exchange);
  scratch_free(d_lane_ops);
  scratch_free(d_empty_image);
  if (h_exchange) pinned_release(h_exchange, h_exchange_bytes);
  for (auto &e : events) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  for (auto &e : copied) if (e) (void)hipEventDestroy(e);
  for (auto &e : ev_fold) if (e) (void)hipEventDestroy(e);
}

// The tile is the canonical unit of the reduction (one partial per tile — per (tile, wave) for LDS-resident states —
// folded in order): its length depends on the plan alone, never on the table size, the shard or the GPU count, so
// results are bit-identical across launch geometries.  Register-resident states amortise their block reduction quickly
// and like many small tiles (4 096 rows); LDS-resident grouped states use 16 384-row tiles and let a workgroup stream
// several of them (pick_scan_grid).
static uint32_t pick_tile_rows(const LoweredPlan &p) {
  // shared-image plans: tiles are only a work list (every lane is order-free); one tile = the plan's U steps of 2 048 rows,
  // which the kernel requests as one batch of loads (image_scan_body) — not tunable
  if (p.acc_image) return 2048u * (uint32_t)p.unroll;
  if (const char *e = std::getenv("LLKV_HIP_TILE_ROWS")) {
    long v = std::atol(e);
    if (v >= 512 && v % 512 == 0) return (uint32_t)v;
  }
  // (the late form of a register-state plan requests the next group of U steps while it works on this one — inside a tile:
  // 8 192 rows = four groups; Q6 SF10 226 / 212 / 219 µs and SF1 32.2 / 28.7 / 31.5 µs at 4 096 / 8 192 / 16 384)
  return p.acc_lds ? 16384u : p.late_columns ? 8192u : 4096u;
}

// Workgroups of a shared-image scan (1024 threads each, persistent: workgroup b takes tiles b, b + g, …): one per CU.
// The kernel is bound by the rate of DS atomics to scattered addresses (~60–100 cycles per 64-lane instruction: GROUP BY
// l_shipdate with only count(*) — 4 B/row — still takes 0.15 ms), and a second workgroup on a CU adds nothing but its
// image: 512 workgroups 0.25 ms, 1 024: 0.47 ms (profiles/r02/pmc_image.txt).
static uint32_t pick_image_grid(const LoweredPlan &p, uint32_t n_tiles) {
  uint32_t grid = g_ctx.cu_count;
  if (const char *e = std::getenv("LLKV_HIP_IMAGE_WGS")) {
    long v = std::atol(e);
    if (v >= 1) grid = (uint32_t)std::min<long>(v, 1 << 16);
  }
  grid = std::max(grid, p.image_min_grid); // fewer workgroups: an image would see more rows than its fixed-point lanes were sized for
  return std::max(1u, std::min(grid, n_tiles));
}

// Launch geometry of an LDS-accumulator plan, from the LOCAL tile count: persistent-style, about one workgroup per
// CU, each streaming a contiguous run of tiles (workgroup b of g: tiles [b·n/g, (b+1)·n/g)).  Fewer, longer-running
// workgroups beat many short ones once the tile boundary is cheap, and on a table of thousands of tiles 160–208
// workgroups beat one per CU by 2–4 % on every box tried (profiles/r02/sweep_tiles_q1.txt: SF10 on one GPU, 3 662 tiles:
// 176 workgroups 358 µs, 192: 363, 256: 372, 512: 381, 128: 408; interleaving the tiles over the workgroups instead
// of contiguous runs changes nothing) — the scan sits at the device's multi-stream read ceiling and leaving a quarter of
// the CUs idle seems to buy the fabric some clock; a 1/8 shard (457 tiles) wants every CU: 256 workgroups 49.8 µs, 192:
// 56 µs.  A shard with fewer tiles than CUs launches one workgroup per tile.
static uint32_t pick_scan_grid(const LoweredPlan &p, uint32_t n_tiles) {
  if (!p.acc_lds) { // register-resident states: one tile per workgroup, unless the table is small enough for the dispatch of its
    // workgroups to show (kRegGridTiles, swept on Q6 SF1: profiles/r03/sweep_q6_sf1.txt)
    uint32_t grid = 0;
    if (const char *e = std::getenv("LLKV_HIP_SCAN_WGS")) {
      long v = std::atol(e);
      if (v >= 1) grid = (uint32_t)std::min<long>(v, 1 << 20);
    }
    return grid >= n_tiles ? 0 : grid;
  }
  // (a state of more than 6 lanes per row keeps every CU busy with its DS atomics: 12 lanes 0.334 ms at 256, 0.428 ms at 192)
  // Within that range the count that leaves the least idle time wins: workgroups take whole tiles, so 3 662 tiles over
  // 192 workgroups are 19 or 20 each (the kernel lasts 20, the average is 19.07: 352 µs), over 193 they are 18 or 19
  // (342 µs), over 216 or 229 likewise.
  // (the counts below were swept on the 256 CUs of an MI355X; another part keeps the proportions)
  const uint32_t cus = g_ctx.cu_count;
  uint32_t grid = cus;
  if (n_tiles >= 8 * cus) {
    const uint32_t lo = (p.k <= 6 ? 184u : 232u) * cus / 256u, hi = (p.k <= 6 ? 232u : 256u) * cus / 256u;
    // the smallest count that keeps the workgroups ≥ 99.5 % busy (else the busiest): on the slower boxes fewer
    // workgroups still win among balanced counts (193: 354 µs, 204: 356, 216: 357, 229: 361; 192: 362)
    double best = 0.0;
    for (uint32_t g = lo; g <= hi; ++g) {
      const uint32_t most = (n_tiles + g - 1) / g;
      const double busy = (double)n_tiles / ((double)most * g);
      if (busy >= 0.995) { grid = g; break; }
      if (busy > best + 1e-9) { best = busy; grid = g; }
    }
  }
  if (const char *e = std::getenv("LLKV_HIP_SCAN_WGS")) {
    long v = std::atol(e);
    if (v >= 1) grid = (uint32_t)std::min<long>(v, 1 << 20);
  }
  return std::max(1u, std::min(grid, n_tiles));
}

int prepare_query(const Table *table, const llkv_filter *filters, uint32_t n_filters, const llkv_eval_op *ops,
                  uint32_t n_ops, const uint32_t *key_fields, uint32_t n_keys, const llkv_aggregate_spec *aggs,
                  uint32_t n_aggs, bool grouped, bool order_by_keys, Query **out) {
  int rc = ensure_device();
  if (rc) return rc;
  if (!table) return set_error(LLKV_INVALID_ARGUMENT, "table is NULL");
  auto resolve = [&](uint32_t fid) -> const ColumnInfo * {
    auto it = table->cols.find(fid);
    return it == table->cols.end() ? nullptr : &it->second.info;
  };
  std::unique_ptr<Query> q(new Query());
  q->table = table;
  q->table_generation = table->generation;
  q->order_by_keys = order_by_keys;
  q->n_user_aggs = n_aggs;
  q->n_user_keys = grouped ? n_keys : 0;
  std::string err;
  // ungrouped DISTINCT aggregates are evaluated by their own sort-based pipeline at finish; the fused scan
  // carries a COUNT(*) in their place.  MIN / MAX ignore DISTINCT.
  std::vector<llkv_aggregate_spec> subst;
  if (!grouped) {
    bool any = false;
    for (uint32_t a = 0; a < n_aggs; ++a) any |= aggs[a].distinct != 0;
    if (any) {
      subst.assign(aggs, aggs + n_aggs);
      q->distinct.resize(n_aggs);
      for (uint32_t a = 0; a < n_aggs; ++a) {
        if (!aggs[a].distinct) continue;
        subst[a].distinct = 0;
        const int kind = aggs[a].kind;
        if (kind == LLKV_AGG_MIN || kind == LLKV_AGG_MAX || kind == LLKV_AGG_COUNT_STAR) continue;
        if (kind != LLKV_AGG_COUNT && kind != LLKV_AGG_SUM && kind != LLKV_AGG_TOTAL && kind != LLKV_AGG_AVG)
          return set_error(LLKV_UNSUPPORTED, "DISTINCT form of aggregate kind " + std::to_string(kind));
        if (!aggs[a].expr || !aggs[a].expr_len) return set_error(LLKV_INVALID_ARGUMENT, "aggregate requires an argument");
        Query::DistinctAgg &da = q->distinct[a];
        if ((rc = lower_emit(resolve, filters, n_filters, ops, n_ops, aggs[a].expr, aggs[a].expr_len, &da.plan, &err, /*allow_f64=*/true, &da.is_f64, nullptr, &da.key_dtype)))
          return set_error(rc, err);
        if (da.key_dtype == LLKV_DT_UTF8 || da.key_dtype == LLKV_DT_DECIMAL128) {
          const ColumnInfo *ci = resolve(aggs[a].expr[0].field_id);
          da.precision = ci->precision; da.scale = ci->scale;
          if (da.key_dtype == LLKV_DT_UTF8 && kind != LLKV_AGG_COUNT) { // SUM / TOTAL / AVG over strings: Float64 accumulators fed by array_value_to_numeric (:400-449)
            for (const std::string &w : ci->dictionary) da.key_numeric.push_back(parse_numeric_or_zero(w));
          }
        }
        da.kind = kind;
        subst[a].kind = LLKV_AGG_COUNT_STAR;
      }
      aggs = subst.data();
    }
  }
  rc = lower_plan(resolve, filters, n_filters, ops, n_ops, key_fields, n_keys, aggs, n_aggs, grouped, /*track_first=*/!order_by_keys, &q->plan, &err);
  if (rc == LLKV_UNSUPPORTED && grouped && !std::getenv("LLKV_HIP_GROUP_NO_IMAGE")) {
    // more groups or a wider state than per-thread accumulator columns hold: ONE image per workgroup, shared by its
    // threads (image_scan_body) — as long as the image fits the LDS and every f64 sum has a bound that makes it exact
    std::string image_err;
    LoweredPlan ip;
    if (lower_plan(resolve, filters, n_filters, ops, n_ops, key_fields, n_keys, aggs, n_aggs, true, !order_by_keys, &ip, &image_err, /*image=*/true) == LLKV_OK) {
      q->route_note = "shared-image GROUP BY (per-thread accumulators: " + err + ")";
      q->plan = std::move(ip);
      rc = LLKV_OK;
    } else {
      err += "; shared-image route: " + image_err;
    }
  }
  if (rc == LLKV_UNSUPPORTED && grouped) {
    // sparse or unbounded integer keys, too many groups, unbounded f64 sums: sort-based route
    const std::string dense_err = err;
    if (sorted_groupby_prepare(table, filters, n_filters, ops, n_ops, key_fields, n_keys, aggs, n_aggs, order_by_keys, &q->sorted) == LLKV_OK) {
      q->route_note = std::string(sorted_groupby_partitioned(q->sorted) ? "partitioned GROUP BY (" : "sort-based GROUP BY (") + dense_err + ")";
      *out = q.release();
      return LLKV_OK;
    }
    return set_error(rc, dense_err + "; sort-based route: " + g_last_error);
  }
  if (rc) return set_error(rc, err);
  const LoweredPlan &p = q->plan;
  if (!grouped) {
    q->exact_plans.resize(n_aggs);
    for (uint32_t a = 0; a < n_aggs; ++a)
      if (p.aggs[a].fin == AggFinal::SumI64 || p.aggs[a].fin == AggFinal::AvgI64) {
        std::string ignore;
        if (lower_emit(resolve, filters, n_filters, ops, n_ops, aggs[a].expr, aggs[a].expr_len, &q->exact_plans[a], &ignore) != LLKV_OK)
          q->exact_plans[a] = LoweredPlan{};
      }
  }

  if (!p.always_false) {
    q->entry = (std::getenv("LLKV_HIP_FORCE_JIT") || p.acc_image) ? nullptr : catalog_find(p.type_string.c_str());
    if (!q->entry) {
      rc = jit_compile(p.acc_image ? JitKind::Image : JitKind::Scan, p.type_string, &q->jit, &err);
      if (rc) return set_error(rc, err);
    }
  }
  const TileSet *ts = nullptr;
  if ((rc = get_tileset(*table, pick_tile_rows(p), &ts))) return rc;
  q->tiles = ts;

  std::memset(&q->params, 0, sizeof q->params);
  for (size_t s = 0; s < p.slot_fields.size(); ++s) q->params.col[s] = slot_buffer(table->cols, p, s);
  if (!p.dict_num.empty()) { // numeric images of the dictionaries some aggregate reads (DictNum<slot>)
    std::vector<double> image(p.slot_fields.size() * 256, 0.0);
    for (auto &d : p.dict_num) std::copy(d.second.begin(), d.second.end(), image.begin() + (size_t)d.first * 256);
    q->d_dict_num = (double *)scratch_alloc(image.size() * 8);
    if (!q->d_dict_num) return set_error(LLKV_INTERNAL, "device allocation failed");
    HIP_TRY(hipMemcpyAsync(q->d_dict_num, image.data(), image.size() * 8, hipMemcpyHostToDevice, g_ctx.stream));
    HIP_TRY(hipStreamSynchronize(g_ctx.stream)); // `image` is pageable and goes out of scope
    q->params.dict_num = q->d_dict_num;
  }
  for (size_t i = 0; i < p.lit_i.size(); ++i) q->params.lit_i[i] = p.lit_i[i];
  for (size_t i = 0; i < p.lit_f.size(); ++i) q->params.lit_f[i] = p.lit_f[i];
  for (size_t i = 0; i < p.key_strides.size(); ++i) q->params.key_stride[i] = p.key_strides[i];
  q->params.tiles = ts->d_tiles;
  q->params.n_tiles = ts->n_tiles;
  q->params.scan_grid = pick_scan_grid(p, ts->n_tiles);

  const size_t lanes = (size_t)p.lanes;
  const uint32_t parts_per_tile = p.acc_lds ? (uint32_t)(kBlock / 64) : 1u; // LDS-accumulator plans publish one partial per (tile, wave)
  q->image_grid = p.acc_image ? pick_image_grid(p, ts->n_tiles) : 0;
  const size_t image_slice_words = p.acc_image ? ((size_t)p.ng + p.image_passes - 1) / p.image_passes * p.k_image + 1 : 0;
  q->partials_len = std::max<size_t>(1, p.acc_image ? image_slice_words * q->image_grid * p.image_passes : lanes * ts->n_tiles * parts_per_tile);
  q->d_tile_partials = (uint64_t *)scratch_alloc(2 * q->partials_len * sizeof(uint64_t));
  // lane ops, then (shared-image plans) the fold's expansion tables: [lanes] ops, [k] source kernel lane, [k] transform
  std::vector<uint8_t> lane_tables(p.lane_ops.begin(), p.lane_ops.end());
  lane_tables.insert(lane_tables.end(), p.image_src.begin(), p.image_src.end());
  lane_tables.resize(lanes + p.k, 0);
  lane_tables.insert(lane_tables.end(), p.image_xf.begin(), p.image_xf.end());
  lane_tables.resize(lanes + 2 * (size_t)p.k, 0);
  q->d_lane_ops = (uint8_t *)scratch_alloc(lane_tables.size());
  if (!q->d_tile_partials || !q->d_lane_ops) return set_error(LLKV_INTERNAL, "device allocation failed");
  HIP_TRY(hipMemcpyAsync(q->d_lane_ops, lane_tables.data(), lane_tables.size(), hipMemcpyHostToDevice, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream)); // `lane_tables` is pageable and goes out of scope
  for (auto &e : q->ev_fold) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto &e : q->copied) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const size_t ring_bytes = Query::kMaxDepth * kOctantsHost * lanes * sizeof(uint64_t);
  q->h_exchange_bytes = ring_bytes;
  q->h_exchange = (uint64_t *)pinned_acquire(&q->h_exchange_bytes);
  if (!q->h_exchange) return set_error(LLKV_INTERNAL, "pinned host allocation failed");
  std::memset(q->h_exchange, 0, ring_bytes);
  // nobody else reads the image: let the kernel write it to the host directly (not the transposing fold of a
  // shared-image plan: thousands of scattered 8-byte stores belong in HBM, one copy brings the image over)
  q->host_mapped = table->world == 1 && !p.acc_image;
  if (q->host_mapped) q->d_exchange = q->h_exchange;
  else {
    q->d_exchange = (uint64_t *)scratch_alloc(ring_bytes);
    if (!q->d_exchange) return set_error(LLKV_INTERNAL, "device allocation failed");
    HIP_TRY(hipMemsetAsync(q->d_exchange, 0, ring_bytes, g_ctx.stream));
  }
  { // image of an execution that launches no workgroup: identities for owned octants, zero for the others
    std::vector<uint64_t> img(kOctantsHost * lanes, 0);
    for (int o = 0; o < kOctantsHost; ++o)
      if ((table->owned_mask >> o) & 1u)
        for (size_t l = 0; l < lanes; ++l) img[o * lanes + l] = p.lane_ops[l] == 2 ? 0x7FFFFFFFFFFFFFFFull : p.lane_ops[l] == 3 ? 0x8000000000000000ull : 0ull;
    q->d_empty_image = (uint64_t *)scratch_alloc(img.size() * 8);
    if (!q->d_empty_image) return set_error(LLKV_INTERNAL, "device allocation failed");
    // (on the stream of the memset above — it does not wait for the null stream — and complete before `img` goes)
    HIP_TRY(hipMemcpyAsync(q->d_empty_image, img.data(), img.size() * 8, hipMemcpyHostToDevice, g_ctx.stream));
    if (p.acc_image) // the fold of a shared-image plan only rewrites the first owned octant of a slot: the rest is constant
      for (uint32_t sl = 0; sl < Query::kMaxDepth; ++sl)
        HIP_TRY(hipMemcpyAsync(q->d_exchange + sl * q->exchange_len(), img.data(), img.size() * 8, hipMemcpyHostToDevice, g_ctx.stream));
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  }
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  q->params.tile_partials = q->d_tile_partials;
  for (int o = 0; o <= kOctantsHost; ++o) q->params.octant_tile_begin[o] = ts->octant_tile_begin[o];
  q->params.owned_mask = table->owned_mask;
  std::memset(&q->fold, 0, sizeof q->fold);
  q->fold.lane_ops = q->d_lane_ops;
  for (int o = 0; o <= kOctantsHost; ++o) q->fold.octant_tile_begin[o] = ts->octant_tile_begin[o];
  q->fold.n_tiles = ts->n_tiles;
  q->fold.lanes = (uint32_t)lanes;
  q->fold.owned_mask = table->owned_mask;
  q->fold.parts_per_tile = parts_per_tile;
  *out = q.release();
  return LLKV_OK;
}

// One execution = ONE kernel launch in steady state: the fused scan of execution i, whose first workgroups
// also fold the tile partials of execution i-1 into that execution's exchange image (two partial images).
// Whatever is still unfolded when its result is asked for is flushed by the standalone fold kernel.
int Query::flush_pending() {
  if (!pending) return LLKV_OK;
  FoldParams f = fold;
  f.tile_partials = d_tile_partials + pending_pb * partials_len;
  f.exchange = d_exchange + pending_slot * exchange_len();
  HIP_TRY(launch_fold_octants(f, pending_stream));
  HIP_TRY(hipEventRecord(ev_fold[pending_slot], pending_stream));
  pending = false;
  return LLKV_OK;
}

int Query::launch(hipStream_t stream) {
  if (table && table->generation != table_generation)
    return set_error(LLKV_INVALID_ARGUMENT, "the table was appended to after this query was prepared (its buffers, statistics and tile lists have changed): prepare it again");
  if (sorted) { // the sort-based route runs to completion here; submit / collect only hand the result over
    if (n_launched != n_collected) return set_error(LLKV_INVALID_ARGUMENT, "a sort-based GROUP BY keeps one execution in flight");
    const int rc = sorted_groupby_run(sorted, &lazy);
    if (rc) return rc;
    n_launched++;
    return LLKV_OK;
  }
  if (!stream) stream = g_ctx.stream;
  if (n_launched - n_collected >= depth)
    return set_error(LLKV_INVALID_ARGUMENT, "query pipeline is full: collect a finished execution first (depth " + std::to_string(depth) + ")");
  const uint32_t slot = (uint32_t)(n_launched % depth);
  const uint32_t pb = (uint32_t)(n_launched & 1);
  const bool run_main = !plan.always_false && tiles->n_tiles > 0;
  uint64_t *image = d_exchange + slot * exchange_len();
  const uint32_t fold_blocks = (uint32_t)kOctantsHost * (uint32_t)((plan.lanes + kBlock / 64 - 1) / (kBlock / 64));
  bool piggy = false;
  int rc;
  if (plan.acc_image) {
    // shared-image GROUP BY: scan (persistent workgroups, one LDS image each) + fold of the workgroup images into the
    // slot's exchange image, back to back on the stream; nothing is deferred to the next launch
    if (pending && (rc = flush_pending())) return rc;
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    if (run_main && profiling && (launches % profile_every) == 0) {
      if (events_used == events.size()) {
        HIP_TRY(hipEventCreate(&ev.first));
        HIP_TRY(hipEventCreate(&ev.second));
        events.push_back(ev);
      }
      ev = events[events_used++];
      HIP_TRY(hipEventRecord(ev.first, stream));
    }
    if (run_main) {
      const uint32_t ngs = (plan.ng + plan.image_passes - 1) / plan.image_passes;
      for (int pass = 0; pass < plan.image_passes; ++pass) { // one scan per slice of the groups
        ScanParams p = params;
        p.group_base = (uint32_t)pass * ngs;
        p.tile_partials = d_tile_partials + (size_t)pass * image_grid * ((size_t)ngs * plan.k_image + 1);
        if ((rc = jit_launch_raw(jit.fn, image_grid, &p, sizeof p, stream, 1024))) return rc;
      }
      HIP_TRY(launch_image_fold(d_tile_partials, image, d_lane_ops, image_grid, plan.ng, (uint32_t)plan.k, table->owned_mask, (uint32_t)plan.image_passes,
                                (uint32_t)plan.k_image, d_lane_ops + plan.lanes, d_lane_ops + plan.lanes + plan.k, stream));
    } else {
      HIP_TRY(hipMemcpyAsync(image, d_empty_image, exchange_len() * sizeof(uint64_t), hipMemcpyDefault, stream));
    }
    if (ev.second) HIP_TRY(hipEventRecord(ev.second, stream));
    HIP_TRY(hipEventRecord(ev_fold[slot], stream));
    slot_stream[slot] = stream;
    launches++;
    n_launched++;
    return LLKV_OK;
  }
  const uint32_t grid = params.scan_grid ? params.scan_grid : tiles->n_tiles;
  if (pending && !(run_main && grid >= fold_blocks && stream == pending_stream) && (rc = flush_pending())) return rc;
  if (run_main) {
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    // event pairs bracket every `profile_every`-th scan: each record is a packet between back-to-back kernels
    if (profiling && (launches % profile_every) == 0) {
      if (events_used == events.size()) {
        HIP_TRY(hipEventCreate(&ev.first));
        HIP_TRY(hipEventCreate(&ev.second));
        events.push_back(ev);
      }
      ev = events[events_used++];
      HIP_TRY(hipEventRecord(ev.first, stream));
    }
    ScanParams p = params;
    p.tile_partials = d_tile_partials + pb * partials_len;
    if (pending) {
      p.prev_partials = d_tile_partials + pending_pb * partials_len;
      p.prev_exchange = d_exchange + pending_slot * exchange_len();
      piggy = true;
    }
    if (entry) HIP_TRY(entry->launch(p, stream));
    else if ((rc = jit_launch(jit, p, stream))) return rc;
    if (ev.second) HIP_TRY(hipEventRecord(ev.second, stream));
    if (piggy) HIP_TRY(hipEventRecord(ev_fold[pending_slot], stream));
    pending = true;
    pending_slot = slot;
    pending_pb = pb;
    pending_stream = stream;
  } else {
    HIP_TRY(hipMemcpyAsync(image, d_empty_image, exchange_len() * sizeof(uint64_t), hipMemcpyDefault, stream));
    HIP_TRY(hipEventRecord(ev_fold[slot], stream));
  }
  slot_stream[slot] = stream;
  launches++;
  n_launched++;
  return LLKV_OK;
}

static inline uint64_t host_identity(int op) { return op == 2 ? 0x7FFFFFFFFFFFFFFFull : op == 3 ? 0x8000000000000000ull : 0ull; }
static inline uint64_t host_combine(int op, uint64_t a, uint64_t b) {
  switch (op) {
  case 0: { double x, y; std::memcpy(&x, &a, 8); std::memcpy(&y, &b, 8); double z = x + y; uint64_t r; std::memcpy(&r, &z, 8); return r; }
  case 1: return a + b;
  case 2: return (int64_t)b < (int64_t)a ? b : a;
  case 3: return (int64_t)b > (int64_t)a ? b : a;
  default: return b > a ? b : a;
  }
}

// Fold the 8 octant partials in octant order (same association for every GPU count).
void fold_exchange_host(const uint64_t *exchange, const uint8_t *lane_ops, uint32_t lanes, uint64_t *state) {
  for (uint32_t l = 0; l < lanes; ++l) {
    uint64_t v = host_identity(lane_ops[l]);
    for (int o = 0; o < kOctantsHost; ++o) v = host_combine(lane_ops[l], v, exchange[(size_t)o * lanes + l]);
    state[l] = v;
  }
}

typedef __int128 i128;
typedef unsigned __int128 u128;

static double key_to_f64(int64_t key) {
  int64_t b = key < 0 ? (key ^ 0x7FFFFFFFFFFFFFFFll) : key;
  double d;
  std::memcpy(&d, &b, 8);
  return d;
}

// Finalize one aggregate of one group: AggregateAccumulator::finalize
// llkv-aggregate/src/lib.rs:1488-1939 on the folded lane state.
int finalize_value(const AggOut &a, const uint64_t *g /*group lanes*/, int base, llkv_value *out, std::string *err, bool prefixes_checked) {
  std::memset(out, 0, sizeof *out);
  // rows the accumulator saw: the group's rows, or the non-NULL argument rows when the argument has NULL cells
  const int64_t rows = a.count_lane >= 0 ? (int64_t)g[base + a.count_lane] : (int64_t)g[0];
  const uint64_t *l = a.lane >= 0 ? g + base + a.lane : nullptr;
  auto as_f64 = [](uint64_t b) { double d; std::memcpy(&d, &b, 8); return d; };
  // an f64 sum: one lane, or the exact grid-level lanes of the shared-image plans (fused_scan.hip.h: SumF64X), added
  // smallest level first
  auto f64_sum = [&]() {
    if (a.fixed_point) { // SumF64Q: grid steps as low 32 bits + high part, both summed over workgroups, octants and ranks
      const i128 steps = ((i128)(int64_t)l[1] << 32) + (i128)(int64_t)l[0];
      return std::ldexp((double)steps, a.fixed_exp); // (the conversion rounds to nearest even; the scaling is exact)
    }
    if (a.exact_levels <= 1) return as_f64(l[0]);
    double v = as_f64(l[a.exact_levels - 1]);
    for (int j = a.exact_levels - 2; j >= 0; --j) v += as_f64(l[j]);
    return v;
  };
  auto exact_sum = [&](int64_t *sum, const char *overflow_msg) -> int {
    const i128 total = ((i128)(int64_t)l[1] << 32) + (i128)(u128)l[0];
    if (total > (i128)INT64_MAX || total < (i128)INT64_MIN) { *err = overflow_msg; return LLKV_INVALID_ARGUMENT; }
    if (!prefixes_checked && (u128)l[2] * (u128)(uint64_t)rows > (u128)INT64_MAX) {
      // the reference's checked_add chain is order dependent: a prefix may overflow although
      // the total fits.  Not decidable from the order-free state → caller's CPU route decides.
      *err = "possible intermediate i64 overflow in SUM: order-dependent check is not on the GPU path";
      return LLKV_UNSUPPORTED;
    }
    *sum = (int64_t)total;
    return LLKV_OK;
  };
  if (a.typed_by_first_value && rows == 0 && (a.fin == AggFinal::SumF64 || a.fin == AggFinal::MinF64 || a.fin == AggFinal::MaxF64 ||
                                              a.fin == AggFinal::SumDec || a.fin == AggFinal::MinDec || a.fin == AggFinal::MaxDec)) {
    out->dtype = LLKV_DT_INT64; // an all-NULL temp column is an Int64 column: SUM / MIN / MAX come back as Int64 NULLs
    out->is_null = 1;
    return LLKV_OK;
  }
  int32_t dec_precision = a.precision;
  if (a.digits_lane >= 0) {
    // a computed DECIMAL argument of a GROUP BY: the group's temp column is Decimal128(digits of its first non-NULL value, scale)
    // (plan_values_to_arrow_array llkv-executor/src/lib.rs:298-330) — arrow-rs refuses a positive scale above the precision,
    // whatever the aggregate; without a non-NULL value the column is Int64 (AVG → Float64 NULL, TOTAL → 0.0)
    if (rows == 0) {
      if (a.fin == AggFinal::AvgDec) { out->dtype = LLKV_DT_FLOAT64; out->is_null = 1; return LLKV_OK; }
      if (a.fin == AggFinal::TotalDec) { out->dtype = LLKV_DT_FLOAT64; out->f64 = 0.0; return LLKV_OK; }
    } else {
      dec_precision = (int32_t)((g[base + a.digits_lane] >> a.digits_shift) & 63u);
      if (a.scale > 0 && a.scale > dec_precision) {
        *err = "invalid Decimal128 precision/scale: scale " + std::to_string(a.scale) + " is greater than precision " + std::to_string(dec_precision);
        return LLKV_INVALID_ARGUMENT;
      }
    }
  }
  if (a.fin == AggFinal::SumDec || a.fin == AggFinal::TotalDec || a.fin == AggFinal::AvgDec || a.fin == AggFinal::MinDec || a.fin == AggFinal::MaxDec) {
    // Decimal128 finalize (llkv-aggregate/src/lib.rs:1567-1582,1640-1655,1720-1760): the 64-bit values cannot
    // carry a sum of < 2^63 rows out of i128, so "Decimal128 sum overflow" is unreachable on this path
    out->dtype = LLKV_DT_DECIMAL128;
    out->precision = dec_precision;
    out->scale = a.scale;
    i128 v = 0;
    if (a.fin == AggFinal::MinDec || a.fin == AggFinal::MaxDec) {
      out->is_null = rows == 0;
      if (a.wide_delta) { // values beyond 64 bits: the lane is the largest v − min(column) (MAX) or max(column) − v (MIN)
        const i128 base = (i128)(((u128)a.wide_base_hi << 64) | a.wide_base_lo);
        v = !rows ? 0 : a.wide_delta == 1 ? base + (i128)(u128)l[0] : base - (i128)(u128)l[0];
      } else v = rows ? (i128)(int64_t)l[0] : 0;
    } else {
      // (wide values: the limb sums mod 2^128 — the lowering excluded a prefix outside i128, so the true sum is inside)
      const i128 sum = a.wide ? (i128)((u128)l[0] + ((u128)l[1] << 32) + ((u128)l[2] << 64) + ((u128)l[3] << 96))
                              : a.fast_sum ? (i128)(int64_t)l[0] : (((i128)(int64_t)l[1] << 32) + (i128)(u128)l[0]);
      if (a.fin == AggFinal::AvgDec) {
        if (rows > 0) { // sum / count, rounded half away from zero
          const i128 n = rows, rem = sum % n;
          v = sum / n;
          if ((rem < 0 ? -rem : rem) * 2 >= n) v += sum > 0 ? 1 : -1;
        } else out->is_null = 1;
      } else if (a.null_without_values && rows == 0) out->is_null = 1; // SUM(DISTINCT) over no value
      else v = sum; // SUM / TOTAL: `vec![sum]` — 0, not NULL, without rows
    }
    out->i64 = (int64_t)(uint64_t)v;
    out->i64_hi = (int64_t)(v >> 64);
    return LLKV_OK;
  }
  switch (a.fin) {
  case AggFinal::CountRows: out->dtype = LLKV_DT_INT64; out->i64 = rows; return LLKV_OK;
  case AggFinal::CountNullsZero: out->dtype = LLKV_DT_INT64; out->i64 = 0; return LLKV_OK;
  case AggFinal::SumDec: case AggFinal::TotalDec: case AggFinal::AvgDec: case AggFinal::MinDec: case AggFinal::MaxDec: break; // handled above
  case AggFinal::CountValid: out->dtype = LLKV_DT_INT64; out->i64 = (int64_t)l[0]; return LLKV_OK;
  case AggFinal::CountNulls: out->dtype = LLKV_DT_INT64; out->i64 = (int64_t)g[0] - (int64_t)l[0]; return LLKV_OK;
  case AggFinal::SumI64Fast: out->dtype = LLKV_DT_INT64; out->is_null = rows == 0; out->i64 = rows ? (int64_t)l[0] : 0; return LLKV_OK;
  case AggFinal::SumI64: {
    out->dtype = LLKV_DT_INT64;
    if (rows == 0) { out->is_null = 1; return LLKV_OK; }
    return exact_sum(&out->i64, "integer overflow");
  }
  case AggFinal::SumF64: out->dtype = LLKV_DT_FLOAT64; out->is_null = rows == 0; out->f64 = rows ? f64_sum() : 0.0; return LLKV_OK;
  case AggFinal::TotalF64: out->dtype = LLKV_DT_FLOAT64; out->f64 = f64_sum(); return LLKV_OK;
  case AggFinal::AvgI64Fast: out->dtype = LLKV_DT_FLOAT64; out->is_null = rows == 0; if (rows) out->f64 = (double)(int64_t)l[0] / (double)rows; return LLKV_OK;
  case AggFinal::AvgI64: {
    out->dtype = LLKV_DT_FLOAT64;
    if (rows == 0) { out->is_null = 1; return LLKV_OK; }
    int64_t s;
    int rc = exact_sum(&s, "AVG aggregate sum exceeds i64 range");
    if (rc) return rc;
    out->f64 = (double)s / (double)rows;
    return LLKV_OK;
  }
  case AggFinal::AvgF64: out->dtype = LLKV_DT_FLOAT64; out->is_null = rows == 0; if (rows) out->f64 = f64_sum() / (double)rows; return LLKV_OK;
  case AggFinal::MinI64: case AggFinal::MaxI64: out->dtype = LLKV_DT_INT64; out->is_null = rows == 0; out->i64 = rows ? (int64_t)l[0] : 0; return LLKV_OK;
  case AggFinal::MinF64: case AggFinal::MaxF64: {
    out->dtype = LLKV_DT_FLOAT64;
    if (rows == 0) { out->is_null = 1; return LLKV_OK; }
    if (a.plain_minmax) { out->f64 = key_to_f64((int64_t)l[0]); return LLKV_OK; } // (no NaN, no −0.0 in the column: the order key is the answer)
    if (l[2] & 1u) { out->f64 = std::nan(""); return LLKV_OK; } // a leading NaN sticks (:1319-1330)
    const uint64_t none = a.fin == AggFinal::MinF64 ? 0x7FFFFFFFFFFFFFFFull : 0x8000000000000000ull;
    if (l[0] == none) { out->f64 = std::nan(""); return LLKV_OK; } // unreachable: first row is not NaN
    double v = key_to_f64((int64_t)l[0]);
    if (v == 0.0 && l[1] != 0x7FFFFFFFFFFFFFFFull && (l[1] & 1u)) v = -0.0; // ±0 ties keep the earlier row
    out->f64 = v;
    return LLKV_OK;
  }
  }
  return LLKV_INTERNAL;
}

// Exact replay of the reference's sequential checked_add chain (llkv-aggregate/src/lib.rs:801-830) for one
// aggregate: selected argument values in row order → 128-bit inclusive scan → any prefix outside i64?
// Argument values of the rows a predicate selects, in row order (EmitPlan: count / scan / write).
int Query::emit_values(const LoweredPlan &ep, Scratch *vals, uint64_t *n_out) {
  hipStream_t s = g_ctx.stream;
  *n_out = 0;
  if (ep.always_false || table->local_rows == 0) return LLKV_OK;
  JitKernel k;
  std::string err;
  int rc = jit_compile(JitKind::Emit, ep.type_string, &k, &err);
  if (rc) return set_error(rc, err);
  const TileSet *ts = nullptr;
  if ((rc = get_tileset(*table, 8192, &ts))) return rc;
  const uint32_t n_slots = ts->n_tiles * (kBlock / 64);
  Scratch counts, offsets;
  if ((rc = counts.alloc((size_t)n_slots * 8)) || (rc = offsets.alloc((size_t)(n_slots + 1) * 8))) return rc;
  ScanParams sp;
  std::memset(&sp, 0, sizeof sp);
  for (size_t i = 0; i < ep.slot_fields.size(); ++i) sp.col[i] = slot_buffer(table->cols, ep, i);
  for (size_t i = 0; i < ep.lit_i.size(); ++i) sp.lit_i[i] = ep.lit_i[i];
  for (size_t i = 0; i < ep.lit_f.size(); ++i) sp.lit_f[i] = ep.lit_f[i];
  sp.tiles = ts->d_tiles;
  sp.n_tiles = ts->n_tiles;
  sp.sub_rows = 8192 / (kBlock / 64);
  sp.tile_partials = counts.as<uint64_t>();
  if ((rc = jit_launch_raw(k.fn, ts->n_tiles, &sp, sizeof sp, s))) return rc;
  HIP_TRY(launch_exclusive_scan(counts.as<uint64_t>(), offsets.as<uint64_t>(), n_slots, s));
  uint64_t n = 0;
  HIP_TRY(hipMemcpyAsync(&n, offsets.as<uint64_t>() + n_slots, 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (n >= kPredErrorBit) return set_error(LLKV_INTERNAL, "Arithmetic overflow: Overflow happened in a comparison");
  if (n == 0) return LLKV_OK;
  if ((rc = vals->alloc(n * 8))) return rc;
  sp.aux_in = offsets.as<uint64_t>();
  sp.aux_out = vals->as<uint64_t>();
  if ((rc = jit_launch_raw(k.fn2, ts->n_tiles, &sp, sizeof sp, s))) return rc;
  HIP_TRY(hipStreamSynchronize(s)); // counts / offsets are released on return
  *n_out = n;
  return LLKV_OK;
}

// Exact replay of the reference's sequential checked_add chain (llkv-aggregate/src/lib.rs:801-830) for one
// aggregate: selected argument values in row order → 128-bit inclusive scan → any prefix outside i64?
static int prefix_overflow_of(const int64_t *d_vals, uint64_t n, bool *overflow, __int128 *total) {
  hipStream_t s = g_ctx.stream;
  *overflow = false;
  if (total) *total = 0;
  if (n == 0) return LLKV_OK;
  Scratch prefix, flag, tmp;
  int rc;
  if ((rc = prefix.alloc(n * 16)) || (rc = flag.alloc(4))) return rc;
  HIP_TRY(hipMemsetAsync(flag.p, 0, 4, s));
  size_t tb = 0;
  HIP_TRY(hj_prefix_overflow(nullptr, &tb, d_vals, n, prefix.p, flag.as<uint32_t>(), s));
  if ((rc = tmp.alloc(tb ? tb : 8))) return rc;
  HIP_TRY(hj_prefix_overflow(tmp.p, &tb, d_vals, n, prefix.p, flag.as<uint32_t>(), s));
  uint32_t f = 0;
  uint64_t last[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(&f, flag.p, 4, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(last, static_cast<const char *>(prefix.p) + (n - 1) * 16, 16, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  *overflow = f != 0;
  if (total) *total = (__int128)(((unsigned __int128)last[1] << 64) | last[0]);
  return LLKV_OK;
}

int Query::exact_prefix_overflow(size_t agg, bool *overflow) {
  if (table->world != 1) return set_error(LLKV_UNSUPPORTED, "possible intermediate i64 overflow in SUM: the order-dependent check needs the whole table on one rank");
  Scratch vals;
  uint64_t n = 0;
  int rc = emit_values(exact_plans[agg], &vals, &n);
  if (rc) return rc;
  return prefix_overflow_of(vals.as<int64_t>(), n, overflow, nullptr);
}

// DISTINCT aggregates (llkv-aggregate/src/lib.rs: CountDistinctColumn :787-799, SumDistinct* :831-867,889-924,
// TotalDistinct* :990-1066, AvgDistinct* :1145-1232; keys: Int by value, Float by bit pattern :252-331): the
// accumulator adds a value the first time it sees it, so the sum runs over the distinct values in order of
// FIRST APPEARANCE.  Selected values in row order → stable sort by value → run heads (= first appearances) →
// back into row order → exact i64 prefix scan / ordered f64 sum.
// The distinct values of this rank's selected rows, in order of first appearance (device memory).
int Query::distinct_set(size_t agg, Scratch *dv_out, uint64_t *m_out) {
  const DistinctAgg &da = distinct[agg];
  hipStream_t s = g_ctx.stream;
  Scratch vals;
  uint64_t n = 0;
  int rc = emit_values(da.plan, &vals, &n);
  if (rc) return rc;
  if (n >= (1ull << 32)) return set_error(LLKV_UNSUPPORTED, "more than 2^32 selected rows in a DISTINCT aggregate");
  uint64_t m = 0; // distinct values
  Scratch &dv = *dv_out;
  if (n) {
    Scratch pos, vals_s, pos_s, flags, offs, tmp, hv, hp, hp_s;
    if ((rc = pos.alloc(n * 4)) || (rc = vals_s.alloc(n * 8)) || (rc = pos_s.alloc(n * 4)) || (rc = flags.alloc((n + 1) * 8)) || (rc = offs.alloc((n + 1) * 8))) return rc;
    HIP_TRY(hj_launch_iota(pos.as<uint32_t>(), (uint32_t)n, s));
    // only equality matters: sort (value − min) over the bits the value range leaves (integers of a narrow range:
    // 3 radix passes instead of 8); the minimum is added back to the distinct values afterwards
    uint64_t vmin = 0;
    uint32_t sort_bits = 64;
    {
      Scratch mm;
      int64_t init[2] = {INT64_MAX, INT64_MIN}, got[2];
      if ((rc = mm.alloc(16))) return rc;
      HIP_TRY(hipMemcpyAsync(mm.p, init, 16, hipMemcpyHostToDevice, s));
      HIP_TRY(launch_minmax_i64(vals.as<int64_t>(), n, mm.as<int64_t>(), s));
      Readback rb;
      if ((rc = rb.add(got, mm.p, 16, s)) || (rc = rb.wait())) return rc; // (also: `init` is on the stack)
      const uint64_t span = (uint64_t)got[1] - (uint64_t)got[0];
      uint32_t b = 1;
      while (b < 64 && (span >> b) != 0) ++b;
      if (b <= 56) { // at least one 8-bit radix pass fewer
        vmin = (uint64_t)got[0];
        sort_bits = b;
        HIP_TRY(hj_launch_add_u64(vals.as<uint64_t>(), n, (uint64_t)0 - vmin, s));
      }
    }
    size_t tb = 0;
    HIP_TRY(hj_sort_u64_u32_bits(nullptr, &tb, vals.as<uint64_t>(), vals_s.as<uint64_t>(), pos.as<uint32_t>(), pos_s.as<uint32_t>(), n, sort_bits, s));
    if ((rc = tmp.alloc(tb ? tb : 8))) return rc;
    HIP_TRY(hj_sort_u64_u32_bits(tmp.p, &tb, vals.as<uint64_t>(), vals_s.as<uint64_t>(), pos.as<uint32_t>(), pos_s.as<uint32_t>(), n, sort_bits, s));
    HIP_TRY(hipMemsetAsync(flags.p, 0, (n + 1) * 8, s));
    HIP_TRY(hj_launch_run_heads(vals_s.as<uint64_t>(), n, flags.as<uint64_t>(), s));
    HIP_TRY(hipStreamSynchronize(s));
    tb = 0;
    HIP_TRY(hj_exclusive_scan_u64(nullptr, &tb, flags.as<uint64_t>(), offs.as<uint64_t>(), n + 1, s));
    if ((rc = tmp.alloc(tb ? tb : 8))) return rc;
    HIP_TRY(hj_exclusive_scan_u64(tmp.p, &tb, flags.as<uint64_t>(), offs.as<uint64_t>(), n + 1, s));
    HIP_TRY(hipMemcpyAsync(&m, offs.as<uint64_t>() + n, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if ((rc = hv.alloc(m * 8)) || (rc = hp.alloc(m * 4)) || (rc = hp_s.alloc(m * 4)) || (rc = dv.alloc(m * 8))) return rc;
    // heads: (position of the first appearance, value); the stable sort put the smallest position first in its run
    HIP_TRY(hj_launch_compact_pairs(pos_s.as<uint32_t>(), vals_s.as<uint64_t>(), flags.as<uint64_t>(), offs.as<uint64_t>(), n, hp.as<uint32_t>(), hv.as<uint64_t>(), s));
    uint32_t bits = 1;
    while (bits < 32 && (n >> bits) != 0) ++bits;
    tb = 0;
    HIP_TRY(hj_sort_u32_u64(nullptr, &tb, hp.as<uint32_t>(), hp_s.as<uint32_t>(), hv.as<uint64_t>(), dv.as<uint64_t>(), m, bits, s));
    HIP_TRY(hipStreamSynchronize(s));
    if ((rc = tmp.alloc(tb ? tb : 8))) return rc;
    HIP_TRY(hj_sort_u32_u64(tmp.p, &tb, hp.as<uint32_t>(), hp_s.as<uint32_t>(), hv.as<uint64_t>(), dv.as<uint64_t>(), m, bits, s));
    if (vmin) HIP_TRY(hj_launch_add_u64(dv.as<uint64_t>(), m, vmin, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  *m_out = m;
  return LLKV_OK;
}

int Query::distinct_value(size_t agg, llkv_value *out) {
  const DistinctAgg &da = distinct[agg];
  std::memset(out, 0, sizeof *out);
  hipStream_t s = g_ctx.stream;
  Scratch dv;
  uint64_t m = 0;
  int rc = distinct_set(agg, &dv, &m);
  if (rc) return rc;
  auto f64_sum = [&](int as_int, double *sum) -> int {
    *sum = 0.0;
    if (m == 0) return LLKV_OK;
    Scratch d, partials;
    int r = d.alloc(8);
    if (r || (r = partials.alloc((m / 65536 + 1) * 8))) return r;
    HIP_TRY(hj_launch_sum_f64_ordered(dv.as<uint64_t>(), m, as_int, d.as<double>(), partials.as<double>(), s));
    HIP_TRY(hipMemcpyAsync(sum, d.p, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return LLKV_OK;
  };
  if (da.kind == LLKV_AGG_COUNT) { out->dtype = LLKV_DT_INT64; out->i64 = (int64_t)m; return LLKV_OK; }
  if (da.key_dtype == LLKV_DT_UTF8 || da.key_dtype == LLKV_DT_BOOLEAN || da.key_dtype == LLKV_DT_DATE32) {
    // Sum / Total / AvgDistinctFloat64 over a non-float column (:889-924,1035-1066,1200-1232): each NEW key adds its numeric
    // image — Str through array_value_to_numeric, Bool 1 / 0, Date its day number — in order of first appearance.  The
    // keys are few (a dictionary, two booleans, the days of a table): the chain runs on the host.
    std::vector<uint64_t> keys(m);
    if (m) {
      HIP_TRY(hipMemcpyAsync(keys.data(), dv.p, m * 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
    }
    double sum = 0.0;
    for (uint64_t k : keys) {
      if (da.key_dtype == LLKV_DT_UTF8) { if (k >= da.key_numeric.size()) return set_error(LLKV_INTERNAL, "dictionary code beyond the dictionary"); sum += da.key_numeric[k]; }
      else if (da.key_dtype == LLKV_DT_BOOLEAN) sum += k ? 1.0 : 0.0;
      else sum += (double)(int32_t)(int64_t)k;
    }
    out->dtype = LLKV_DT_FLOAT64;
    if (da.kind == LLKV_AGG_TOTAL) { out->f64 = sum; return LLKV_OK; }
    if (m == 0) { out->is_null = 1; return LLKV_OK; }
    out->f64 = da.kind == LLKV_AGG_AVG ? sum / (double)m : sum;
    return LLKV_OK;
  }
  if (da.key_dtype == LLKV_DT_DECIMAL128) {
    // Sum / Total / AvgDistinctDecimal128 (:943-967,1089-1112,1260-1284; finalize :1583-1612,1656-1672,1762-1800): an i128
    // sum of the distinct raw values (64-bit images, < 2^32 of them: no i128 overflow), AVG half away from zero
    out->dtype = LLKV_DT_DECIMAL128;
    out->precision = da.precision; out->scale = da.scale;
    if (m == 0) { out->is_null = da.kind != LLKV_AGG_TOTAL; return LLKV_OK; }
    bool overflow = false;
    __int128 total = 0;
    if ((rc = prefix_overflow_of(dv.as<int64_t>(), m, &overflow, &total))) return rc;
    if (da.kind == LLKV_AGG_AVG) {
      const __int128 cnt = (__int128)m;
      __int128 avg = total / cnt;
      const __int128 rem = total % cnt;
      if ((rem < 0 ? -rem : rem) * 2 >= cnt) avg += total >= 0 ? 1 : -1;
      total = avg;
    }
    out->i64 = (int64_t)(uint64_t)(unsigned __int128)total;
    out->i64_hi = (int64_t)(total >> 64);
    return LLKV_OK;
  }
  switch (da.kind) {
  case LLKV_AGG_TOTAL: {
    out->dtype = LLKV_DT_FLOAT64;
    return f64_sum(da.is_f64 ? 0 : 1, &out->f64);
  }
  case LLKV_AGG_SUM: case LLKV_AGG_AVG: {
    const bool avg = da.kind == LLKV_AGG_AVG;
    out->dtype = (avg || da.is_f64) ? LLKV_DT_FLOAT64 : LLKV_DT_INT64;
    if (m == 0) { out->is_null = 1; return LLKV_OK; }
    if (da.is_f64) {
      double sum;
      if ((rc = f64_sum(0, &sum))) return rc;
      out->f64 = avg ? sum / (double)m : sum;
      return LLKV_OK;
    }
    bool overflow = false;
    __int128 total = 0;
    if ((rc = prefix_overflow_of(dv.as<int64_t>(), m, &overflow, &total))) return rc;
    if (overflow) return set_error(LLKV_INVALID_ARGUMENT, avg ? "AVG(DISTINCT) aggregate sum exceeds i64 range" : "integer overflow");
    if (avg) out->f64 = (double)(int64_t)total / (double)m;
    else out->i64 = (int64_t)total;
    return LLKV_OK;
  }
  default: return set_error(LLKV_INTERNAL, "not a DISTINCT aggregate");
  }
}

// Sharded tables: this rank's distinct values go to the host; the binding all-gathers them.
int Query::distinct_partial(size_t agg, const uint64_t **values, uint64_t *n) {
  if (agg >= distinct.size() || distinct[agg].kind < 0) return set_error(LLKV_INVALID_ARGUMENT, "not a DISTINCT aggregate");
  if (distinct[agg].key_dtype != LLKV_DT_INT64 && distinct[agg].key_dtype != LLKV_DT_FLOAT64)
    return set_error(LLKV_UNSUPPORTED, std::string("DISTINCT aggregate over a ") + dtype_name(distinct[agg].key_dtype) + " column of a sharded table");
  Scratch dv;
  uint64_t m = 0;
  int rc = distinct_set(agg, &dv, &m);
  if (rc) return rc;
  if (distinct_host.size() < distinct.size()) distinct_host.resize(distinct.size());
  std::vector<uint64_t> &h = distinct_host[agg];
  h.resize(m);
  if (m && (rc = fetch_to_host(h.data(), dv.p, m * 8))) return rc;
  *values = h.data();
  *n = m;
  return LLKV_OK;
}

// The ranks' distinct values in rank order = the table's order of first appearance; the aggregate over their
// union follows the accumulator's own rules (llkv-aggregate/src/lib.rs: a value is added the first time it is
// seen; i64 sums checked, f64 sums 0.0 then += in that order).
int Query::merge_distinct(size_t agg, uint32_t world, const uint64_t *counts, const uint64_t *const *values) {
  if (agg >= distinct.size() || distinct[agg].kind < 0) return set_error(LLKV_INVALID_ARGUMENT, "not a DISTINCT aggregate");
  if (groups.empty() || agg >= groups.n_values) return set_error(LLKV_INVALID_ARGUMENT, "finish the query before merging");
  const DistinctAgg &da = distinct[agg];
  std::unordered_set<uint64_t> seen;
  std::vector<uint64_t> all;
  for (uint32_t r = 0; r < world; ++r) {
    if (counts[r] && !values[r]) return set_error(LLKV_INVALID_ARGUMENT, "distinct values of a rank are missing");
    for (uint64_t i = 0; i < counts[r]; ++i)
      if (seen.insert(values[r][i]).second) all.push_back(values[r][i]);
  }
  const uint64_t m = all.size();
  llkv_value out;
  std::memset(&out, 0, sizeof out);
  auto as_f64 = [&](uint64_t bits) { double d; if (da.is_f64) std::memcpy(&d, &bits, 8); else d = (double)(int64_t)bits; return d; };
  switch (da.kind) {
  case LLKV_AGG_COUNT: out.dtype = LLKV_DT_INT64; out.i64 = (int64_t)m; break;
  case LLKV_AGG_TOTAL: {
    out.dtype = LLKV_DT_FLOAT64;
    double acc = 0.0;
    for (uint64_t v : all) acc += as_f64(v);
    out.f64 = acc;
    break;
  }
  case LLKV_AGG_SUM: case LLKV_AGG_AVG: {
    const bool avg = da.kind == LLKV_AGG_AVG;
    out.dtype = (avg || da.is_f64) ? LLKV_DT_FLOAT64 : LLKV_DT_INT64;
    if (m == 0) { out.is_null = 1; break; }
    if (da.is_f64) {
      double acc = 0.0;
      for (uint64_t v : all) acc += as_f64(v);
      out.f64 = avg ? acc / (double)m : acc;
      break;
    }
    int64_t acc = 0;
    for (uint64_t v : all)
      if (__builtin_add_overflow(acc, (int64_t)v, &acc))
        return set_error(LLKV_INVALID_ARGUMENT, avg ? "AVG(DISTINCT) aggregate sum exceeds i64 range" : "integer overflow");
    if (avg) out.f64 = (double)acc / (double)m;
    else out.i64 = acc;
    break;
  }
  default: return set_error(LLKV_INTERNAL, "not a DISTINCT aggregate");
  }
  groups.value(0, agg) = out;
  return LLKV_OK;
}

int Query::finish_from_exchange(const uint64_t *exchange) {
  const LoweredPlan &p = plan;
  const bool trace = std::getenv("LLKV_HIP_TRACE") != nullptr; // phase times of the host-side finalize on stderr
  const auto t0 = std::chrono::steady_clock::now();
  auto mark = [&](const char *what) {
    if (trace) std::fprintf(stderr, "[llkv_hip] finalize %-10s %8.1f us\n", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  };
  std::vector<uint64_t> state(p.lanes);
  if (p.acc_image && table->world == 1) // the image fold left everything in the first octant (the others hold the lane identities)
    std::memcpy(state.data(), exchange + (size_t)__builtin_ctz(table->owned_mask | 0x100u) * p.lanes, (size_t)p.lanes * 8);
  else fold_exchange_host(exchange, p.lane_ops.data(), (uint32_t)p.lanes, state.data());
  mark("fold");
  groups.reset(0, 0, 0);
  if (state[(size_t)p.ng * p.k] != 0) // checked arithmetic failed on a selected row
    return set_error(LLKV_INTERNAL, arith_error_message(state[(size_t)p.ng * p.k]));
  const int base = p.track_first ? 2 : 1;
  // The groups that appeared, in output order, BEFORE any per-group object exists: ORDER BY keys ASC NULLS FIRST = the order
  // of a mixed-radix number whose digits are the ranks of the key codes (an integer key's code is its rank, a dictionary
  // code ranks by its string, the NULL code ranks first); else first-appearance order (llkv-executor/src/lib.rs:5065-5089)
  // by the first-row lane; else dense id order.  (Sorting 2 526 finished group objects by their key vectors cost as much
  // as the kernel that produced them.)
  struct Ref { uint64_t order; uint32_t g; };
  std::vector<Ref> present;
  present.reserve(p.ng);
  std::vector<std::vector<uint32_t>> rank(p.key_fields.size());
  if (p.grouped && order_by_keys)
    for (size_t k = 0; k < p.key_fields.size(); ++k) {
      const uint32_t card = p.key_cards[k], n_codes = p.key_nullable[k] ? card - 1 : card;
      rank[k].assign(card, 0);
      std::vector<uint32_t> by_value(n_codes);
      for (uint32_t c = 0; c < n_codes; ++c) by_value[c] = c;
      if (!p.key_is_int[k]) {
        const auto &dict = table->cols.at(p.key_fields[k]).info.dictionary;
        auto word = [&](uint32_t c) -> const std::string & { static const std::string none; return c < dict.size() ? dict[c] : none; };
        std::stable_sort(by_value.begin(), by_value.end(), [&](uint32_t x, uint32_t y) { return word(x) < word(y); });
      }
      for (uint32_t r = 0; r < n_codes; ++r) rank[k][by_value[r]] = r + 1; // 0 is the NULL group's
    }
  for (uint32_t g = 0; g < p.ng; ++g) {
    const uint64_t *gl = &state[(size_t)g * p.k];
    if (p.grouped && gl[0] == 0) continue; // group never appeared
    uint64_t order = g;
    if (p.grouped && order_by_keys) {
      order = 0;
      for (size_t k = 0; k < p.key_fields.size(); ++k) order = order * (p.key_cards[k] + 1) + rank[k][(g / p.key_strides[k]) % p.key_cards[k]];
    } else if (p.track_first) {
      order = gl[1];
    }
    present.push_back({order, g});
  }
  auto before = [](const Ref &x, const Ref &y) { return x.order < y.order; };
  if (p.grouped && !std::is_sorted(present.begin(), present.end(), before)) std::stable_sort(present.begin(), present.end(), before); // (integer keys: already in order)
  mark("order");
  groups.reset(present.size(), p.grouped ? p.key_fields.size() : 0, p.aggs.size());
  std::vector<const std::vector<std::string> *> dicts(p.key_fields.size(), nullptr);
  for (size_t k = 0; p.grouped && k < p.key_fields.size(); ++k)
    if (!p.key_is_int[k]) dicts[k] = &table->cols.at(p.key_fields[k]).info.dictionary;
  std::string err;
  for (const Ref &ref : present) {
    const uint32_t g = ref.g;
    const uint64_t *gl = &state[(size_t)g * p.k];
    const size_t at = groups.n++;
    if (p.grouped)
      for (size_t k = 0; k < p.key_fields.size(); ++k) {
        const uint32_t code = (g / p.key_strides[k]) % p.key_cards[k];
        groups.keys.emplace_back();
        GroupKey &gk = groups.keys.back();
        if (p.key_nullable[k] && code == p.key_cards[k] - 1) {
          gk.is_null = true;
          gk.is_int = p.key_is_int[k] != 0;
        } else if (p.key_is_int[k]) {
          gk.is_int = true;
          gk.i = p.key_bases[k] + (int64_t)code;
        } else if (code < dicts[k]->size()) {
          gk.s = (*dicts[k])[code];
        }
      }
    for (size_t a = 0; a < p.aggs.size(); ++a) {
      llkv_value *out = &groups.value(at, a);
      int rc = finalize_value(p.aggs[a], gl, base, out, &err, false);
      if (rc == LLKV_UNSUPPORTED && !p.grouped && a < exact_plans.size() && !exact_plans[a].type_string.empty()) {
        // the order-free state cannot tell whether a PREFIX of the reference's checked_add chain overflows:
        // decide it exactly from the selected values in row order
        bool overflow = false;
        if ((rc = exact_prefix_overflow(a, &overflow))) return rc;
        if (overflow) return set_error(LLKV_INVALID_ARGUMENT, p.aggs[a].fin == AggFinal::AvgI64 ? "AVG aggregate sum exceeds i64 range" : "integer overflow");
        rc = finalize_value(p.aggs[a], gl, base, out, &err, true);
      }
      if (rc) return set_error(rc, err);
    }
    if (!p.grouped)
      for (size_t a = 0; a < distinct.size() && a < p.aggs.size(); ++a)
        if (distinct[a].kind >= 0 && table->world == 1) { int rc = distinct_value(a, &groups.value(at, a)); if (rc) return rc; } // sharded: merge_distinct
  }
  mark("groups");
  return LLKV_OK;
}

// Make `stream` wait until the exchange image of the OLDEST not-yet-submitted execution is complete (for a
// caller that runs the all-reduce on a communication stream before submit).
int Query::wait_folded(hipStream_t stream) {
  if (sorted) return LLKV_OK;
  if (n_submitted >= n_launched) return set_error(LLKV_INVALID_ARGUMENT, "no launched execution awaits submission");
  if (!stream) stream = g_ctx.stream;
  const uint32_t slot = (uint32_t)(n_submitted % depth);
  int rc;
  if (pending && pending_slot == slot && (rc = flush_pending())) return rc;
  if (stream != slot_stream[slot]) HIP_TRY(hipStreamWaitEvent(stream, ev_fold[slot], 0));
  return LLKV_OK;
}

// Enqueue the copy-out of the oldest launched-but-not-submitted execution (after the caller's
// collective, if any, on `stream`).  Single-rank images already live in host memory.
int Query::submit(hipStream_t stream) {
  if (n_submitted >= n_launched) return set_error(LLKV_INVALID_ARGUMENT, "submit without a launched execution");
  if (sorted) { n_submitted++; return LLKV_OK; }
  const uint32_t slot = (uint32_t)(n_submitted % depth);
  if (!host_mapped) {
    int rc;
    if (pending && pending_slot == slot && (rc = flush_pending())) return rc;
    if (!stream) stream = slot_stream[slot];
    else if (stream != slot_stream[slot]) HIP_TRY(hipStreamWaitEvent(stream, ev_fold[slot], 0));
    size_t bytes = exchange_len() * sizeof(uint64_t), first = 0;
    if (plan.acc_image && table->world == 1) { // only the first octant carries data (finish_from_exchange)
      first = (size_t)__builtin_ctz(table->owned_mask | 0x100u) * plan.lanes;
      bytes = (size_t)plan.lanes * sizeof(uint64_t);
    }
    HIP_TRY(hipMemcpyAsync(h_exchange + slot * exchange_len() + first, d_exchange + slot * exchange_len() + first, bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipEventRecord(copied[slot], stream));
  }
  n_submitted++;
  return LLKV_OK;
}

// Wait for the oldest submitted execution, fold and finalize it.
int Query::collect() {
  if (n_collected >= n_submitted) return set_error(LLKV_INVALID_ARGUMENT, "collect without a submitted execution");
  if (sorted) { n_collected++; return LLKV_OK; }
  const uint32_t slot = (uint32_t)(n_collected % depth);
  int rc;
  if (pending && pending_slot == slot && (rc = flush_pending())) return rc; // nothing was launched behind it
  HIP_TRY(hipEventSynchronize(host_mapped ? ev_fold[slot] : copied[slot]));
  n_collected++;
  return finish_from_exchange(h_exchange + slot * exchange_len());
}

int Query::finish(hipStream_t stream) {
  while (n_submitted < n_launched) {
    int rc = submit(stream);
    if (rc) return rc;
  }
  int rc = LLKV_OK;
  while (n_collected < n_submitted && rc == LLKV_OK) rc = collect();
  return rc;
}

} // namespace llkv

// =====================================================================================
// C ABI
// =====================================================================================
using namespace llkv;

extern "C" {

const char *llkv_hip_last_error(void) { return g_last_error.c_str(); }
uint32_t llkv_hip_abi_version(void) { return LLKV_HIP_ABI_VERSION; }

int32_t llkv_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

llkv_status llkv_hip_init(int32_t device_ordinal) {
  std::lock_guard<std::mutex> lk(g_ctx.mu);
  if (g_ctx.ready) {
    if (g_ctx.device == device_ordinal) return LLKV_OK;
    return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "already bound to another device (one process per GPU)");
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return (llkv_status)set_error(LLKV_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
  if (device_ordinal < 0 || device_ordinal >= n) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "device ordinal out of range");
  if ((e = hipSetDevice(device_ordinal)) != hipSuccess) return (llkv_status)set_error(LLKV_NO_DEVICE, hipGetErrorString(e));
  if ((e = hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking)) != hipSuccess) return (llkv_status)set_error(LLKV_INTERNAL, hipGetErrorString(e));
  g_ctx.device = device_ordinal;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) == hipSuccess && prop.multiProcessorCount > 0) g_ctx.cu_count = (uint32_t)prop.multiProcessorCount;
  g_ctx.ready = true;
  if (!std::getenv("LLKV_HIP_NO_STAGING_PRIME")) staging_prime();
  return LLKV_OK;
}

void llkv_hip_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_ctx.mu);
  if (!g_ctx.ready) return;
  jit_shutdown();
  scratch_release_all();
  pinned_release_all();
  staging_release();
  (void)hipStreamDestroy(g_ctx.stream);
  g_ctx.stream = nullptr;
  g_ctx.ready = false;
}

void llkv_hip_free(void *ptr) {
  if (!result_release(ptr)) std::free(ptr);
}
// ---- queries ------------------------------------------------------------------------
llkv_status llkv_hip_query_prepare_aggregate(const llkv_hip_table *table, const llkv_filter *filters, uint32_t n_filters,
                                             const llkv_eval_op *ops, uint32_t n_ops, const llkv_aggregate_spec *aggs,
                                             uint32_t n_aggs, llkv_hip_query **out) {
  if (!out) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "out is NULL");
  Query *q = nullptr;
  int rc = prepare_query(reinterpret_cast<const Table *>(table), filters, n_filters, ops, n_ops, nullptr, 0, aggs, n_aggs, false, false, &q);
  if (rc) return (llkv_status)rc;
  *out = reinterpret_cast<llkv_hip_query *>(q);
  return LLKV_OK;
}

llkv_status llkv_hip_query_prepare_groupby(const llkv_hip_table *table, const llkv_filter *filters, uint32_t n_filters,
                                           const llkv_eval_op *ops, uint32_t n_ops, const uint32_t *key_fields,
                                           uint32_t n_keys, const llkv_aggregate_spec *aggs, uint32_t n_aggs,
                                           int32_t order_by_keys, llkv_hip_query **out) {
  if (!out) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "out is NULL");
  Query *q = nullptr;
  int rc = prepare_query(reinterpret_cast<const Table *>(table), filters, n_filters, ops, n_ops, key_fields, n_keys, aggs, n_aggs, true, order_by_keys != 0, &q);
  if (rc) return (llkv_status)rc;
  *out = reinterpret_cast<llkv_hip_query *>(q);
  return LLKV_OK;
}

void llkv_hip_query_free(llkv_hip_query *query) { delete reinterpret_cast<Query *>(query); }

llkv_status llkv_hip_query_launch(llkv_hip_query *query, void *hip_stream) {
  if (int rc_dev = ensure_device()) return (llkv_status)rc_dev;
  if (!query) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "query is NULL");
  return (llkv_status) reinterpret_cast<Query *>(query)->launch((hipStream_t)hip_stream);
}

llkv_status llkv_hip_query_exchange_buffer(llkv_hip_query *query, void **device_ptr, uint64_t *len_i64) {
  if (!query) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "query is NULL");
  Query *q = reinterpret_cast<Query *>(query);
  if (q->sorted) return (llkv_status)set_error(LLKV_UNSUPPORTED, "a sort-based GROUP BY has no exchange image (single rank only)");
  // image of the oldest execution awaiting submission (slot 0 before the first launch); consecutive
  // executions use consecutive slots of one ring: slot s lives at base + s * len
  const uint64_t cur = q->n_submitted < q->n_launched ? q->n_submitted : (q->n_launched ? q->n_launched - 1 : 0);
  if (device_ptr) *device_ptr = q->d_exchange + (cur % q->depth) * q->exchange_len();
  if (len_i64) *len_i64 = (uint64_t)kOctantsHost * (uint64_t)q->plan.lanes;
  return LLKV_OK;
}

llkv_status llkv_hip_query_finish(llkv_hip_query *query, void *hip_stream) {
  if (int rc_dev = ensure_device()) return (llkv_status)rc_dev;
  if (!query) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "query is NULL");
  return (llkv_status) reinterpret_cast<Query *>(query)->finish((hipStream_t)hip_stream);
}

llkv_status llkv_hip_query_read_exchange(llkv_hip_query *query, uint64_t *out, uint64_t len_i64) {
  if (int rc_dev = ensure_device()) return (llkv_status)rc_dev;
  Query *q = reinterpret_cast<Query *>(query);
  if (!q || !out) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  if (q->n_launched == 0) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "no execution launched");
  if (len_i64 != q->exchange_len()) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "exchange length mismatch");
  const uint32_t slot = (uint32_t)((q->n_launched - 1) % q->depth);
  if (q->pending && q->pending_slot == slot && q->flush_pending()) return LLKV_INTERNAL;
  if (hipEventSynchronize(q->ev_fold[slot]) != hipSuccess) return (llkv_status)set_error(LLKV_INTERNAL, "event wait failed");
  if (hipMemcpy(out, q->d_exchange + slot * q->exchange_len(), len_i64 * 8, hipMemcpyDefault) != hipSuccess)
    return (llkv_status)set_error(LLKV_INTERNAL, "exchange copy failed");
  return LLKV_OK;
}

llkv_status llkv_hip_query_finish_from_host(llkv_hip_query *query, const uint64_t *exchange, uint64_t len_i64) {
  if (int rc_dev = ensure_device()) return (llkv_status)rc_dev;
  if (!query || !exchange) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  Query *q = reinterpret_cast<Query *>(query);
  if (len_i64 != (uint64_t)kOctantsHost * (uint64_t)q->plan.lanes) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "exchange length mismatch");
  return (llkv_status)q->finish_from_exchange(exchange);
}

llkv_status llkv_hip_query_set_depth(llkv_hip_query *query, uint32_t depth) {
  Query *q = reinterpret_cast<Query *>(query);
  if (!q) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "query is NULL");
  if (depth == 0 || depth > Query::kMaxDepth) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "depth must be 1..8");
  if (q->n_launched != q->n_collected) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "executions in flight");
  q->depth = depth;
  q->n_launched = q->n_submitted = q->n_collected = 0;
  return LLKV_OK;
}

llkv_status llkv_hip_query_wait_folded(llkv_hip_query *query, void *hip_stream) {
  if (int rc_dev = ensure_device()) return (llkv_status)rc_dev;
  if (!query) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "query is NULL");
  return (llkv_status) reinterpret_cast<Query *>(query)->wait_folded((hipStream_t)hip_stream);
}

llkv_status llkv_hip_query_submit(llkv_hip_query *query, void *hip_stream) {
  if (int rc_dev = ensure_device()) return (llkv_status)rc_dev;
  if (!query) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "query is NULL");
  return (llkv_status) reinterpret_cast<Query *>(query)->submit((hipStream_t)hip_stream);
}

llkv_status llkv_hip_query_collect(llkv_hip_query *query) {
  if (int rc_dev = ensure_device()) return (llkv_status)rc_dev;
  if (!query) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "query is NULL");
  return (llkv_status) reinterpret_cast<Query *>(query)->collect();
}

uint32_t llkv_hip_query_num_groups(const llkv_hip_query *query) {
  if (!query) return 0;
  const Query *q = reinterpret_cast<const Query *>(query);
  return q->lazy.active ? (uint32_t)q->lazy.n : (uint32_t)q->groups.size();
}
uint32_t llkv_hip_query_num_keys(const llkv_hip_query *query) { return query ? (uint32_t) reinterpret_cast<const Query *>(query)->n_user_keys : 0; }
uint32_t llkv_hip_query_num_aggregates(const llkv_hip_query *query) { return query ? reinterpret_cast<const Query *>(query)->n_user_aggs : 0; }

llkv_status llkv_hip_query_group_key(const llkv_hip_query *query, uint32_t group, uint32_t key, llkv_value *out) {
  const Query *q = reinterpret_cast<const Query *>(query);
  if (q && out && q->lazy.active) { // sort-based route: decode the cell on request
    const LazyGroups &lz = q->lazy;
    if (group >= lz.n || key >= lz.n_keys) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "group/key index out of range");
    std::memset(out, 0, sizeof *out);
    const ColumnInfo *ci = lz.key_cols[key];
    const int64_t v = lz.key_vals[(size_t)key * lz.n + group];
    out->is_null = lz.key_valid[(size_t)key * lz.n + group] ? 0 : 1;
    if (ci->dtype == LLKV_DT_UTF8) {
      out->dtype = LLKV_DT_UTF8;
      out->str = (!out->is_null && (uint64_t)v < ci->dictionary.size()) ? ci->dictionary[(size_t)v].c_str() : "";
    } else {
      out->dtype = LLKV_DT_INT64;
      out->i64 = out->is_null ? 0 : v;
    }
    return LLKV_OK;
  }
  if (!q || !out || group >= q->groups.size() || key >= q->groups.n_keys)
    return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "group/key index out of range");
  std::memset(out, 0, sizeof *out);
  const GroupKey &gk = q->groups.key(group, key);
  if (gk.is_int) { out->dtype = LLKV_DT_INT64; out->i64 = gk.i; }
  else { out->dtype = LLKV_DT_UTF8; out->str = gk.s.c_str(); }
  out->is_null = gk.is_null ? 1 : 0;
  return LLKV_OK;
}

llkv_status llkv_hip_query_value(const llkv_hip_query *query, uint32_t group, uint32_t agg, llkv_value *out) {
  const Query *q = reinterpret_cast<const Query *>(query);
  if (q && out && q->lazy.active) {
    const LazyGroups &lz = q->lazy;
    if (group >= lz.n || agg >= lz.plan->aggs.size()) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "group/aggregate index out of range");
    std::string err;
    const int rc = finalize_value(lz.plan->aggs[agg], lz.lanes + (size_t)group * lz.k, 2, out, &err, false);
    return rc ? (llkv_status)set_error(rc, err) : LLKV_OK;
  }
  if (!q || !out || group >= q->groups.size() || agg >= q->groups.n_values)
    return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "group/aggregate index out of range");
  *out = q->groups.value(group, agg);
  return LLKV_OK;
}

llkv_status llkv_hip_query_partial_groups(const llkv_hip_query *query, uint64_t *n_groups, uint32_t *n_keys, uint32_t *lanes_per_group,
                                          const int64_t **key_values, const uint8_t **key_valid, const uint64_t **lanes) {
  const Query *q = reinterpret_cast<const Query *>(query);
  if (!q || !q->sorted || !q->lazy.active) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "not a finished sort-based GROUP BY");
  if (n_groups) *n_groups = q->lazy.n;
  if (n_keys) *n_keys = q->lazy.n_keys;
  if (lanes_per_group) *lanes_per_group = (uint32_t)q->lazy.k;
  if (key_values) *key_values = q->lazy.key_vals;
  if (key_valid) *key_valid = q->lazy.key_valid;
  if (lanes) *lanes = q->lazy.lanes;
  return LLKV_OK;
}

llkv_status llkv_hip_query_merge_groups(llkv_hip_query *query, uint32_t world, const uint64_t *rank_groups, const int64_t *const *key_values,
                                        const uint8_t *const *key_valid, const uint64_t *const *lanes) {
  Query *q = reinterpret_cast<Query *>(query);
  if (!q || !q->sorted || !q->lazy.active) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "not a finished sort-based GROUP BY");
  if (world == 0 || !rank_groups || !key_values || !key_valid || !lanes) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  return (llkv_status)sorted_groupby_merge(q->sorted, world, rank_groups, key_values, key_valid, lanes, &q->lazy);
}

llkv_status llkv_hip_query_distinct_partial(llkv_hip_query *query, uint32_t agg, const uint64_t **values, uint64_t *n_values) {
  Query *q = reinterpret_cast<Query *>(query);
  if (!q || !values || !n_values) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  if (int rc = ensure_device()) return (llkv_status)rc;
  return (llkv_status)q->distinct_partial(agg, values, n_values);
}

llkv_status llkv_hip_query_merge_distinct(llkv_hip_query *query, uint32_t agg, uint32_t world, const uint64_t *rank_counts,
                                          const uint64_t *const *rank_values) {
  Query *q = reinterpret_cast<Query *>(query);
  if (!q || world == 0 || !rank_counts || !rank_values) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  return (llkv_status)q->merge_distinct(agg, world, rank_counts, rank_values);
}

llkv_status llkv_hip_query_set_profiling(llkv_hip_query *query, int32_t enabled) {
  if (!query) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "query is NULL");
  Query *q = reinterpret_cast<Query *>(query);
  q->profiling = enabled != 0;
  q->profile_every = enabled > 1 ? (uint32_t)enabled : 1u;
  q->events_used = 0;
  return LLKV_OK;
}

llkv_status llkv_hip_query_kernel_time(llkv_hip_query *query, double *total_ms, uint64_t *launches, const char **kernel_name) {
  if (!query) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "query is NULL");
  Query *q = reinterpret_cast<Query *>(query);
  double sum = 0;
  for (size_t i = 0; i < q->events_used; ++i) {
    float ms = 0;
    if (hipEventSynchronize(q->events[i].second) != hipSuccess || hipEventElapsedTime(&ms, q->events[i].first, q->events[i].second) != hipSuccess)
      return (llkv_status)set_error(LLKV_INTERNAL, "event timing failed");
    sum += ms;
  }
  if (total_ms) *total_ms = sum;
  if (launches) *launches = q->events_used;
  if (kernel_name) *kernel_name = "fused_scan_kernel";
  q->events_used = 0;
  return LLKV_OK;
}

uint64_t llkv_hip_query_algorithmic_bytes(const llkv_hip_query *query) {
  const Query *q = reinterpret_cast<const Query *>(query);
  return q ? q->plan.bytes_per_row * q->table->local_rows : 0;
}

const char *llkv_hip_query_route_note(const llkv_hip_query *query) {
  const Query *q = reinterpret_cast<const Query *>(query);
  if (!q) return "";
  if (!q->route_note.empty()) return q->route_note.c_str();
  return q->plan.acc_lds ? "GROUP BY with per-thread accumulator columns in LDS" : q->plan.grouped ? "GROUP BY with register accumulators" : "ungrouped aggregates, register accumulators";
}

const char *llkv_hip_query_kernel_signature(const llkv_hip_query *query) {
  const Query *q = reinterpret_cast<const Query *>(query);
  return q ? q->plan.type_string.c_str() : "";
}

llkv_status llkv_hip_shard_layout(uint32_t n_chunks, uint32_t world, uint32_t *octant_chunk_begin, uint32_t *octant_owner) {
  if (world == 0 || world > (uint32_t)kOctantsHost) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "world must be 1..8");
  for (int j = 0; j <= kOctantsHost; ++j) if (octant_chunk_begin) octant_chunk_begin[j] = (uint32_t)((uint64_t)j * n_chunks / kOctantsHost);
  for (int o = 0; o < kOctantsHost; ++o) if (octant_owner) octant_owner[o] = (uint32_t)((uint64_t)o * world / kOctantsHost);
  return LLKV_OK;
}

llkv_status llkv_hip_query_lane_ops(const llkv_hip_query *query, uint8_t *ops_out, uint32_t *lanes) {
  const Query *q = reinterpret_cast<const Query *>(query);
  if (!q) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "query is NULL");
  if (lanes) *lanes = (uint32_t)q->plan.lanes;
  if (ops_out) std::memcpy(ops_out, q->plan.lane_ops.data(), q->plan.lane_ops.size());
  return LLKV_OK;
}

llkv_status llkv_hip_fold_exchange(const uint64_t *exchange, const uint8_t *lane_ops, uint32_t lanes, uint64_t *state_out) {
  if (!exchange || !lane_ops || !state_out) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  fold_exchange_host(exchange, lane_ops, lanes, state_out);
  return LLKV_OK;
}

llkv_status llkv_hip_aggregate(const llkv_hip_table *table, const llkv_filter *filters, uint32_t n_filters,
                               const llkv_eval_op *ops, uint32_t n_ops, const llkv_aggregate_spec *aggs, uint32_t n_aggs,
                               llkv_value *out_values) {
  llkv_hip_query *q = nullptr;
  llkv_status rc = llkv_hip_query_prepare_aggregate(table, filters, n_filters, ops, n_ops, aggs, n_aggs, &q);
  if (rc) return rc;
  rc = llkv_hip_query_launch(q, nullptr);
  if (!rc) rc = llkv_hip_query_finish(q, nullptr);
  if (!rc) for (uint32_t i = 0; i < n_aggs; ++i) llkv_hip_query_value(q, 0, i, &out_values[i]);
  llkv_hip_query_free(q);
  return rc;
}

} // extern "C"
