// table.cpp — HBM-resident table images behind include/llkv_hip.h: chunk → tile → octant layout, staging of
// columns (fixed width, Utf8 dictionary coding, Decimal128 narrowing, validity masks), column statistics.
#include "comm.hpp"
#include "engine.hpp"
#include "join.hpp"

#include <algorithm>
#include <cmath>
#include <atomic>
#include <chrono>
#include <cstring>
#include <map>
#include <thread>

namespace llkv {

static uint64_t round_up(uint64_t v, uint64_t m) { return (v + m - 1) / m * m; }

// ---------------------------------------------------------------------------------
// Table image
// ---------------------------------------------------------------------------------
Table::~Table() {
  for (auto &kv : cols) {
    if (kv.second.owned && kv.second.d_values) (void)hipFree(kv.second.d_values);
    if (kv.second.d_valid) (void)hipFree(kv.second.d_valid);
    if (kv.second.d_hi) (void)hipFree(kv.second.d_hi);
  }
  if (d_row_ids) (void)hipFree(d_row_ids);
  for (void *p : retired) (void)hipFree(p);
  for (auto &kv : key_images) if (kv.second.d) (void)hipFree(kv.second.d);
  for (auto &kv : tilesets) {
    if (kv.second.d_tiles) (void)hipFree(kv.second.d_tiles);
    if (kv.second.d_sample) (void)hipFree(kv.second.d_sample);
  }
}

// Canonical octant bounds over the global chunk list and the shard of this rank
// (DESIGN.md "Sharding").  Pure host logic: also exercised by the CPU tests.
void compute_layout(Table &t) {
  const uint32_t C = (uint32_t)t.global_chunk_rows.size();
  for (int j = 0; j <= kOctantsHost; ++j) t.octant_chunk_begin[j] = (uint32_t)((uint64_t)j * C / kOctantsHost);
  t.owned_mask = 0;
  for (int o = 0; o < kOctantsHost; ++o)
    if ((uint32_t)((uint64_t)o * t.world / kOctantsHost) == t.rank) t.owned_mask |= 1u << o;
  int first_o = -1, last_o = -1;
  for (int o = 0; o < kOctantsHost; ++o) if ((t.owned_mask >> o) & 1u) { if (first_o < 0) first_o = o; last_o = o; }
  t.first_chunk = first_o < 0 ? 0 : t.octant_chunk_begin[first_o];
  t.n_local_chunks = first_o < 0 ? 0 : t.octant_chunk_begin[last_o + 1] - t.first_chunk;
  t.total_rows = 0;
  t.local_logical_start = 0;
  for (uint32_t c = 0; c < C; ++c) {
    if (c == t.first_chunk) t.local_logical_start = t.total_rows;
    t.total_rows += t.global_chunk_rows[c];
  }
  t.local_rows = 0;
  t.chunk_dev_off.assign(t.n_local_chunks + 1, 0);
  for (uint32_t c = 0; c < t.n_local_chunks; ++c) {
    const uint64_t rows = t.global_chunk_rows[t.first_chunk + c];
    t.local_rows += rows;
    // every chunk starts on a 16-row boundary of the device image so that 16-byte loads of
    // 8-byte columns and 2-byte loads of code columns stay aligned for ragged chunks
    t.chunk_dev_off[c + 1] = round_up(t.chunk_dev_off[c] + rows, 16);
  }
  t.dev_rows = t.chunk_dev_off[t.n_local_chunks];
}

uint32_t octant_of_chunk(const Table &t, uint32_t global_chunk) {
  for (int o = 0; o < kOctantsHost; ++o)
    if (global_chunk >= t.octant_chunk_begin[o] && global_chunk < t.octant_chunk_begin[o + 1]) return (uint32_t)o;
  return kOctantsHost - 1;
}

void build_tiles_host(const Table &t, uint32_t tile_rows, std::vector<TileDesc> &tiles, uint32_t (&octant_tile_begin)[kOctantsHost + 1]) {
  tiles.clear();
  uint64_t logical = t.local_logical_start;
  std::vector<uint32_t> per_octant(kOctantsHost, 0);
  for (uint32_t c = 0; c < t.n_local_chunks; ++c) {
    const uint64_t rows = t.global_chunk_rows[t.first_chunk + c];
    const uint32_t o = octant_of_chunk(t, t.first_chunk + c);
    for (uint64_t r = 0; r < rows; r += tile_rows) {
      TileDesc d;
      d.dev_row = t.chunk_dev_off[c] + r;
      d.logical_row = logical + r;
      d.rows = (uint32_t)std::min<uint64_t>(tile_rows, rows - r);
      d.octant = o;
      tiles.push_back(d);
      per_octant[o]++;
    }
    logical += rows;
  }
  octant_tile_begin[0] = 0;
  for (int o = 0; o < kOctantsHost; ++o) octant_tile_begin[o + 1] = octant_tile_begin[o] + per_octant[o];
}

int get_tileset(const Table &tc, uint32_t tile_rows, const TileSet **out) {
  Table &t = const_cast<Table &>(tc);
  std::lock_guard<std::mutex> lk(t.mu);
  auto it = t.tilesets.find(tile_rows);
  if (it != t.tilesets.end()) { *out = &it->second; return LLKV_OK; }
  TileSet ts;
  std::vector<TileDesc> tiles;
  build_tiles_host(t, tile_rows, tiles, ts.octant_tile_begin);
  ts.n_tiles = (uint32_t)tiles.size();
  ts.tile_rows = tile_rows;
  if (ts.n_tiles) {
    HIP_TRY(hipMalloc((void **)&ts.d_tiles, tiles.size() * sizeof(TileDesc)));
    HIP_TRY(hipMemcpy(ts.d_tiles, tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
    std::vector<TileDesc> sample;
    for (size_t i = kTileSampleStride / 2; i < tiles.size(); i += kTileSampleStride) { sample.push_back(tiles[i]); ts.sample_rows += tiles[i].rows; }
    ts.n_sample = (uint32_t)sample.size();
    if (ts.n_sample) {
      HIP_TRY(hipMalloc((void **)&ts.d_sample, sample.size() * sizeof(TileDesc)));
      HIP_TRY(hipMemcpy(ts.d_sample, sample.data(), sample.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
    }
  }
  auto ins = t.tilesets.emplace(tile_rows, ts);
  *out = &ins.first->second;
  return LLKV_OK;
}

static const uint64_t kSlackRows = 8192; // readable rows past the image end (unrolled tail steps)

int get_key_image(const Table &tc, uint32_t field, uint64_t min_rows, const KeyImage **out) {
  *out = nullptr;
  Table &t = const_cast<Table &>(tc);
  std::lock_guard<std::mutex> lk(t.mu);
  if (t.local_rows < min_rows) return LLKV_OK;
  auto have = t.key_images.find(field);
  if (have != t.key_images.end()) { *out = &have->second; return LLKV_OK; }
  auto it = t.cols.find(field);
  if (it == t.cols.end()) return LLKV_OK;
  const DeviceColumn &c = it->second;
  if (c.info.dtype != LLKV_DT_INT64 || !c.info.has_stats || c.info.wide128 || !c.d_values) return LLKV_OK;
  if (c.info.min_i < INT32_MIN || c.info.max_i > INT32_MAX) return LLKV_OK;
  KeyImage img;
  img.info = c.info;
  img.info.dtype = LLKV_DT_INT32;
  // (padding rows between chunks and the slack behind the image hold what the column holds there — zeros or copies of real rows,
  // truncated like any other: no tile names them)
  const uint64_t rows = t.dev_rows + kSlackRows;
  HIP_TRY(hipMalloc(&img.d, rows * 4));
  HIP_TRY(launch_narrow_i64((const int64_t *)c.d_values, rows, (int32_t *)img.d, g_ctx.stream));
  auto ins = t.key_images.emplace(field, img);
  *out = &ins.first->second;
  return LLKV_OK;
}

uint64_t table_chunk_rows(const llkv_hip_table *table, uint32_t global_chunk) {
  const Table *t = reinterpret_cast<const Table *>(table);
  return t && global_chunk < t->global_chunk_rows.size() ? t->global_chunk_rows[global_chunk] : 0;
}


// `rows_overwritten`: every row of the image is about to be staged over (value buffers of a table without padding rows
// between its chunks): only the slack behind the image is zeroed — the runtime's fill moves ~130 GB/s, which made zeroing
// SF10's five 480 MB columns a fifth of their staging time
static int alloc_column(Table &t, uint32_t width, void **d_out, bool rows_overwritten = false) {
  const uint64_t bytes = (t.dev_rows + kSlackRows) * width;
  HIP_TRY(hipMalloc(d_out, bytes));
  if (rows_overwritten && t.dev_rows == t.local_rows) HIP_TRY(hipMemsetAsync((char *)*d_out + t.dev_rows * width, 0, kSlackRows * width, g_ctx.stream));
  else HIP_TRY(hipMemsetAsync(*d_out, 0, bytes, g_ctx.stream));
  return LLKV_OK;
}

// host-side preparation of a column image (dictionary coding, bitmap expansion, Decimal128 narrowing) runs chunk
// by chunk on a few threads; fn(chunk) returns a status, the first failure wins
template <class Fn> static int for_each_chunk_parallel(uint32_t n_chunks, Fn &&fn) {
  const unsigned hw = std::max(1u, std::min(16u, host_thread_limit()));
  const unsigned n_threads = std::min<unsigned>(hw, std::max(1u, n_chunks / 4));
  std::atomic<uint32_t> next{0};
  std::atomic<int> failed{LLKV_OK};
  std::string message;
  std::mutex message_mu;
  auto work = [&] {
    for (uint32_t i; failed == LLKV_OK && (i = next.fetch_add(1)) < n_chunks;) {
      const int rc = fn(i);
      if (rc) {
        std::lock_guard<std::mutex> g(message_mu);
        if (failed.exchange(rc) == LLKV_OK) message = g_last_error;
      }
    }
  };
  std::vector<std::thread> threads;
  for (unsigned k = 1; k < n_threads; ++k) threads.emplace_back(work);
  work();
  for (std::thread &t : threads) t.join();
  if (failed != LLKV_OK) return set_error(failed, message);
  return LLKV_OK;
}

// see fill_padding_kernel (catalog.hip)
static int fill_chunk_padding(Table &t, void *d_values, uint32_t width) {
  std::vector<uint64_t> pad;
  for (uint32_t i = 0; i < t.n_local_chunks; ++i) {
    const uint64_t rows = t.global_chunk_rows[t.first_chunk + i], end = t.chunk_dev_off[i] + rows;
    if (rows && end < t.chunk_dev_off[i + 1]) { pad.push_back(end); pad.push_back(t.chunk_dev_off[i + 1] - end); pad.push_back(end - 1); }
  }
  if (pad.empty()) return LLKV_OK;
  Scratch d;
  int rc = d.alloc(pad.size() * 8);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(d.p, pad.data(), pad.size() * 8, hipMemcpyHostToDevice, g_ctx.stream));
  HIP_TRY(launch_fill_padding(d_values, width, d.as<uint64_t>(), (uint32_t)(pad.size() / 3), g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream)); // `pad` is pageable and dies here
  return LLKV_OK;
}

static int column_stats_device(Table &t, DeviceColumn &c) {
  if (t.dev_rows != t.local_rows && c.d_values) {
    const int rc = fill_chunk_padding(t, c.d_values, dtype_width(c.info.dtype));
    if (rc) return rc;
  }
  if ((c.info.dtype == LLKV_DT_FLOAT64 || c.info.dtype == LLKV_DT_FLOAT32) && t.dev_rows) {
    // largest and smallest non-zero finite |v|: they bound aggregate arguments from above and below, which lets the
    // shared-image GROUP BY keep f64 sums exact (padding rows between ragged chunks hold copies of real values)
    uint64_t *d = nullptr, bits[3] = {0, 0x7FF0000000000000ull, 0};
    HIP_TRY(hipMalloc((void **)&d, 24));
    HIP_TRY(hipMemcpyAsync(d, bits, 24, hipMemcpyHostToDevice, g_ctx.stream));
    if (c.info.dtype == LLKV_DT_FLOAT64) HIP_TRY(launch_absrange_f64((const double *)c.d_values, t.dev_rows, d, g_ctx.stream));
    else HIP_TRY(launch_absrange_f32((const float *)c.d_values, t.dev_rows, d, g_ctx.stream));
    HIP_TRY(hipMemcpyAsync(bits, d, 24, hipMemcpyDeviceToHost, g_ctx.stream));
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    (void)hipFree(d);
    c.has_local_fstats = true;
    std::memcpy(&c.local_f_absmax, &bits[0], 8);
    std::memcpy(&c.local_f_absmin_nz, &bits[1], 8);
    if (!std::isfinite(c.local_f_absmin_nz)) c.local_f_absmin_nz = 0.0; // no non-zero value
    c.local_f_all_finite = (bits[2] & 1u) == 0;
    c.local_f_no_neg_zero = (bits[2] & 2u) == 0;
    if (t.world == 1) { c.info.has_fstats = true; c.info.f_absmax = c.local_f_absmax; c.info.f_absmin_nz = c.local_f_absmin_nz; c.info.f_all_finite = c.local_f_all_finite; c.info.f_no_neg_zero = c.local_f_no_neg_zero; }
    return LLKV_OK;
  }
  if (c.info.dtype != LLKV_DT_INT64 && c.info.dtype != LLKV_DT_INT32 && c.info.dtype != LLKV_DT_DATE32 && c.info.dtype != LLKV_DT_DECIMAL128) return LLKV_OK;
  if (t.dev_rows == 0) return LLKV_OK;
  int64_t init[3] = {INT64_MAX, INT64_MIN, 0}, *d = nullptr;
  HIP_TRY(hipMalloc((void **)&d, sizeof init));
  HIP_TRY(hipMemcpyAsync(d, init, sizeof init, hipMemcpyHostToDevice, g_ctx.stream));
  // (padding rows between ragged chunks hold a copy of a real value: fill_chunk_padding)
  if (c.info.dtype == LLKV_DT_INT64 || c.info.dtype == LLKV_DT_DECIMAL128) HIP_TRY(launch_minmax_i64((const int64_t *)c.d_values, t.dev_rows, d, g_ctx.stream));
  else HIP_TRY(launch_minmax_i32((const int32_t *)c.d_values, t.dev_rows, d, g_ctx.stream));
  // … and whether the rows are in strictly ascending value order (a clustered key): a dimension selected from such a
  // column is in key order and has no key twice (join_agg.cpp)
  const bool ordered_type = c.info.dtype != LLKV_DT_DECIMAL128;
  const TileSet *ts = nullptr;
  if (ordered_type && t.world == 1) {
    const int rc = get_tileset(t, 8192, &ts);
    if (rc) { (void)hipFree(d); return rc; }
    HIP_TRY(launch_ascending_check(c.d_values, c.info.dtype == LLKV_DT_INT64 ? 8 : 4, ts->d_tiles, ts->n_tiles, reinterpret_cast<uint32_t *>(d + 2), g_ctx.stream));
  }
  int64_t mm[3];
  HIP_TRY(hipMemcpyAsync(mm, d, sizeof mm, hipMemcpyDeviceToHost, g_ctx.stream));
  HIP_TRY(hipStreamSynchronize(g_ctx.stream));
  (void)hipFree(d);
  c.has_local_stats = true;
  c.local_min = mm[0];
  c.local_max = mm[1];
  if (t.world == 1) { // plans must not depend on the shard: sharded tables get table-wide statistics from the binding
    c.info.has_stats = true;
    c.info.min_i = mm[0];
    c.info.max_i = mm[1];
    c.info.ascending = ts != nullptr && mm[2] == 0;
  }
  return LLKV_OK;
}

} // namespace llkv

using namespace llkv;

extern "C" {

// ---- tables -------------------------------------------------------------------------
llkv_status llkv_hip_table_create(uint16_t table_id, const uint64_t *global_chunk_rows, uint32_t n_global_chunks,
                                  uint32_t rank, uint32_t world, llkv_hip_table **out) {
  if (!out) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "out is NULL");
  if (world == 0 || world > (uint32_t)kOctantsHost || rank >= world)
    return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "world must be 1..8 and rank < world");
  if (n_global_chunks && !global_chunk_rows) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "chunk rows is NULL");
  auto *t = new Table();
  t->table_id = table_id;
  t->rank = rank;
  t->world = world;
  t->global_chunk_rows.assign(global_chunk_rows, global_chunk_rows + n_global_chunks);
  compute_layout(*t);
  *out = reinterpret_cast<llkv_hip_table *>(t);
  return LLKV_OK;
}

void llkv_hip_table_free(llkv_hip_table *table) { delete reinterpret_cast<Table *>(table); }

llkv_status llkv_hip_table_local_chunks(const llkv_hip_table *table, uint32_t *first, uint32_t *count) {
  if (!table) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "table is NULL");
  auto *t = reinterpret_cast<const Table *>(table);
  if (first) *first = t->first_chunk;
  if (count) *count = t->n_local_chunks;
  return LLKV_OK;
}
uint64_t llkv_hip_table_total_rows(const llkv_hip_table *table) { return table ? reinterpret_cast<const Table *>(table)->total_rows : 0; }
uint64_t llkv_hip_table_local_rows(const llkv_hip_table *table) { return table ? reinterpret_cast<const Table *>(table)->local_rows : 0; }

static int check_new_column(Table *t, uint32_t field_id, uint32_t n_chunks) {
  if (!t) return set_error(LLKV_INVALID_ARGUMENT, "table is NULL");
  if (t->cols.count(field_id)) return set_error(LLKV_INVALID_ARGUMENT, "field " + std::to_string(field_id) + " already staged");
  if (n_chunks != t->n_local_chunks)
    return set_error(LLKV_INVALID_ARGUMENT, "expected " + std::to_string(t->n_local_chunks) + " local chunks, got " + std::to_string(n_chunks));
  return LLKV_OK;
}

llkv_status llkv_hip_table_set_row_ids(llkv_hip_table *table, const uint64_t *const *chunk_row_ids, uint32_t n_chunks) {
  Table *t = reinterpret_cast<Table *>(table);
  if (!t || (n_chunks && !chunk_row_ids)) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  if (n_chunks != t->n_local_chunks)
    return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "expected " + std::to_string(t->n_local_chunks) + " local chunks, got " + std::to_string(n_chunks));
  if (t->d_row_ids) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "row ids already set");
  int rc = ensure_device();
  if (rc) return (llkv_status)rc;
  // strictly ascending over the local chunks (positions and ids order alike: scans, windows, first-appearance order and
  // the joins' probe order all follow the position); dense ids from the table's first position need no translation
  bool dense = true, have = false;
  uint64_t prev = 0, at = t->local_logical_start;
  for (uint32_t i = 0; i < n_chunks; ++i) {
    const uint64_t rows = t->global_chunk_rows[t->first_chunk + i];
    if (rows && !chunk_row_ids[i]) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "chunk row-id pointer is NULL");
    for (uint64_t r = 0; r < rows; ++r, ++at) {
      const uint64_t id = chunk_row_ids[i][r];
      if (have && id <= prev) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "row ids must ascend strictly (chunk " + std::to_string(i) + ", row " + std::to_string(r) + ")");
      prev = id;
      have = true;
      dense &= id == at;
    }
  }
  if (dense) return LLKV_OK;
  void *d = nullptr;
  if ((rc = alloc_column(*t, 8, &d))) return (llkv_status)rc;
  std::vector<StagePiece> pieces;
  for (uint32_t i = 0; i < n_chunks; ++i) {
    const uint64_t rows = t->global_chunk_rows[t->first_chunk + i];
    if (rows) pieces.push_back({(char *)d + t->chunk_dev_off[i] * 8, chunk_row_ids[i], rows * 8});
  }
  if (hipStreamSynchronize(g_ctx.stream) != hipSuccess || (rc = stage_to_device(pieces))) { (void)hipFree(d); return (llkv_status)(rc ? rc : set_error(LLKV_INTERNAL, "staging copy failed")); }
  t->d_row_ids = static_cast<uint64_t *>(d);
  t->row_ids_cap = t->dev_rows + kSlackRows;
  t->last_row_id = prev;
  return LLKV_OK;
}

llkv_status llkv_hip_table_append_column(llkv_hip_table *table, uint32_t field_id, int32_t dtype,
                                         const void *const *chunk_values, uint32_t n_chunks) {
  Table *t = reinterpret_cast<Table *>(table);
  int rc = check_new_column(t, field_id, n_chunks);
  if (rc) return (llkv_status)rc;
  const uint32_t w = dtype_width(dtype);
  if (w == 0 || dtype == LLKV_DT_UTF8) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, std::string("append_column: unsupported dtype ") + dtype_name(dtype));
  if ((rc = ensure_device())) return (llkv_status)rc;
  DeviceColumn c;
  c.info.field_id = field_id;
  c.info.dtype = dtype;
  c.info.rows = t->total_rows;
  c.owned = true;
  if ((rc = alloc_column(*t, w, &c.d_values, true))) return (llkv_status)rc;
  std::vector<StagePiece> pieces;
  for (uint32_t i = 0; i < n_chunks; ++i) {
    const uint64_t rows = t->global_chunk_rows[t->first_chunk + i];
    if (rows && !chunk_values[i]) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "chunk values pointer is NULL");
    if (rows) pieces.push_back({(char *)c.d_values + t->chunk_dev_off[i] * w, chunk_values[i], rows * w});
  }
  if (hipStreamSynchronize(g_ctx.stream) != hipSuccess) return (llkv_status)set_error(LLKV_INTERNAL, "staging copy failed"); // the image is zeroed
  if ((rc = stage_to_device(pieces))) return (llkv_status)rc;
  if ((rc = column_stats_device(*t, c))) return (llkv_status)rc;
  t->cols.emplace(field_id, std::move(c));
  return LLKV_OK;
}

// every string of the chunk is one byte long (then string r is the byte data[off[0] + r])
static bool all_unit_strings(const int32_t *off, uint64_t rows) {
  if (rows == 0 || (int64_t)off[rows] - (int64_t)off[0] != (int64_t)rows) return false;
  uint32_t bad = 0;
  for (uint64_t r = 0; r < rows; ++r) bad |= (uint32_t)(off[r + 1] - off[r]) ^ 1u; // (branch-free: vectorises)
  return bad == 0;
}

llkv_status llkv_hip_table_append_utf8_column(llkv_hip_table *table, uint32_t field_id,
                                              const int32_t *const *chunk_offsets, const uint8_t *const *chunk_data,
                                              uint32_t n_chunks, const char *const *dictionary, uint32_t dict_size) {
  Table *t = reinterpret_cast<Table *>(table);
  int rc = check_new_column(t, field_id, n_chunks);
  if (rc) return (llkv_status)rc;
  if (!dictionary && t->world > 1)
    return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "sharded Utf8 columns need a table-wide dictionary (ranks must agree on the codes)");
  if (dict_size > 256) return (llkv_status)set_error(LLKV_UNSUPPORTED, "Utf8 column has more than 256 distinct values");
  if ((rc = ensure_device())) return (llkv_status)rc;
  DeviceColumn c;
  c.info.field_id = field_id;
  c.info.dtype = LLKV_DT_UTF8;
  c.info.rows = t->total_rows;
  c.owned = true;
  // dictionary-encode on the host at staging (SURVEY.md §7 "Utf8 group keys"): 1 B/row in HBM
  // (not value-initialised: 60 MB of zeroes written by one thread cost as much as coding the column on sixteen; the padding
  // rows between ragged chunks are zeroed below, every other byte is written by the coding pass)
  const bool trace = std::getenv("LLKV_HIP_TRACE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char *what) {
    if (!trace) return;
    const auto now = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[llkv utf8 staging] %-18s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  // … and not a fresh heap block either: 60 MB of new pages are faulted in while they are written, pinned for the copy and
  // unmapped on the way out (≈ 10 ms a column at SF10) — a recycled block of the pinned cache is written and copied at once
  size_t codes_bytes = (size_t)t->dev_rows + 16;
  struct PinnedBlock {
    void *p = nullptr; size_t bytes = 0;
    ~PinnedBlock() { if (p) pinned_release(p, bytes); }
  } codes_block;
  codes_block.p = pinned_acquire(&codes_bytes);
  codes_block.bytes = codes_bytes;
  if (!codes_block.p) return (llkv_status)set_error(LLKV_INTERNAL, "no pinned host memory for the dictionary codes");
  struct { uint8_t *p; uint8_t *data() const { return p; } } codes{static_cast<uint8_t *>(codes_block.p)};
  for (uint32_t i = 0; i < n_chunks; ++i) {
    const uint64_t end = t->chunk_dev_off[i] + t->global_chunk_rows[t->first_chunk + i];
    std::memset(codes.data() + end, 0, t->chunk_dev_off[i + 1] - end);
  }
  std::memset(codes.data() + t->dev_rows, 0, 16);
  std::map<std::string, uint8_t> dict;
  const bool fixed = dictionary != nullptr;
  for (uint32_t d = 0; d < dict_size && fixed; ++d) {
    std::string s = dictionary[d] ? dictionary[d] : "";
    if (!dict.emplace(s, (uint8_t)d).second) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "duplicate dictionary entry '" + s + "'");
    c.info.dictionary.push_back(s);
  }
  mark("buffer");
  if (!fixed) {
    // codes follow the order of first appearance: every chunk lists its distinct values in that order (in
    // parallel), the lists are merged in chunk order
    std::vector<std::vector<std::string>> seen(n_chunks);
    rc = for_each_chunk_parallel(n_chunks, [&](uint32_t i) -> int {
      const uint64_t rows = t->global_chunk_rows[t->first_chunk + i];
      const int32_t *off = chunk_offsets[i];
      const uint8_t *data = chunk_data[i];
      bool one_byte[256] = {};
      std::map<std::string, int> local;
      if (all_unit_strings(off, rows)) { // a chunk of 1-byte strings (TPC-H flags): the byte is the string
        const uint8_t *b = data + off[0];
        for (uint64_t r = 0; r < rows; ++r)
          if (!one_byte[b[r]]) { one_byte[b[r]] = true; seen[i].emplace_back(1, (char)b[r]); }
        return LLKV_OK;
      }
      for (uint64_t r = 0; r < rows; ++r) {
        const int32_t len = off[r + 1] - off[r];
        if (len == 1) {
          const uint8_t ch = data[off[r]];
          if (!one_byte[ch]) { one_byte[ch] = true; seen[i].emplace_back(1, (char)ch); }
        } else {
          std::string s((const char *)data + off[r], (size_t)len);
          if (local.emplace(s, 0).second) seen[i].push_back(std::move(s));
        }
        if (seen[i].size() > 256) return set_error(LLKV_UNSUPPORTED, "Utf8 column has more than 256 distinct values");
      }
      return LLKV_OK;
    });
    if (rc) return (llkv_status)rc;
    for (uint32_t i = 0; i < n_chunks; ++i)
      for (std::string &s : seen[i])
        if (!dict.count(s)) {
          if (dict.size() >= 256) return (llkv_status)set_error(LLKV_UNSUPPORTED, "Utf8 column has more than 256 distinct values");
          dict.emplace(s, (uint8_t)dict.size());
          c.info.dictionary.push_back(s);
        }
  }
  mark("distinct values");
  rc = for_each_chunk_parallel(n_chunks, [&](uint32_t i) -> int {
    const uint64_t rows = t->global_chunk_rows[t->first_chunk + i];
    const int32_t *off = chunk_offsets[i];
    const uint8_t *data = chunk_data[i];
    uint8_t *dst = codes.data() + t->chunk_dev_off[i];
    int16_t lut[256]; // fast path for 1-byte strings (TPC-H flags)
    std::fill(std::begin(lut), std::end(lut), (int16_t)-1);
    auto code_of = [&](const std::string &s, uint8_t *out) -> int {
      auto it = dict.find(s);
      if (it == dict.end()) return set_error(LLKV_INVALID_ARGUMENT, "value '" + s + "' is not in the supplied dictionary");
      *out = it->second;
      return LLKV_OK;
    };
    if (all_unit_strings(off, rows)) {
      uint8_t code_of_byte[256];
      bool known[256] = {};
      const uint8_t *b = data + off[0];
      for (uint64_t r = 0; r < rows; ++r) { // (first: which bytes occur — a dictionary miss is reported before anything is written)
        if (known[b[r]]) continue;
        int e;
        if ((e = code_of(std::string(1, (char)b[r]), &code_of_byte[b[r]]))) return e;
        known[b[r]] = true;
      }
      for (uint64_t r = 0; r < rows; ++r) dst[r] = code_of_byte[b[r]];
      return LLKV_OK;
    }
    for (uint64_t r = 0; r < rows; ++r) {
      const int32_t len = off[r + 1] - off[r];
      int e;
      if (len == 1) {
        const uint8_t ch = data[off[r]];
        if (lut[ch] < 0) {
          uint8_t code;
          if ((e = code_of(std::string(1, (char)ch), &code))) return e;
          lut[ch] = code;
        }
        dst[r] = (uint8_t)lut[ch];
      } else if ((e = code_of(std::string((const char *)data + off[r], (size_t)len), &dst[r]))) {
        return e;
      }
    }
    return LLKV_OK;
  });
  if (rc) return (llkv_status)rc;
  mark("codes");
  if ((rc = alloc_column(*t, 1, &c.d_values))) return (llkv_status)rc;
  if (hipStreamSynchronize(g_ctx.stream) != hipSuccess) return (llkv_status)set_error(LLKV_INTERNAL, "staging copy failed");
  mark("device buffer");
  if ((rc = stage_from_pinned(c.d_values, codes.data(), (size_t)t->dev_rows))) return (llkv_status)rc;
  mark("copy");
  t->cols.emplace(field_id, std::move(c));
  return LLKV_OK;
}

llkv_status llkv_hip_table_local_column_stats(const llkv_hip_table *table, uint32_t field_id, int32_t *has_stats,
                                              int64_t *min_value, int64_t *max_value) {
  const Table *t = reinterpret_cast<const Table *>(table);
  if (!t) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "table is NULL");
  auto it = t->cols.find(field_id);
  if (it == t->cols.end()) return (llkv_status)set_error(LLKV_NOT_FOUND, "field " + std::to_string(field_id) + " is not staged");
  if (has_stats) *has_stats = it->second.has_local_stats ? 1 : 0;
  if (min_value) *min_value = it->second.local_min;
  if (max_value) *max_value = it->second.local_max;
  return LLKV_OK;
}

void llkv_hip_pinned_stats(uint64_t *cached_bytes, uint64_t *outstanding_bytes) { pinned_stats(cached_bytes, outstanding_bytes); }

void llkv_hip_staging_stats(uint64_t *bytes, double *seconds) {
  staging_totals(bytes, seconds);
}

llkv_status llkv_hip_table_set_column_stats(llkv_hip_table *table, uint32_t field_id, int64_t min_value, int64_t max_value) {
  Table *t = reinterpret_cast<Table *>(table);
  if (!t) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "table is NULL");
  auto it = t->cols.find(field_id);
  if (it == t->cols.end()) return (llkv_status)set_error(LLKV_NOT_FOUND, "field " + std::to_string(field_id) + " is not staged");
  DeviceColumn &c = it->second;
  if (min_value > max_value) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "min exceeds max");
  // a bound that does not cover this rank's rows would make "provably no overflow" and dense group ids wrong
  if (c.has_local_stats && t->local_rows && !(min_value <= c.local_min && max_value >= c.local_max))
    return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "column statistics do not cover the staged values");
  c.info.has_stats = true;
  c.info.min_i = min_value;
  c.info.max_i = max_value;
  return LLKV_OK;
}

llkv_status llkv_hip_table_local_column_float_stats(const llkv_hip_table *table, uint32_t field_id, int32_t *has_stats, double *abs_max, double *abs_min_nonzero) {
  const Table *t = reinterpret_cast<const Table *>(table);
  if (!t) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "table is NULL");
  auto it = t->cols.find(field_id);
  if (it == t->cols.end()) return (llkv_status)set_error(LLKV_NOT_FOUND, "field " + std::to_string(field_id) + " is not staged");
  if (has_stats) *has_stats = it->second.has_local_fstats ? 1 : 0;
  if (abs_max) *abs_max = it->second.local_f_absmax;
  if (abs_min_nonzero) *abs_min_nonzero = it->second.local_f_absmin_nz;
  return LLKV_OK;
}

llkv_status llkv_hip_table_set_column_float_stats(llkv_hip_table *table, uint32_t field_id, double abs_max, double abs_min_nonzero) {
  Table *t = reinterpret_cast<Table *>(table);
  if (!t) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "table is NULL");
  auto it = t->cols.find(field_id);
  if (it == t->cols.end()) return (llkv_status)set_error(LLKV_NOT_FOUND, "field " + std::to_string(field_id) + " is not staged");
  DeviceColumn &c = it->second;
  if (!(abs_max >= 0.0) || !(abs_min_nonzero >= 0.0) || !std::isfinite(abs_max) || abs_min_nonzero > abs_max)
    return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "float statistics must satisfy 0 ≤ smallest non-zero |v| ≤ largest |v| < ∞");
  // bounds that do not cover this rank's rows would break the exactness of the sums built on them
  if (c.has_local_fstats && t->local_rows && (abs_max < c.local_f_absmax || (c.local_f_absmin_nz > 0.0 && !(abs_min_nonzero > 0.0 && abs_min_nonzero <= c.local_f_absmin_nz))))
    return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "column statistics do not cover the staged values");
  c.info.has_fstats = true;
  c.info.f_absmax = abs_max;
  c.info.f_absmin_nz = abs_min_nonzero;
  return LLKV_OK;
}

llkv_status llkv_hip_table_local_column_all_finite(const llkv_hip_table *table, uint32_t field_id, int32_t *all_finite) {
  const Table *t = reinterpret_cast<const Table *>(table);
  if (!t || !all_finite) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  auto it = t->cols.find(field_id);
  if (it == t->cols.end()) return (llkv_status)set_error(LLKV_NOT_FOUND, "field " + std::to_string(field_id) + " is not staged");
  *all_finite = it->second.has_local_fstats && it->second.local_f_all_finite ? 1 : 0;
  return LLKV_OK;
}

llkv_status llkv_hip_table_set_column_all_finite(llkv_hip_table *table, uint32_t field_id, int32_t all_finite) {
  Table *t = reinterpret_cast<Table *>(table);
  if (!t) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "table is NULL");
  auto it = t->cols.find(field_id);
  if (it == t->cols.end()) return (llkv_status)set_error(LLKV_NOT_FOUND, "field " + std::to_string(field_id) + " is not staged");
  DeviceColumn &c = it->second;
  if (all_finite && c.has_local_fstats && t->local_rows && !c.local_f_all_finite)
    return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "this rank's rows of the column hold NaN or ±∞");
  c.info.f_all_finite = all_finite != 0;
  return LLKV_OK;
}

llkv_status llkv_hip_table_append_decimal128_column(llkv_hip_table *table, uint32_t field_id, int32_t precision, int32_t scale,
                                                    const void *const *chunk_values, uint32_t n_chunks) {
  Table *t = reinterpret_cast<Table *>(table);
  int rc = check_new_column(t, field_id, n_chunks);
  if (rc) return (llkv_status)rc;
  if (precision < 1 || precision > 38 || scale > precision || scale < -128)
    return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "invalid Decimal128 precision/scale");
  if ((rc = ensure_device())) return (llkv_status)rc;
  // narrow the 16-byte raw values to the 8 B/row device image.  A column with a value that needs more than 64 bits is
  // staged as two 8 B/row buffers instead (low halves, high halves): SUM / TOTAL / AVG and the counts take it, every
  // other use keeps the caller's CPU route (plan.cpp: slot_of)
  std::vector<int64_t> narrow(t->dev_rows + 16, 0), high;
  std::atomic<bool> any_wide{false};
  rc = for_each_chunk_parallel(n_chunks, [&](uint32_t i) -> int {
    const uint64_t rows = t->global_chunk_rows[t->first_chunk + i];
    if (rows && !chunk_values[i]) return set_error(LLKV_INVALID_ARGUMENT, "chunk values pointer is NULL");
    const int64_t *src = static_cast<const int64_t *>(chunk_values[i]); // (lo, hi) pairs, little endian
    int64_t *dst = narrow.data() + t->chunk_dev_off[i];
    bool wide = false;
    for (uint64_t r = 0; r < rows; ++r) {
      const int64_t lo = src[2 * r], hi = src[2 * r + 1];
      wide |= hi != (lo >> 63);
      dst[r] = lo;
    }
    if (wide) any_wide = true;
    return LLKV_OK;
  });
  if (rc) return (llkv_status)rc;
  DeviceColumn c;
  c.info.field_id = field_id;
  c.info.dtype = LLKV_DT_DECIMAL128;
  c.info.precision = precision;
  c.info.scale = scale;
  c.info.rows = t->total_rows;
  c.owned = true;
  if (any_wide) {
    // (a sharded table would need the ranks to agree on the layout and on max|v|: not exchanged yet)
    if (t->world != 1) return (llkv_status)set_error(LLKV_UNSUPPORTED, "Decimal128 value beyond 64 bits in field " + std::to_string(field_id) + " of a sharded table");
    high.assign(t->dev_rows + 16, 0);
    unsigned __int128 absmax = 0;
    __int128 vmin = (__int128)(~(unsigned __int128)0 >> 1), vmax = -vmin - 1; // smallest / largest value (MIN / MAX run over v − min, plan.cpp)
    for (uint32_t i = 0; i < n_chunks; ++i) { // (sequential: wide columns are rare, and max|v| is one value)
      const uint64_t rows = t->global_chunk_rows[t->first_chunk + i];
      const int64_t *src = static_cast<const int64_t *>(chunk_values[i]);
      int64_t *dst = high.data() + t->chunk_dev_off[i];
      for (uint64_t r = 0; r < rows; ++r) {
        dst[r] = src[2 * r + 1];
        const __int128 v = ((__int128)src[2 * r + 1] << 64) | (unsigned __int128)(uint64_t)src[2 * r];
        const unsigned __int128 mag = v < 0 ? (unsigned __int128)0 - (unsigned __int128)v : (unsigned __int128)v;
        absmax = mag > absmax ? mag : absmax;
        vmin = v < vmin ? v : vmin;
        vmax = v > vmax ? v : vmax;
      }
    }
    c.info.wide128 = true;
    c.info.wide_absmax_hi = (uint64_t)(absmax >> 64);
    c.info.wide_absmax_lo = (uint64_t)absmax;
    c.info.wide_min_hi = (uint64_t)(vmin >> 64); c.info.wide_min_lo = (uint64_t)vmin;
    c.info.wide_max_hi = (uint64_t)(vmax >> 64); c.info.wide_max_lo = (uint64_t)vmax;
  }
  if ((rc = alloc_column(*t, 8, &c.d_values)) || (any_wide && (rc = alloc_column(*t, 8, &c.d_hi)))) {
    if (c.d_values) (void)hipFree(c.d_values);
    return (llkv_status)rc;
  }
  if (hipStreamSynchronize(g_ctx.stream) != hipSuccess) return (llkv_status)set_error(LLKV_INTERNAL, "staging copy failed");
  rc = any_wide ? stage_to_device({{c.d_values, narrow.data(), (size_t)t->dev_rows * 8}, {c.d_hi, high.data(), (size_t)t->dev_rows * 8}})
                : stage_to_device({{c.d_values, narrow.data(), (size_t)t->dev_rows * 8}});
  if (!rc && !any_wide) rc = column_stats_device(*t, c); // (integer statistics describe the narrowed image only)
  if (rc) {
    (void)hipFree(c.d_values);
    if (c.d_hi) (void)hipFree(c.d_hi);
    return (llkv_status)rc;
  }
  t->cols.emplace(field_id, std::move(c));
  return LLKV_OK;
}

llkv_status llkv_hip_table_set_column_validity(llkv_hip_table *table, uint32_t field_id,
                                               const uint8_t *const *chunk_validity, uint32_t n_chunks) {
  Table *t = reinterpret_cast<Table *>(table);
  if (!t) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "table is NULL");
  auto it = t->cols.find(field_id);
  if (it == t->cols.end()) return (llkv_status)set_error(LLKV_NOT_FOUND, "field " + std::to_string(field_id) + " is not staged");
  if (n_chunks != t->n_local_chunks)
    return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "expected " + std::to_string(t->n_local_chunks) + " local chunks, got " + std::to_string(n_chunks));
  if (!chunk_validity) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "chunk validity array is NULL");
  int rc = ensure_device();
  if (rc) return (llkv_status)rc;
  DeviceColumn &c = it->second;
  // Arrow bitmaps → 1 B/row in the device row layout (rows of padding between chunks stay 0; tiles never
  // select them)
  std::vector<uint8_t> mask(t->dev_rows + 16, 0);
  std::atomic<uint64_t> nulls{0};
  rc = for_each_chunk_parallel(n_chunks, [&](uint32_t i) -> int {
    const uint64_t rows = t->global_chunk_rows[t->first_chunk + i];
    uint8_t *dst = mask.data() + t->chunk_dev_off[i];
    const uint8_t *bits = chunk_validity[i];
    if (!bits) { std::memset(dst, 1, rows); return LLKV_OK; }
    uint64_t n = 0;
    for (uint64_t r = 0; r < rows; ++r) {
      const uint8_t v = (bits[r >> 3] >> (r & 7)) & 1u;
      dst[r] = v;
      n += !v;
    }
    nulls += n;
    return LLKV_OK;
  });
  if (rc) return (llkv_status)rc;
  if (c.d_valid) { (void)hipFree(c.d_valid); c.d_valid = nullptr; }
  c.info.nullable = false;
  // whether a column "has NULL cells" must not depend on the shard: with world > 1 any supplied bitmap makes
  // the column nullable on every rank (plans must agree across ranks)
  bool any_bitmap = false;
  for (uint32_t i = 0; i < n_chunks; ++i) any_bitmap |= chunk_validity[i] != nullptr;
  if (nulls == 0 && !(t->world > 1 && any_bitmap)) return LLKV_OK;
  void *d = nullptr;
  if ((rc = alloc_column(*t, 1, &d))) return (llkv_status)rc;
  if (hipStreamSynchronize(g_ctx.stream) != hipSuccess) { (void)hipFree(d); return (llkv_status)set_error(LLKV_INTERNAL, "staging copy failed"); }
  if ((rc = stage_to_device({{d, mask.data(), (size_t)t->dev_rows}}))) { (void)hipFree(d); return (llkv_status)rc; }
  c.d_valid = (uint8_t *)d;
  c.info.nullable = true;
  return LLKV_OK;
}

// Sharded tables: what plans depend on must not depend on the shard (a rank without NULL cells lowered a
// plan without the validity slots its neighbour had — different lanes, different exchange image).
llkv_status llkv_hip_table_share_metadata(llkv_hip_table *table) {
  Table *t = reinterpret_cast<Table *>(table);
  if (!t) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "table is NULL");
  if (t->world == 1) return LLKV_OK;
  if (!comm_ready() || comm_world() != t->world || comm_rank() != t->rank)
    return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "the table's (rank, world) is not the communicator's");
  struct Rec { uint32_t field; int32_t has_stats; int64_t lo, hi; int32_t nullable; uint32_t local_rows_nonzero; int32_t has_fstats, all_finite; double f_absmax, f_absmin_nz; int32_t no_neg_zero, pad_; };
  std::vector<Rec> mine;
  for (auto &kv : t->cols) // std::map: ascending field ids on every rank
    mine.push_back({kv.first, kv.second.has_local_stats ? 1 : 0, kv.second.local_min, kv.second.local_max, kv.second.info.nullable ? 1 : 0, t->local_rows ? 1u : 0u,
                    kv.second.has_local_fstats ? 1 : 0, kv.second.local_f_all_finite ? 1 : 0, kv.second.local_f_absmax, kv.second.local_f_absmin_nz,
                    kv.second.local_f_no_neg_zero ? 1 : 0, 0});
  std::vector<uint8_t> all;
  std::vector<uint64_t> off;
  int rc = comm_allgather_v(mine.data(), mine.size() * sizeof(Rec), &all, &off);
  if (rc) return (llkv_status)rc;
  for (uint32_t r = 0; r < t->world; ++r) {
    if (off[r + 1] - off[r] != mine.size() * sizeof(Rec)) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "the ranks staged different column sets");
    const Rec *theirs = reinterpret_cast<const Rec *>(all.data() + off[r]);
    for (size_t i = 0; i < mine.size(); ++i)
      if (theirs[i].field != mine[i].field) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "the ranks staged different column sets");
  }
  size_t i = 0;
  for (auto &kv : t->cols) {
    DeviceColumn &c = kv.second;
    bool all_stats = true, all_fstats = true, any_rows = false, any_nullable = false, all_finite = true, no_neg_zero = true;
    int64_t lo = INT64_MAX, hi = INT64_MIN;
    double fmax = 0.0, fmin_nz = 0.0;
    for (uint32_t r = 0; r < t->world; ++r) {
      const Rec &x = reinterpret_cast<const Rec *>(all.data() + off[r])[i];
      any_nullable |= x.nullable != 0;
      if (!x.local_rows_nonzero) continue; // a rank without rows constrains nothing
      any_rows = true;
      all_finite &= x.all_finite != 0;
      no_neg_zero &= x.no_neg_zero != 0;
      if (x.has_fstats) {
        fmax = std::max(fmax, x.f_absmax);
        if (x.f_absmin_nz > 0.0) fmin_nz = fmin_nz > 0.0 ? std::min(fmin_nz, x.f_absmin_nz) : x.f_absmin_nz;
      } else all_fstats = false;
      if (!x.has_stats) { all_stats = false; continue; }
      lo = std::min(lo, x.lo);
      hi = std::max(hi, x.hi);
    }
    if (all_fstats && any_rows) {
      c.info.has_fstats = true;
      c.info.f_absmax = fmax;
      c.info.f_absmin_nz = fmin_nz;
      c.info.f_all_finite = all_finite;
      c.info.f_no_neg_zero = no_neg_zero;
    }
    if (all_stats && any_rows && lo <= hi) {
      c.info.has_stats = true;
      c.info.min_i = lo;
      c.info.max_i = hi;
    }
    if (any_nullable && !c.info.nullable) { // no NULL cell here: an all-present mask keeps this rank's plans those of its neighbours
      if ((rc = ensure_device())) return (llkv_status)rc;
      void *d = nullptr;
      const uint64_t bytes = t->dev_rows + kSlackRows;
      if (hipMalloc(&d, bytes) != hipSuccess) return (llkv_status)set_error(LLKV_INTERNAL, "device allocation failed");
      if (hipMemsetAsync(d, 1, bytes, g_ctx.stream) != hipSuccess || hipStreamSynchronize(g_ctx.stream) != hipSuccess) {
        (void)hipFree(d);
        return (llkv_status)set_error(LLKV_INTERNAL, "validity mask could not be initialised");
      }
      c.d_valid = static_cast<uint8_t *>(d);
      c.info.nullable = true;
    }
    ++i;
  }
  return LLKV_OK;
}

// ---- incremental growth ---------------------------------------------------------------------------------------------------------
// The reference appends chunks to its columns (ColumnStore::append llkv-column-map/src/store/core.rs:787: a RecordBatch of new rows
// becomes one more chunk per column, with the row-id shadow chunk beside it); re-staging a 2.3 GB table image for every INSERT
// costs 60 ms of PCIe — 180 × the query.  Here n_new chunks follow the table's last chunk: only THEIR bytes cross the link; a
// column image that has no room moves once into a larger buffer on the device (a D2D copy at HBM speed, 25 % headroom for the
// appends to come); dictionaries grow by the new strings (codes of the old rows stay); validity masks, statistics (one device
// pass per column, as at staging) and row ids follow.  Everything that can fail for a reason of the DATA is checked before the
// first buffer is touched.  Unsharded tables only: with world > 1 the canonical octants ⌊j·C/8⌋ move with the chunk count and
// rows would change ranks.
namespace {
struct GrownBuffer {
  void **slot = nullptr;   // the DeviceColumn member to repoint
  void *fresh = nullptr;
  uint32_t width = 0;
  uint8_t fill = 0;
};
uint64_t with_headroom(uint64_t rows) { return rows + rows / 4 + kSlackRows; }
} // namespace

static int append_chunks_impl(Table *t, const uint64_t *chunk_rows, uint32_t n_new, const llkv_column_chunks *columns, uint32_t n_columns,
                              const uint64_t *const *chunk_row_ids) {
  if (!t || (n_new && !chunk_rows) || (n_columns && !columns)) return set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  if (t->world != 1) return set_error(LLKV_UNSUPPORTED, "append to a sharded table (the canonical octants move with the chunk count): re-stage the shards");
  if (n_new == 0) return LLKV_OK;
  int rc = ensure_device();
  if (rc) return rc;
  if (n_columns != t->cols.size()) return set_error(LLKV_INVALID_ARGUMENT, "an append carries the new chunks of EVERY staged column (" + std::to_string(t->cols.size()) + " staged, " + std::to_string(n_columns) + " given)");
  std::map<uint32_t, const llkv_column_chunks *> by_field;
  for (uint32_t i = 0; i < n_columns; ++i) {
    if (!t->cols.count(columns[i].field_id)) return set_error(LLKV_NOT_FOUND, "field " + std::to_string(columns[i].field_id) + " is not staged");
    if (!by_field.emplace(columns[i].field_id, &columns[i]).second) return set_error(LLKV_INVALID_ARGUMENT, "field " + std::to_string(columns[i].field_id) + " given twice");
  }
  // ---- the new layout (nothing of the table is touched yet) ------------------------------------------------------------------
  const uint32_t old_chunks = t->n_local_chunks;
  const uint64_t old_dev_rows = t->dev_rows;
  std::vector<uint64_t> off(n_new + 1);
  off[0] = old_dev_rows; // (compute_layout rounds every chunk's end up to 16 rows: the old image ends on such a boundary)
  uint64_t new_rows = 0;
  for (uint32_t i = 0; i < n_new; ++i) { off[i + 1] = round_up(off[i] + chunk_rows[i], 16); new_rows += chunk_rows[i]; }
  const uint64_t new_dev_rows = off[n_new];
  if (t->total_rows + new_rows >= (1ull << 38)) return set_error(LLKV_UNSUPPORTED, "table beyond 2^38 rows");

  // ---- host-side preparation per column: everything the DATA can refuse ------------------------------------------------------
  struct Prepared {
    std::vector<StagePiece> pieces;        // straight from the caller's buffers (fixed width)
    std::vector<int64_t> narrow;           // Decimal128: the 64-bit images of the new rows, [new_dev_rows − old_dev_rows]
    std::vector<uint8_t> codes;            // Utf8: dictionary codes of the new rows
    std::vector<std::string> new_words;    // … and the strings that join the dictionary
    std::vector<uint8_t> mask;             // validity bytes of the new rows (empty: every cell present and the column stays without a mask)
    bool new_nulls = false;
  };
  std::map<uint32_t, Prepared> prep;
  const uint64_t span = new_dev_rows - old_dev_rows;
  for (auto &kv : t->cols) {
    DeviceColumn &c = kv.second;
    const llkv_column_chunks &in = *by_field[kv.first];
    Prepared &p = prep[kv.first];
    if (!c.owned) return set_error(LLKV_UNSUPPORTED, "append to a column the library does not own");
    if (c.info.wide128) return set_error(LLKV_UNSUPPORTED, "append to a Decimal128 column with values beyond 64 bits: re-stage it");
    const uint32_t w = dtype_width(c.info.dtype);
    if (c.info.dtype == LLKV_DT_UTF8) {
      if (!in.offsets || !in.data) return set_error(LLKV_INVALID_ARGUMENT, "Utf8 field " + std::to_string(kv.first) + " needs offsets and data");
      std::map<std::string, uint8_t> dict;
      for (size_t d = 0; d < c.info.dictionary.size(); ++d) dict.emplace(c.info.dictionary[d], (uint8_t)d);
      p.codes.assign(span + 16, 0);
      for (uint32_t i = 0; i < n_new; ++i) {
        if (chunk_rows[i] && (!in.offsets[i] || !in.data[i])) return set_error(LLKV_INVALID_ARGUMENT, "chunk pointer is NULL");
        const int32_t *o = in.offsets[i];
        for (uint64_t r = 0; r < chunk_rows[i]; ++r) {
          if (o[r + 1] < o[r]) return set_error(LLKV_INVALID_ARGUMENT, "Utf8 offsets must not descend");
          std::string sv((const char *)in.data[i] + o[r], (size_t)(o[r + 1] - o[r]));
          auto it = dict.find(sv);
          if (it == dict.end()) {
            if (dict.size() >= 256) return set_error(LLKV_UNSUPPORTED, "the append takes Utf8 field " + std::to_string(kv.first) + " beyond 256 distinct values: re-stage the table");
            it = dict.emplace(sv, (uint8_t)dict.size()).first;
            p.new_words.push_back(sv);
          }
          p.codes[off[i] - old_dev_rows + r] = it->second;
        }
      }
    } else if (c.info.dtype == LLKV_DT_DECIMAL128) {
      if (!in.values) return set_error(LLKV_INVALID_ARGUMENT, "field " + std::to_string(kv.first) + " needs values");
      p.narrow.assign(span + 16, 0);
      for (uint32_t i = 0; i < n_new; ++i) {
        if (chunk_rows[i] && !in.values[i]) return set_error(LLKV_INVALID_ARGUMENT, "chunk values pointer is NULL");
        const int64_t *src = static_cast<const int64_t *>(in.values[i]);
        for (uint64_t r = 0; r < chunk_rows[i]; ++r) {
          if (src[2 * r + 1] != (src[2 * r] >> 63)) return set_error(LLKV_UNSUPPORTED, "the append brings a Decimal128 value beyond 64 bits into field " + std::to_string(kv.first) + ": re-stage the column");
          p.narrow[off[i] - old_dev_rows + r] = src[2 * r];
        }
      }
    } else {
      if (!in.values || w == 0) return set_error(LLKV_INVALID_ARGUMENT, "field " + std::to_string(kv.first) + " needs values");
      for (uint32_t i = 0; i < n_new; ++i) {
        if (chunk_rows[i] && !in.values[i]) return set_error(LLKV_INVALID_ARGUMENT, "chunk values pointer is NULL");
        if (chunk_rows[i]) p.pieces.push_back({nullptr, in.values[i], (size_t)chunk_rows[i] * w}); // (destination: once the buffer is known)
      }
    }
    bool any_bitmap = false;
    for (uint32_t i = 0; in.validity && i < n_new; ++i) any_bitmap |= in.validity[i] != nullptr;
    if (any_bitmap || c.info.nullable) {
      p.mask.assign(span + 16, 0);
      for (uint32_t i = 0; i < n_new; ++i) {
        uint8_t *dst = p.mask.data() + (off[i] - old_dev_rows);
        const uint8_t *bits = in.validity ? in.validity[i] : nullptr;
        if (!bits) { std::memset(dst, 1, chunk_rows[i]); continue; }
        for (uint64_t r = 0; r < chunk_rows[i]; ++r) { dst[r] = (bits[r >> 3] >> (r & 7)) & 1u; p.new_nulls |= !dst[r]; }
      }
      if (!c.info.nullable && !p.new_nulls) p.mask.clear(); // still no NULL cell: still no mask
    }
  }
  // row ids: ascending beyond the table's last id; the dense continuation of a table with dense ids needs no image
  bool ids_dense = true;
  if (chunk_row_ids) {
    uint64_t prev = t->d_row_ids ? t->last_row_id : (t->total_rows ? t->local_logical_start + t->local_rows - 1 : 0), at = t->local_logical_start + t->local_rows;
    bool have = t->total_rows != 0;
    for (uint32_t i = 0; i < n_new; ++i) {
      if (chunk_rows[i] && !chunk_row_ids[i]) return set_error(LLKV_INVALID_ARGUMENT, "chunk row-id pointer is NULL");
      for (uint64_t r = 0; r < chunk_rows[i]; ++r, ++at) {
        const uint64_t id = chunk_row_ids[i][r];
        if (have && id <= prev) return set_error(LLKV_INVALID_ARGUMENT, "row ids must ascend strictly beyond the table's last id");
        prev = id; have = true;
        ids_dense &= id == at;
      }
    }
  } else if (t->d_row_ids) {
    return set_error(LLKV_INVALID_ARGUMENT, "the table has its own row ids: the appended chunks need theirs");
  }
  const bool want_id_image = t->d_row_ids != nullptr || (chunk_row_ids && !ids_dense);

  // ---- device side: larger buffers first (all of them, or none), then the copies ---------------------------------------------
  HIP_TRY(hipDeviceSynchronize()); // executions launched over the old image have finished before a buffer moves
  hipStream_t s = g_ctx.stream;
  const uint64_t need = new_dev_rows + kSlackRows;
  std::vector<GrownBuffer> grown;
  auto release_fresh = [&] { for (GrownBuffer &g : grown) if (g.fresh) (void)hipFree(g.fresh); };
  auto want_room = [&](void **slot, uint64_t cap, uint32_t width, uint8_t fill) -> int {
    if (cap >= need) return LLKV_OK;
    GrownBuffer g{slot, nullptr, width, fill};
    if (hipMalloc(&g.fresh, with_headroom(new_dev_rows) * width) != hipSuccess) { release_fresh(); return set_error(LLKV_INTERNAL, "device allocation failed while growing a column"); }
    grown.push_back(g);
    return LLKV_OK;
  };
  for (auto &kv : t->cols) {
    DeviceColumn &c = kv.second;
    const uint32_t w = dtype_width(c.info.dtype);
    const uint64_t cap = c.cap_rows ? c.cap_rows : old_dev_rows + kSlackRows;
    if ((rc = want_room(&c.d_values, cap, w, 0))) return rc;
    Prepared &p = prep[kv.first];
    if (!p.mask.empty()) {
      void **vslot = reinterpret_cast<void **>(&c.d_valid);
      const uint64_t vcap = c.d_valid ? (c.valid_cap_rows ? c.valid_cap_rows : old_dev_rows + kSlackRows) : 0;
      if ((rc = want_room(vslot, vcap, 1, 1))) return rc;
    }
  }
  void *ids_slot = t->d_row_ids;
  if (want_id_image && (rc = want_room(&ids_slot, t->d_row_ids ? t->row_ids_cap : 0, 8, 0))) return rc;
  for (GrownBuffer &g : grown) { // old rows move on the device; what lies behind them starts as `fill`
    void *old = *g.slot;
    const uint64_t cap_rows = with_headroom(new_dev_rows);
    if (old) HIP_TRY(hipMemcpyAsync(g.fresh, old, old_dev_rows * g.width, hipMemcpyDeviceToDevice, s));
    else if (g.fill) HIP_TRY(hipMemsetAsync(g.fresh, g.fill, old_dev_rows * g.width, s)); // a validity mask that did not exist: every old cell present
    HIP_TRY(hipMemsetAsync((char *)g.fresh + old_dev_rows * g.width, 0, (cap_rows - old_dev_rows) * g.width, s));
  }
  HIP_TRY(hipStreamSynchronize(s));
  for (GrownBuffer &g : grown) {
    if (*g.slot && g.slot != &ids_slot) (void)hipFree(*g.slot);
    *g.slot = g.fresh;
  }
  const uint64_t grown_cap = with_headroom(new_dev_rows);
  for (auto &kv : t->cols) {
    DeviceColumn &c = kv.second;
    for (GrownBuffer &g : grown) {
      if (g.slot == &c.d_values) c.cap_rows = grown_cap;
      if (g.slot == reinterpret_cast<void **>(&c.d_valid)) c.valid_cap_rows = grown_cap;
    }
    if (!c.cap_rows) c.cap_rows = old_dev_rows + kSlackRows;
  }
  if (want_id_image) {
    const bool moved = ids_slot != t->d_row_ids;
    if (moved) {
      if (t->d_row_ids) (void)hipFree(t->d_row_ids);
      else HIP_TRY(hj_launch_iota_u64(static_cast<uint64_t *>(ids_slot), old_dev_rows, t->local_logical_start, s)); // dense until now (world = 1: no padding rows matter — ids of padding rows are never read)
      t->d_row_ids = static_cast<uint64_t *>(ids_slot);
      t->row_ids_cap = grown_cap;
    }
  }
  // the new chunks' bytes: the only host → HBM traffic of the append
  for (auto &kv : t->cols) {
    DeviceColumn &c = kv.second;
    Prepared &p = prep[kv.first];
    const uint32_t w = dtype_width(c.info.dtype);
    if (c.info.dtype == LLKV_DT_UTF8) {
      if ((rc = stage_to_device({{(char *)c.d_values + old_dev_rows, p.codes.data(), (size_t)span}}))) return rc;
      for (std::string &wd : p.new_words) c.info.dictionary.push_back(wd);
    } else if (c.info.dtype == LLKV_DT_DECIMAL128) {
      if ((rc = stage_to_device({{(char *)c.d_values + old_dev_rows * 8, p.narrow.data(), (size_t)span * 8}}))) return rc;
    } else {
      size_t k = 0;
      for (uint32_t i = 0; i < n_new; ++i) if (chunk_rows[i]) p.pieces[k++].d_dst = (char *)c.d_values + off[i] * w;
      if ((rc = stage_to_device(p.pieces))) return rc;
    }
    if (!p.mask.empty()) {
      if ((rc = stage_to_device({{c.d_valid + old_dev_rows, p.mask.data(), (size_t)span}}))) return rc;
      c.info.nullable = true;
    }
  }
  if (want_id_image) {
    std::vector<StagePiece> pieces;
    for (uint32_t i = 0; i < n_new; ++i) if (chunk_rows[i]) pieces.push_back({(char *)t->d_row_ids + off[i] * 8, chunk_row_ids[i], (size_t)chunk_rows[i] * 8});
    if ((rc = stage_to_device(pieces))) return rc;
  }
  if (chunk_row_ids) for (uint32_t i = n_new; i-- > 0;) if (chunk_rows[i]) { t->last_row_id = chunk_row_ids[i][chunk_rows[i] - 1]; break; }

  // ---- the table is the grown one from here: layout, tile lists, statistics, generation --------------------------------------
  for (uint32_t i = 0; i < n_new; ++i) t->global_chunk_rows.push_back(chunk_rows[i]);
  compute_layout(*t);
  if (t->n_local_chunks != old_chunks + n_new || t->dev_rows != new_dev_rows) return set_error(LLKV_INTERNAL, "append: the layout disagrees with the table's");
  {
    std::lock_guard<std::mutex> lk(t->mu);
    for (auto &kv : t->tilesets) { if (kv.second.d_tiles) t->retired.push_back(kv.second.d_tiles); if (kv.second.d_sample) t->retired.push_back(kv.second.d_sample); }
    t->tilesets.clear();
    for (auto &kv : t->key_images) if (kv.second.d) t->retired.push_back(kv.second.d);
    t->key_images.clear();
  }
  for (auto &kv : t->cols) {
    kv.second.info.rows = t->total_rows;
    if (kv.second.info.dtype != LLKV_DT_UTF8 && (rc = column_stats_device(*t, kv.second))) return rc;
  }
  t->generation++;
  return LLKV_OK;
}

llkv_status llkv_hip_table_append_chunks(llkv_hip_table *table, const uint64_t *chunk_rows, uint32_t n_new,
                                         const llkv_column_chunks *columns, uint32_t n_columns, const uint64_t *const *chunk_row_ids) {
  return (llkv_status)append_chunks_impl(reinterpret_cast<Table *>(table), chunk_rows, n_new, columns, n_columns, chunk_row_ids);
}

llkv_status llkv_hip_table_key_images(const llkv_hip_table *table, uint32_t *n_images, uint64_t *device_bytes) {
  if (!table) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL table");
  Table *t = const_cast<Table *>(reinterpret_cast<const Table *>(table));
  std::lock_guard<std::mutex> lk(t->mu);
  if (n_images) *n_images = (uint32_t)t->key_images.size();
  if (device_bytes) *device_bytes = t->key_images.size() * (t->dev_rows + kSlackRows) * 4;
  return LLKV_OK;
}
uint64_t llkv_hip_table_generation(const llkv_hip_table *table) { return table ? reinterpret_cast<const Table *>(table)->generation : 0; }

llkv_status llkv_hip_table_adopt_device_column(llkv_hip_table *table, uint32_t field_id, int32_t dtype,
                                               const void *device_values) {
  Table *t = reinterpret_cast<Table *>(table);
  int rc = check_new_column(t, field_id, t ? t->n_local_chunks : 0);
  if (rc) return (llkv_status)rc;
  if ((rc = ensure_device())) return (llkv_status)rc;
  const uint32_t w = dtype_width(dtype);
  if (w == 0 || dtype == LLKV_DT_UTF8) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "adopt_device_column: unsupported dtype");
  if (t->dev_rows != t->local_rows)
    return (llkv_status)set_error(LLKV_UNSUPPORTED, "adopting device buffers needs chunk row counts that are multiples of 16");
  DeviceColumn c;
  c.info.field_id = field_id;
  c.info.dtype = dtype;
  c.info.rows = t->total_rows;
  c.owned = true;
  // the adopted buffer has no slack past its end; keep an owned image with slack instead
  if ((rc = alloc_column(*t, w, &c.d_values))) return (llkv_status)rc;
  if (hipMemcpyAsync(c.d_values, device_values, t->dev_rows * w, hipMemcpyDeviceToDevice, g_ctx.stream) != hipSuccess ||
      hipStreamSynchronize(g_ctx.stream) != hipSuccess)
    return (llkv_status)set_error(LLKV_INTERNAL, "device copy failed");
  if ((rc = column_stats_device(*t, c))) return (llkv_status)rc;
  t->cols.emplace(field_id, std::move(c));
  return LLKV_OK;
}

} // extern "C"
