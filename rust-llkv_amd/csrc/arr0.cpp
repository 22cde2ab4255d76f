// arr0.cpp — reader for llkv-column-map's on-disk chunk format (`ARR0`,
// llkv-column-map/src/serialization.rs:41-140,438-488) and the dense-row-id detector
// (store/scan/filter.rs:1510-1582): lets the GPU path ingest real column chunks instead of
// synthetic buffers.  Host-only code.
#include "engine.hpp"

#include <cstring>
#include <vector>

namespace llkv {

static uint32_t rd_u32(const uint8_t *p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
static uint64_t rd_u64(const uint8_t *p) { uint64_t v; std::memcpy(&v, p, 8); return v; }

// PrimType → llkv_dtype (serialization.rs:146-166)
static int32_t dtype_of_prim(int32_t code) {
  switch (code) {
  case 1: return LLKV_DT_UINT64;
  case 2: return LLKV_DT_INT32;
  case 3: return LLKV_DT_UINT32;
  case 4: return LLKV_DT_FLOAT32;
  case 6: return LLKV_DT_INT64;
  case 11: return LLKV_DT_FLOAT64;
  case 12: return LLKV_DT_UTF8;
  case 15: return LLKV_DT_BOOLEAN;    // bit-packed values buffer
  case 16: return LLKV_DT_DATE32;
  case 18: return LLKV_DT_DECIMAL128; // precision / scale in header bytes 6 / 7 (serialize_primitive, serialization.rs:282-296)
  default: return -1; // Binary, Int16 / Int8 / UInt16 / UInt8 and Date64 (no storage type of theirs on this path), views
  }
}

int arr0_describe(const uint8_t *blob, uint64_t blob_len, llkv_arr0_desc *out) {
  if (!blob || !out) return set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  if (blob_len < 24 || std::memcmp(blob, "ARR0", 4) != 0) return set_error(LLKV_INTERNAL, "bad array blob magic/size");
  std::memset(out, 0, sizeof *out);
  out->layout = blob[4];
  out->type_code = blob[5];
  out->len = rd_u64(blob + 8);
  const uint64_t extra_a = rd_u32(blob + 16), extra_b = rd_u32(blob + 20);
  out->payload_offset = 24;
  const uint64_t payload_len = blob_len - 24;
  out->dtype = -1;
  if (out->layout == 0) { // Primitive: extra_a = values_len
    if (payload_len != extra_a) return set_error(LLKV_INTERNAL, "primitive payload length mismatch");
    out->values_offset = 24;
    out->values_len = extra_a;
    out->dtype = dtype_of_prim(out->type_code);
    if (out->dtype == LLKV_DT_BOOLEAN) { // an arrow bit buffer: at least ⌈len / 8⌉ bytes
      if (extra_a < (out->len + 7) / 8) return set_error(LLKV_INTERNAL, "boolean values buffer shorter than the element count");
    } else if (out->dtype == LLKV_DT_DECIMAL128) {
      if (16 * out->len != extra_a) return set_error(LLKV_INTERNAL, "primitive values length does not match the element count");
    } else if (out->dtype >= 0 && out->dtype != LLKV_DT_UTF8 && dtype_width(out->dtype) * out->len != extra_a)
      return set_error(LLKV_INTERNAL, "primitive values length does not match the element count");
    if (out->dtype == LLKV_DT_UTF8) out->dtype = -1;
  } else if (out->layout == 2) { // Varlen: [offsets][values]
    if (payload_len != extra_a + extra_b) return set_error(LLKV_INTERNAL, "varlen payload length mismatch");
    out->offsets_len = extra_a;
    out->values_offset = 24 + extra_a;
    out->values_len = extra_b;
    if (out->type_code == 12) { // Utf8: i32 offsets
      if (extra_a != (out->len + 1) * 4) return set_error(LLKV_INTERNAL, "utf8 offsets length mismatch");
      out->dtype = LLKV_DT_UTF8;
    }
  }
  return LLKV_OK;
}

} // namespace llkv

using namespace llkv;

extern "C" {

llkv_status llkv_hip_arr0_describe(const uint8_t *blob, uint64_t blob_len, llkv_arr0_desc *out) {
  return (llkv_status)arr0_describe(blob, blob_len, out);
}

llkv_status llkv_hip_dense_row_runs(const llkv_chunk_meta *c, uint32_t n, int32_t *is_dense, uint64_t *first_row_id) {
  if ((!c && n) || !is_dense) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  *is_dense = 1;
  if (first_row_id) *first_row_id = 0;
  bool have = false;
  uint64_t expected = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (c[i].row_count == 0) continue;
    const uint64_t start = c[i].min_val_u64, end = c[i].max_val_u64;
    if (end < start) { *is_dense = 0; return LLKV_OK; }
    const uint64_t delta = end - start;
    if (delta == UINT64_MAX) return (llkv_status)set_error(LLKV_INTERNAL, "row_id span overflow in dense_row_runs");
    if (delta + 1 != c[i].row_count) { *is_dense = 0; return LLKV_OK; }
    if (have && start != expected) { *is_dense = 0; return LLKV_OK; }
    if (!have && first_row_id) *first_row_id = start;
    have = true;
    expected = end + 1;
  }
  return LLKV_OK;
}

llkv_status llkv_hip_table_append_arr0_column(llkv_hip_table *table, uint32_t field_id, const uint8_t *const *blobs,
                                              const uint64_t *blob_lens, uint32_t n_chunks, const char *const *dictionary,
                                              uint32_t dict_size) {
  Table *t = reinterpret_cast<Table *>(table);
  if (!t) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "table is NULL");
  if (n_chunks != t->n_local_chunks)
    return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "expected " + std::to_string(t->n_local_chunks) + " local chunks, got " + std::to_string(n_chunks));
  std::vector<const void *> values(n_chunks);
  std::vector<const int32_t *> offsets(n_chunks);
  std::vector<const uint8_t *> data(n_chunks);
  std::vector<std::vector<uint8_t>> unpacked; // Boolean: one byte per value, as the staging entry point takes them
  int32_t dtype = -2, precision = 0, scale = 0;
  for (uint32_t i = 0; i < n_chunks; ++i) {
    llkv_arr0_desc d;
    int rc = arr0_describe(blobs[i], blob_lens[i], &d);
    if (rc) return (llkv_status)rc;
    if (d.dtype < 0) return (llkv_status)set_error(LLKV_UNSUPPORTED, "ARR0 layout/type " + std::to_string(d.layout) + "/" + std::to_string(d.type_code) + " is not on the GPU path");
    if (dtype == -2) dtype = d.dtype;
    if (d.dtype != dtype) return (llkv_status)set_error(LLKV_INTERNAL, "chunks of one column disagree on the type");
    if (d.len != t->global_chunk_rows[t->first_chunk + i])
      return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "chunk " + std::to_string(i) + " holds " + std::to_string(d.len) + " values, the chunk list says " +
                                                               std::to_string(t->global_chunk_rows[t->first_chunk + i]));
    values[i] = blobs[i] + d.values_offset;
    offsets[i] = reinterpret_cast<const int32_t *>(blobs[i] + d.payload_offset);
    data[i] = blobs[i] + d.values_offset;
    if (d.dtype == LLKV_DT_BOOLEAN) {
      unpacked.emplace_back((size_t)(d.len ? d.len : 1), 0);
      const uint8_t *bits = blobs[i] + d.values_offset;
      for (uint64_t r = 0; r < d.len; ++r) unpacked.back()[(size_t)r] = (bits[r >> 3] >> (r & 7)) & 1u;
    } else if (d.dtype == LLKV_DT_DECIMAL128) {
      const int32_t pr = blobs[i][6], sc = (int8_t)blobs[i][7];
      if (i && (pr != precision || sc != scale)) return (llkv_status)set_error(LLKV_INTERNAL, "chunks of one Decimal128 column disagree on precision / scale");
      precision = pr;
      scale = sc;
    }
  }
  if (dtype == LLKV_DT_BOOLEAN) for (uint32_t i = 0; i < n_chunks; ++i) values[i] = unpacked[i].data(); // (after the loop: the vector of vectors no longer moves)
  if (n_chunks == 0) return (llkv_status)set_error(LLKV_INVALID_ARGUMENT, "no chunks: the column type is unknown");
  if (dtype == LLKV_DT_UTF8) return llkv_hip_table_append_utf8_column(table, field_id, offsets.data(), data.data(), n_chunks, dictionary, dict_size);
  if (dtype == LLKV_DT_DECIMAL128) return llkv_hip_table_append_decimal128_column(table, field_id, precision, scale, values.data(), n_chunks);
  return llkv_hip_table_append_column(table, field_id, dtype, values.data(), n_chunks);
}

} // extern "C"
