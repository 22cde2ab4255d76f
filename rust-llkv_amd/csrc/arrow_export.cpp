// arrow_export.cpp — a scan batch as an Arrow RecordBatch through the Arrow C Data Interface.
//
// The reference hands batches to `on_batch` BY VALUE (Arc-backed Arrow buffers the callee may keep,
// llkv-executor/src/types/storage.rs:20-50); the views of llkv_hip_scan_stream live only during the callback.
// This export copies one view into buffers the consumer owns and releases through the standard callbacks — what a
// binding hands to arrow-rs (`arrow::ffi::from_ffi`), pyarrow (`RecordBatch._import_from_c`) or any other consumer.
// Output schema as the reference's: every field nullable (llkv-scan/src/execute.rs:166-181); Utf8 columns are
// materialised from their dictionary codes.
#include "../../include/llkv_hip.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace llkv {
int set_error(int code, const std::string &msg);
uint64_t table_chunk_rows(const llkv_hip_table *table, uint32_t global_chunk); // table.cpp
}

namespace {

struct Owned { // private_data of one exported array / schema
  std::vector<void *> blocks;            // malloc'd buffers
  std::vector<const void *> buffer_ptrs; // ArrowArray.buffers
  std::vector<ArrowArray *> child_arrays;
  std::vector<ArrowSchema *> child_schemas;
  std::string format, name;
  ~Owned() { for (void *b : blocks) std::free(b); }
  void *take(size_t bytes) {
    void *p = std::malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
    blocks.push_back(p);
    return p;
  }
};

void release_array(ArrowArray *a) {
  if (!a || !a->release) return;
  Owned *o = static_cast<Owned *>(a->private_data);
  for (ArrowArray *c : o->child_arrays) {
    if (c->release) c->release(c); // a child the consumer moved out has release == NULL and lives on
    delete c;
  }
  delete o;
  a->release = nullptr;
}
void release_schema(ArrowSchema *s) {
  if (!s || !s->release) return;
  Owned *o = static_cast<Owned *>(s->private_data);
  for (ArrowSchema *c : o->child_schemas) {
    if (c->release) c->release(c);
    delete c;
  }
  delete o;
  s->release = nullptr;
}

void init_schema(ArrowSchema *s, Owned *o, const std::string &format, const std::string &name) {
  std::memset(s, 0, sizeof *s);
  o->format = format;
  o->name = name;
  s->format = o->format.c_str();
  s->name = o->name.c_str();
  s->flags = 2; // ARROW_FLAG_NULLABLE
  s->release = release_schema;
  s->private_data = o;
}

// validity bitmap copy + null count (Arrow: LSB first, 1 = valid)
const void *copy_validity(Owned *o, const uint8_t *bits, uint64_t n, int64_t *null_count) {
  *null_count = 0;
  if (!bits) return nullptr;
  const size_t bytes = (size_t)((n + 7) / 8);
  uint8_t *dst = static_cast<uint8_t *>(o->take(bytes));
  std::memcpy(dst, bits, bytes);
  uint64_t valid = 0;
  for (uint64_t i = 0; i < n; ++i) valid += (bits[i >> 3] >> (i & 7)) & 1u;
  *null_count = (int64_t)(n - valid);
  return dst;
}

void export_column(const llkv_column_view &c, uint64_t n, const std::string &name, ArrowArray *arr, ArrowSchema *sch) {
  Owned *oa = new Owned(), *os = new Owned();
  std::memset(arr, 0, sizeof *arr);
  arr->length = (int64_t)n;
  arr->release = release_array;
  arr->private_data = oa;
  int64_t nulls = 0;
  const void *validity = copy_validity(oa, c.validity, n, &nulls);
  arr->null_count = nulls;
  std::string format;
  auto fixed = [&](const char *fmt, size_t width) {
    format = fmt;
    void *v = oa->take((size_t)n * width);
    if (n) std::memcpy(v, c.values, (size_t)n * width);
    oa->buffer_ptrs = {validity, v};
  };
  switch (c.dtype) {
  case LLKV_DT_INT64: fixed("l", 8); break;
  case LLKV_DT_UINT64: fixed("L", 8); break;
  case LLKV_DT_FLOAT64: fixed("g", 8); break;
  case LLKV_DT_INT32: fixed("i", 4); break;
  case LLKV_DT_UINT32: fixed("I", 4); break;
  case LLKV_DT_FLOAT32: fixed("f", 4); break;
  case LLKV_DT_DATE32: fixed("tdD", 4); break;
  case LLKV_DT_DECIMAL128: fixed("", 16); format = "d:" + std::to_string(c.precision) + "," + std::to_string(c.scale); break;
  case LLKV_DT_BOOLEAN: { // one byte per value here, bit-packed in Arrow
    format = "b";
    uint8_t *bits = static_cast<uint8_t *>(oa->take((size_t)((n + 7) / 8)));
    std::memset(bits, 0, (size_t)((n + 7) / 8));
    const uint8_t *v = static_cast<const uint8_t *>(c.values);
    for (uint64_t i = 0; i < n; ++i) if (v[i]) bits[i >> 3] |= (uint8_t)(1u << (i & 7));
    oa->buffer_ptrs = {validity, bits};
    break;
  }
  case LLKV_DT_UTF8: { // dictionary codes → offsets + data
    format = "u";
    const uint8_t *codes = static_cast<const uint8_t *>(c.values);
    size_t lens[256];
    bool seen[256] = {false};
    int32_t *off = static_cast<int32_t *>(oa->take((size_t)(n + 1) * 4));
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) {
      const uint8_t k = codes[i];
      if (!seen[k]) { seen[k] = true; lens[k] = c.dictionary && c.dictionary[k] ? std::strlen(c.dictionary[k]) : 0; }
      off[i] = (int32_t)total;
      const bool valid = !c.validity || ((c.validity[i >> 3] >> (i & 7)) & 1u);
      total += valid ? lens[k] : 0;
    }
    off[n] = (int32_t)total;
    char *data = static_cast<char *>(oa->take((size_t)total));
    for (uint64_t i = 0; i < n; ++i) {
      const size_t len = (size_t)(off[i + 1] - off[i]);
      if (len) std::memcpy(data + off[i], c.dictionary[codes[i]], len);
    }
    oa->buffer_ptrs = {validity, off, data};
    break;
  }
  default: format = "n"; arr->null_count = (int64_t)n; oa->buffer_ptrs = {}; break; // Null type
  }
  arr->n_buffers = (int64_t)oa->buffer_ptrs.size();
  arr->buffers = oa->buffer_ptrs.empty() ? nullptr : oa->buffer_ptrs.data();
  init_schema(sch, os, format, name);
}

} // namespace

extern "C" llkv_status llkv_hip_batch_export_arrow(const llkv_batch_view *batch, const char *const *column_names,
                                                   struct ArrowArray *out_array, struct ArrowSchema *out_schema) {
  if (!batch || !out_array || !out_schema) return (llkv_status)llkv::set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  if (batch->num_rows > (uint64_t)INT32_MAX) return (llkv_status)llkv::set_error(LLKV_UNSUPPORTED, "batch too large for 32-bit Arrow offsets");
  try {
    const uint32_t n_children = batch->num_columns + (batch->row_ids ? 1u : 0u);
    Owned *oa = new Owned(), *os = new Owned();
    std::memset(out_array, 0, sizeof *out_array);
    out_array->length = (int64_t)batch->num_rows;
    out_array->release = release_array;
    out_array->private_data = oa;
    oa->buffer_ptrs = {nullptr}; // a struct array has one (validity) buffer
    out_array->n_buffers = 1;
    out_array->buffers = oa->buffer_ptrs.data();
    init_schema(out_schema, os, "+s", "");
    out_schema->flags = 0;
    for (uint32_t i = 0; i < n_children; ++i) {
      ArrowArray *ca = new ArrowArray();
      ArrowSchema *cs = new ArrowSchema();
      oa->child_arrays.push_back(ca);
      os->child_schemas.push_back(cs);
      if (i < batch->num_columns) {
        const std::string name = column_names && column_names[i] ? column_names[i] : "c" + std::to_string(i);
        export_column(batch->columns[i], batch->num_rows, name, ca, cs);
      } else { // include_row_ids: one more, never-NULL UInt64 column (llkv-scan/src/execute.rs: ROW_ID_COLUMN_NAME)
        llkv_column_view rid;
        std::memset(&rid, 0, sizeof rid);
        rid.dtype = LLKV_DT_UINT64;
        rid.values = batch->row_ids;
        export_column(rid, batch->num_rows, "rowid", ca, cs);
        cs->flags = 0;
      }
    }
    out_array->n_children = n_children;
    out_array->children = oa->child_arrays.data();
    out_schema->n_children = n_children;
    out_schema->children = os->child_schemas.data();
  } catch (const std::bad_alloc &) {
    return (llkv_status)llkv::set_error(LLKV_INTERNAL, "out of memory exporting a batch");
  }
  return LLKV_OK;
}

// ---- staging from Arrow arrays --------------------------------------------------------------------------------
// One ArrowArray per local chunk (what the reference's chunks are after `deserialize_array`,
// llkv-column-map/src/serialization.rs:438-488): the value / offset / validity buffers are handed to the staging
// entry points as they lie (bitmaps and Boolean bits that start at a non-zero offset are re-packed on the host).
namespace {
int dtype_of_format(const char *f, int32_t *precision, int32_t *scale) {
  if (!f) return -1;
  if (!std::strcmp(f, "l")) return LLKV_DT_INT64;
  if (!std::strcmp(f, "L")) return LLKV_DT_UINT64;
  if (!std::strcmp(f, "g")) return LLKV_DT_FLOAT64;
  if (!std::strcmp(f, "f")) return LLKV_DT_FLOAT32;
  if (!std::strcmp(f, "i")) return LLKV_DT_INT32;
  if (!std::strcmp(f, "I")) return LLKV_DT_UINT32;
  if (!std::strcmp(f, "tdD")) return LLKV_DT_DATE32;
  if (!std::strcmp(f, "b")) return LLKV_DT_BOOLEAN;
  if (!std::strcmp(f, "u")) return LLKV_DT_UTF8;
  if (!std::strncmp(f, "d:", 2)) {
    int p = 0, s = 0, bits = 128;
    const int got = std::sscanf(f + 2, "%d,%d,%d", &p, &s, &bits);
    if (got < 2 || bits != 128) return -1;
    *precision = p;
    *scale = s;
    return LLKV_DT_DECIMAL128;
  }
  return -1;
}
size_t width_of(int dt) {
  switch (dt) {
  case LLKV_DT_INT64: case LLKV_DT_UINT64: case LLKV_DT_FLOAT64: return 8;
  case LLKV_DT_INT32: case LLKV_DT_UINT32: case LLKV_DT_FLOAT32: case LLKV_DT_DATE32: return 4;
  case LLKV_DT_DECIMAL128: return 16;
  default: return 0;
  }
}
bool bit_at(const uint8_t *bits, int64_t i) { return (bits[i >> 3] >> (i & 7)) & 1u; }
} // namespace

extern "C" llkv_status llkv_hip_table_append_arrow_column(llkv_hip_table *table, uint32_t field_id, const struct ArrowSchema *schema,
                                                          const struct ArrowArray *const *chunks, uint32_t n_chunks,
                                                          const char *const *dictionary, uint32_t dict_size) {
  if (!table || !schema || (n_chunks && !chunks)) return (llkv_status)llkv::set_error(LLKV_INVALID_ARGUMENT, "NULL argument");
  int32_t precision = 0, scale = 0;
  const int dt = dtype_of_format(schema->format, &precision, &scale);
  if (dt < 0) return (llkv_status)llkv::set_error(LLKV_UNSUPPORTED, std::string("Arrow format '") + (schema->format ? schema->format : "") + "' is not staged on the GPU path");
  std::vector<const void *> values(n_chunks, nullptr);
  std::vector<const int32_t *> offsets(n_chunks, nullptr);
  std::vector<const uint8_t *> data(n_chunks, nullptr), validity(n_chunks, nullptr);
  std::vector<std::vector<uint8_t>> repacked; // Boolean bytes, shifted bitmaps
  repacked.reserve((size_t)n_chunks * 2);
  bool any_nulls = false;
  // the staging entry points read `local chunk rows` elements from every buffer: the arrays must hold exactly those
  // (the ARR0 path checks the same, arr0.cpp), their buffers must exist, and Utf8 offsets must be usable as they lie
  uint32_t first_chunk = 0, n_local = 0;
  if (llkv_hip_table_local_chunks(table, &first_chunk, &n_local) != LLKV_OK) return LLKV_INVALID_ARGUMENT;
  if (n_chunks != n_local)
    return (llkv_status)llkv::set_error(LLKV_INVALID_ARGUMENT, "expected " + std::to_string(n_local) + " local chunks, got " + std::to_string(n_chunks));
  for (uint32_t i = 0; i < n_chunks; ++i) {
    const ArrowArray *a = chunks[i];
    if (!a || a->n_buffers < (dt == LLKV_DT_UTF8 ? 3 : 2) || !a->buffers) return (llkv_status)llkv::set_error(LLKV_INVALID_ARGUMENT, "Arrow chunk without its buffers");
    const int64_t off = a->offset, n = a->length;
    const uint64_t want_rows = llkv::table_chunk_rows(table, first_chunk + i);
    if (off < 0 || n < 0 || (uint64_t)n != want_rows)
      return (llkv_status)llkv::set_error(LLKV_INVALID_ARGUMENT, "Arrow chunk " + std::to_string(i) + " holds " + std::to_string(n) + " rows at offset " + std::to_string(off) +
                                                                     ", the table's chunk has " + std::to_string(want_rows));
    if (n > 0 && (!a->buffers[1] || (dt == LLKV_DT_UTF8 && !a->buffers[2] && static_cast<const int32_t *>(a->buffers[1])[off + n] != static_cast<const int32_t *>(a->buffers[1])[off])))
      return (llkv_status)llkv::set_error(LLKV_INVALID_ARGUMENT, "Arrow chunk " + std::to_string(i) + " has a NULL values / offsets / data buffer");
    if (dt == LLKV_DT_UTF8 && n > 0) { // offsets: non-negative and monotone, so no string has a negative length
      const int32_t *o = static_cast<const int32_t *>(a->buffers[1]) + off;
      if (o[0] < 0) return (llkv_status)llkv::set_error(LLKV_INVALID_ARGUMENT, "Arrow Utf8 chunk " + std::to_string(i) + " has a negative offset");
      for (int64_t r = 0; r < n; ++r)
        if (o[r + 1] < o[r]) return (llkv_status)llkv::set_error(LLKV_INVALID_ARGUMENT, "Arrow Utf8 chunk " + std::to_string(i) + " has descending offsets");
    }
    const uint8_t *bits = static_cast<const uint8_t *>(a->buffers[0]);
    if (bits && a->null_count != 0) {
      any_nulls = true;
      if (off % 8 == 0) validity[i] = bits + off / 8;
      else {
        repacked.emplace_back((size_t)((n + 7) / 8), 0);
        for (int64_t r = 0; r < n; ++r) if (bit_at(bits, off + r)) repacked.back()[(size_t)(r >> 3)] |= (uint8_t)(1u << (r & 7));
        validity[i] = repacked.back().data();
      }
    }
    if (dt == LLKV_DT_UTF8) {
      offsets[i] = static_cast<const int32_t *>(a->buffers[1]) + off;
      data[i] = static_cast<const uint8_t *>(a->buffers[2]);
    } else if (dt == LLKV_DT_BOOLEAN) { // bit-packed in Arrow, one byte per value in HBM
      repacked.emplace_back((size_t)(n ? n : 1), 0);
      const uint8_t *vb = static_cast<const uint8_t *>(a->buffers[1]);
      for (int64_t r = 0; r < n; ++r) repacked.back()[(size_t)r] = vb && bit_at(vb, off + r) ? 1 : 0;
      values[i] = repacked.back().data();
    } else {
      values[i] = static_cast<const char *>(a->buffers[1]) + (size_t)off * width_of(dt);
    }
  }
  llkv_status rc;
  if (dt == LLKV_DT_UTF8) rc = llkv_hip_table_append_utf8_column(table, field_id, offsets.data(), data.data(), n_chunks, dictionary, dict_size);
  else if (dt == LLKV_DT_DECIMAL128) rc = llkv_hip_table_append_decimal128_column(table, field_id, precision, scale, values.data(), n_chunks);
  else rc = llkv_hip_table_append_column(table, field_id, dt, values.data(), n_chunks);
  if (rc != LLKV_OK || !any_nulls) return rc;
  return llkv_hip_table_set_column_validity(table, field_id, validity.data(), n_chunks);
}
