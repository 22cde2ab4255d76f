/*
 * llkv_oracle.h — CPU restatement of the reference's scan → filter → (join) →
 * aggregate path.  TEST INFRASTRUCTURE ONLY: nothing outside tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or load this.
 *
 * Parity status: pinned against the known-answer tests the reference holds for the
 * path (transcribed under tests/golden/).  TPC-H answer-set parity
 * is UNPINNED: the reference (Rust, 394 crates) cannot be built offline and the TPC
 * answer sets are not in the tree.
 *
 * The plan vocabulary (literals, filters, predicate program, postfix expressions,
 * aggregate specs) is shared with include/llkv_hip.h so both sides are driven by
 * byte-identical plan descriptions.
 */
#ifndef LLKV_ORACLE_H
#define LLKV_ORACLE_H

#include "llkv_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One column of a table.  Row ids are dense 0..rows-1; a NULL cell is a row id that
 * is absent from the column (llkv-column-map/src/serialization.rs:47-53), modelled
 * here as a cleared bit in `validity` (LSB-first, NULL pointer = all present).    */
typedef struct orc_column {
  uint32_t field_id;
  int32_t dtype;           /* llkv_dtype                                          */
  const void *values;      /* fixed width values; LLKV_DT_UTF8: unused            */
  const uint8_t *validity;
  const int32_t *offsets;  /* LLKV_DT_UTF8: Arrow offsets (rows+1)                */
  const uint8_t *data;     /* LLKV_DT_UTF8: Arrow data                            */
  int32_t precision, scale; /* LLKV_DT_DECIMAL128: values are 16-byte little-endian i128 */
} orc_column;

typedef struct orc_table {
  uint64_t rows;
  uint32_t n_cols;
  const orc_column *cols;
} orc_table;

const char *orc_last_error(void);

/* llkv-scan/src/predicate.rs:32-193 — stack VM over row-id sets. */
int32_t orc_filter_row_ids(const orc_table *t, const llkv_filter *filters, uint32_t n_filters,
                           const llkv_eval_op *ops, uint32_t n_ops, uint64_t **out_ids,
                           uint64_t *out_len);

/* llkv-scan/src/execute.rs:47-295 + row_stream.rs:369-623: filter → 65 536-row
 * windows → gather → computed projections → callback.                            */
typedef struct orc_batch_column {
  int32_t dtype;
  const void *values;      /* i64 / f64 / i32 / u64 / f32 / u32 values; Utf8: NULL */
  const uint8_t *valid;    /* one byte per row, 1 = valid                          */
  const char *const *strings; /* Utf8: per-row NUL-terminated copies              */
  int32_t precision, scale;   /* Decimal128 (16-byte values)                       */
} orc_batch_column;

typedef struct orc_batch {
  uint64_t num_rows;
  uint32_t num_columns;
  const orc_batch_column *columns;
  const uint64_t *row_ids;
} orc_batch;

typedef void (*orc_on_batch)(const orc_batch *batch, void *user);

int32_t orc_scan_stream(const orc_table *t, const llkv_projection *projections,
                        uint32_t n_projections, const llkv_filter *filters, uint32_t n_filters,
                        const llkv_eval_op *ops, uint32_t n_ops, const llkv_scan_options *options,
                        orc_on_batch on_batch, void *user);

/* llkv-executor/src/lib.rs:5357-5682 / :6087-6665 over llkv-aggregate accumulators. */
int32_t orc_aggregate(const orc_table *t, const llkv_filter *filters, uint32_t n_filters,
                      const llkv_eval_op *ops, uint32_t n_ops, const llkv_aggregate_spec *aggs,
                      uint32_t n_aggs, llkv_value *out_values);

/* llkv-executor/src/lib.rs:4405-4542, :5028-5355 (hash group-by, PlanValue row
 * interpreter for non-column aggregate arguments).                               */
typedef struct orc_groups orc_groups;
int32_t orc_groupby(const orc_table *t, const llkv_filter *filters, uint32_t n_filters,
                    const llkv_eval_op *ops, uint32_t n_ops, const uint32_t *key_fields,
                    uint32_t n_keys, const llkv_aggregate_spec *aggs, uint32_t n_aggs,
                    int32_t order_by_keys, orc_groups **out);
uint32_t orc_groups_len(const orc_groups *g);
int32_t orc_groups_key(const orc_groups *g, uint32_t group, uint32_t key, llkv_value *out);
int32_t orc_groups_value(const orc_groups *g, uint32_t group, uint32_t agg, llkv_value *out);
void orc_groups_free(orc_groups *g);

/* llkv-join/src/hash_join.rs:955-1417 (integer fast path) — index pairs. */
typedef void (*orc_on_join_batch)(const uint64_t *left_rows, const uint64_t *right_rows,
                                  uint64_t n_pairs, void *user);
int32_t orc_hash_join(const orc_table *left, const orc_table *right, const llkv_join_key *keys,
                      uint32_t n_keys, const llkv_join_options *options,
                      orc_on_join_batch on_batch, void *user);

/* … and the joined RecordBatches (emit :715-772, output schema :877-943, cross product :1500-1599, the executor's
 * hash_join_table_batches llkv-executor/src/lib.rs:12218-12392): left user columns, right user columns. */
typedef void (*orc_on_join_record_batch)(const orc_batch *batch, const char *const *column_names, void *user);
int32_t orc_hash_join_batches(const orc_table *left, const orc_table *right, const llkv_join_key *keys,
                              uint32_t n_keys, const llkv_join_options *options, const llkv_join_output *output,
                              orc_on_join_record_batch on_batch, void *user);

/* Chunk-parallel fused variants of the same arithmetic (BASELINE.md §2 mode 2,
 * "best-effort parallel"): per-chunk partial sums combined in chunk order.       */
int32_t orc_aggregate_parallel(const orc_table *t, const llkv_filter *filters, uint32_t n_filters,
                               const llkv_aggregate_spec *aggs, uint32_t n_aggs,
                               llkv_value *out_values, int32_t threads);
/* … grouped by up to two one-character Utf8 columns (TPC-H Q1's flags); groups in key-byte order:
 * out_keys[2g], out_keys[2g+1], out_values[g][n_aggs]. */
int32_t orc_groupby_parallel(const orc_table *t, const llkv_filter *filters, uint32_t n_filters,
                             const uint32_t *key_fields, uint32_t n_keys, const llkv_aggregate_spec *aggs,
                             uint32_t n_aggs, llkv_value *out_values, uint8_t *out_keys, uint32_t *out_groups,
                             uint32_t max_groups, int32_t threads);
/* STREAM-like triad over `threads` threads, GB/s (best of `reps`): the host's memory bandwidth, for context. */
double orc_stream_triad(uint64_t n, int32_t threads, int32_t reps);

void orc_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
