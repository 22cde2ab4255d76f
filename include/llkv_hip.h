/*
 * llkv_hip.h — C ABI of the MI355X-native execution path for LLKV.
 *
 * This is the drop-in boundary: the entry points a Rust `extern "C"` shim inside
 * `llkv-executor` would bind in place of the CPU scan → filter → (join) → aggregate
 * pipeline.  Plain pointers and sizes only; no C++/torch types cross it.
 *
 * Each declaration cites the reference interface (relative to the reference tree)
 * whose behaviour it replaces.  Error behaviour follows `llkv-result/src/error.rs`:
 * every call returns a status code, the message is fetched with
 * `llkv_hip_last_error()` (thread-local), nothing unwinds across the boundary.
 *
 * Threads and streams.  Calls may come from any host thread (the traits the shim implements are Send + Sync,
 * llkv-executor/src/types/storage.rs:20-50).  The multi-kernel pipelines — scan_stream, filter_row_ids, the joins,
 * the join → GROUP BY → top-k pipeline, the sort-based and partitioned GROUP BY — run on ONE stream the library owns
 * and recycle their device temporaries in the order of that stream: a block handed back while a kernel still reads
 * it can only be handed to work queued behind that kernel.  Device memory that a CALLER's stream will touch (the
 * buffers of a prepared query: llkv_hip_query_launch / _all_reduce / _submit take a stream) is allocated by the
 * prepare call, which drains the library's stream before it returns — no such buffer aliases an in-flight temporary.
 * Small read-backs of one host thread share one pinned slab: a thread has one of them in flight at a time (the
 * library's own calls keep to that; a handle of the phased join pipeline must not be driven from two threads at once).
 */
#ifndef LLKV_HIP_H
#define LLKV_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LLKV_HIP_ABI_VERSION 1

/* ------------------------------------------------------------------------- */
/* Status codes — llkv-result/src/error.rs (InvalidArgumentError / Internal / */
/* NotFound); UNSUPPORTED tells the caller to keep its own CPU route.         */
/* ------------------------------------------------------------------------- */
typedef enum llkv_status {
  LLKV_OK = 0,
  LLKV_INVALID_ARGUMENT = 1, /* Error::InvalidArgumentError (user visible)    */
  LLKV_INTERNAL = 2,         /* Error::Internal                               */
  LLKV_NOT_FOUND = 3,        /* Error::NotFound (scan of a missing column)    */
  LLKV_UNSUPPORTED = 4,      /* plan shape not on the GPU path: fall back     */
  LLKV_NO_DEVICE = 5,        /* no usable HIP device / extension not built    */
  LLKV_PREDICATE_BUILD = 6   /* Error::PredicateBuild — literal cannot be cast
                                to the column's native type
                                (llkv-table/src/table.rs:1236)                 */
} llkv_status;

/* Column storage types the path accepts.  Mapping from SQL types:
 * llkv-sql/src/lib.rs:25-28 (INT→Int64, DOUBLE→Float64, DATE→Date32,
 * CHAR/VARCHAR→Utf8).  Filterable set: llkv-table/src/table.rs:1156-1168.    */
typedef enum llkv_dtype {
  LLKV_DT_NULL = 0,
  LLKV_DT_INT64 = 1,
  LLKV_DT_FLOAT64 = 2,
  LLKV_DT_INT32 = 3,
  LLKV_DT_DATE32 = 4,   /* i32 days since 1970-01-01                         */
  LLKV_DT_UINT64 = 5,
  LLKV_DT_UINT32 = 6,
  LLKV_DT_FLOAT32 = 7,
  LLKV_DT_UTF8 = 8,     /* staged as 1-byte dictionary codes in HBM          */
  LLKV_DT_BOOLEAN = 9,
  LLKV_DT_DECIMAL128 = 10 /* i128 raw value, little endian, with the column's (precision, scale) */
} llkv_dtype;

/* ------------------------------------------------------------------------- */
/* Literals — llkv-types/src/literal.rs (`Literal`), cast rules :364-520.     */
/* ------------------------------------------------------------------------- */
typedef enum llkv_literal_tag {
  LLKV_LIT_NULL = 0,
  LLKV_LIT_INT128 = 1,     /* lo/hi two's complement                         */
  LLKV_LIT_FLOAT64 = 2,
  LLKV_LIT_DECIMAL128 = 3, /* raw value in lo/hi, `scale`                    */
  LLKV_LIT_BOOLEAN = 4,    /* lo = 0/1                                       */
  LLKV_LIT_STRING = 5,     /* str (NUL terminated, borrowed)                 */
  LLKV_LIT_DATE32 = 6      /* lo = days                                      */
} llkv_literal_tag;

typedef struct llkv_literal {
  int32_t tag;   /* llkv_literal_tag */
  int32_t scale; /* Decimal128 scale */
  uint64_t lo;
  int64_t hi;
  double f64;
  const char *str;
} llkv_literal;

/* ------------------------------------------------------------------------- */
/* Leaf filters — llkv-expr `Filter{field_id, Operator}`, owned form           */
/* llkv-compute/src/program.rs:80-112; typed evaluation                       */
/* llkv-expr/src/typed_predicate.rs:75-146,253-312.                           */
/* ------------------------------------------------------------------------- */
typedef enum llkv_operator_kind {
  LLKV_OP_EQUALS = 1,
  LLKV_OP_RANGE = 2,
  LLKV_OP_GT = 3,
  LLKV_OP_GE = 4,
  LLKV_OP_LT = 5,
  LLKV_OP_LE = 6,
  LLKV_OP_IN = 7,
  LLKV_OP_IS_NULL = 8,
  LLKV_OP_IS_NOT_NULL = 9,
  /* MVCC visibility as a leaf: the row filter every SqlEngine SELECT applies after the
   * predicate (MvccRowIdFilter llkv-transaction/src/helpers.rs:259-312, rule RowVersion::is_visible_for
   * llkv-transaction/src/mvcc.rs:283-333).  field_id = the `created_by` column (UInt64),
   * value.lo = field id of the `deleted_by` column, lower.lo = snapshot.txn_id,
   * upper.lo = snapshot.snapshot_id, in_list = txn ids whose status is NOT Committed
   * (Active / Aborted; ≤ 32 on the GPU path).                                             */
  LLKV_OP_MVCC_VISIBLE = 10,
  /* Expr::Compare { left, op, right } over scalar expressions (the expr-vs-expr route,
   * evaluate_compare_rows llkv-scan/src/predicate.rs:562-663 → compute_compare
   * llkv-compute/src/kernels.rs:269-297): both sides are coerced to their common type and
   * compared with arrow's `cmp` kernels — for floats that is IEEE totalOrder (NaN above
   * everything, -0.0 below +0.0), unlike the leaf predicates' partial_cmp.  Uses cmp_*.   */
  LLKV_OP_COMPARE = 11,
  /* Expr::InList { expr, list, negated } (EvalOp::PushInList, evaluate_in_list_over_rows llkv-scan/src/
   * predicate.rs:443-560): over the rows where every referenced field is present, the target (cmp_left) is
   * coerced item by item to the common type with each list expression and compared with arrow `eq` (floats by
   * totalOrder); the item results are OR-ed, `negated` inverts.  Uses cmp_left, list_*, negated.          */
  LLKV_OP_IN_LIST = 12,
  /* Expr::IsNull { expr, negated } over a scalar expression (EvalOp::PushIsNull, collect_row_ids_for_is_null
   * predicate.rs:249-331): a bare column is the IS [NOT] NULL leaf; otherwise the rows — among those where at
   * least one referenced field is present — whose value is (not) NULL.  Uses cmp_left, negated.            */
  LLKV_OP_IS_NULL_EXPR = 13,
  /* Operator::{StartsWith, EndsWith, Contains}{pattern, case_sensitive} over Utf8 columns (llkv-expr/src/
   * typed_predicate.rs:186-210,439-458): `value` holds the pattern (String literal), `case_sensitive` the flag;
   * case-insensitive = both sides through to_lowercase (ASCII strings on the GPU path).                  */
  LLKV_OP_STARTS_WITH = 14,
  LLKV_OP_ENDS_WITH = 15,
  LLKV_OP_CONTAINS = 16
} llkv_operator_kind;

typedef enum llkv_compare_op { /* llkv_expr::CompareOp */
  LLKV_CMP_EQ = 1,
  LLKV_CMP_NOT_EQ = 2,
  LLKV_CMP_LT = 3,
  LLKV_CMP_LT_EQ = 4,
  LLKV_CMP_GT = 5,
  LLKV_CMP_GT_EQ = 6
} llkv_compare_op;

typedef enum llkv_bound_kind {
  LLKV_BOUND_UNBOUNDED = 0,
  LLKV_BOUND_INCLUDED = 1,
  LLKV_BOUND_EXCLUDED = 2
} llkv_bound_kind;

typedef struct llkv_filter {
  uint32_t field_id;
  int32_t op;           /* llkv_operator_kind                                */
  llkv_literal value;   /* EQUALS / GT / GE / LT / LE                        */
  int32_t lower_kind;   /* RANGE: llkv_bound_kind                            */
  llkv_literal lower;
  int32_t upper_kind;
  llkv_literal upper;
  const llkv_literal *in_list; /* IN                                         */
  uint32_t in_len;
  int32_t cmp_op;              /* LLKV_OP_COMPARE: llkv_compare_op              */
  const struct llkv_expr_token *cmp_left;
  uint32_t cmp_left_len;
  const struct llkv_expr_token *cmp_right;
  uint32_t cmp_right_len;
  const struct llkv_expr_token *const *list_exprs; /* LLKV_OP_IN_LIST: the list's expressions */
  const uint32_t *list_expr_lens;
  uint32_t list_len;
  int32_t negated;             /* LLKV_OP_IN_LIST / LLKV_OP_IS_NULL_EXPR                  */
  int32_t case_sensitive;      /* LLKV_OP_STARTS_WITH / ENDS_WITH / CONTAINS              */
} llkv_filter;

/* Predicate program — `EvalOp` stack program, llkv-compute/src/program.rs:48-78,
 * interpreted by llkv-scan/src/predicate.rs:32-193.  `n_ops == 0` means the
 * conjunction of all filters (`Expr::all_of`); no filters and no ops = TRUE.  */
typedef enum llkv_eval_opcode {
  LLKV_EVAL_PUSH_PREDICATE = 1, /* arg = index into filters[]                 */
  LLKV_EVAL_PUSH_LITERAL = 2,   /* arg = 0/1                                  */
  LLKV_EVAL_AND = 3,            /* arg = child_count                          */
  LLKV_EVAL_OR = 4,             /* arg = child_count                          */
  LLKV_EVAL_NOT = 5
} llkv_eval_opcode;

typedef struct llkv_eval_op {
  int32_t op;
  uint32_t arg;
} llkv_eval_op;

/* ------------------------------------------------------------------------- */
/* Scalar expressions in postfix form — the token program of                  */
/* llkv-compute/src/fast_numeric.rs:19-37 (`Token::{Column,Literal,Binary}`).  */
/* ------------------------------------------------------------------------- */
typedef enum llkv_token_kind {
  LLKV_TOK_COLUMN = 1,
  LLKV_TOK_LITERAL = 2,
  LLKV_TOK_BINARY = 3
} llkv_token_kind;

typedef enum llkv_binary_op {
  LLKV_BIN_ADD = 1,
  LLKV_BIN_SUB = 2,
  LLKV_BIN_MUL = 3,
  LLKV_BIN_DIV = 4,
  LLKV_BIN_MOD = 5
} llkv_binary_op;

typedef struct llkv_expr_token {
  int32_t kind;  /* llkv_token_kind                                         */
  int32_t binop; /* llkv_binary_op for LLKV_TOK_BINARY                       */
  uint32_t field_id;
  llkv_literal literal;
} llkv_expr_token;

/* ------------------------------------------------------------------------- */
/* Aggregates — `AggregateSpec{alias,kind}` llkv-aggregate/src/lib.rs:25-69;  */
/* the argument is a bare column or a computed projection                     */
/* (llkv-executor/src/lib.rs:470-501).                                        */
/* ------------------------------------------------------------------------- */
typedef enum llkv_aggregate_kind {
  LLKV_AGG_COUNT_STAR = 1,
  LLKV_AGG_COUNT = 2,
  LLKV_AGG_SUM = 3,
  LLKV_AGG_TOTAL = 4,
  LLKV_AGG_AVG = 5,
  LLKV_AGG_MIN = 6,
  LLKV_AGG_MAX = 7,
  LLKV_AGG_COUNT_NULLS = 8
} llkv_aggregate_kind;

typedef struct llkv_aggregate_spec {
  int32_t kind;     /* llkv_aggregate_kind                                   */
  int32_t distinct; /* DISTINCT forms of COUNT / SUM / TOTAL / AVG (MIN / MAX ignore it): ungrouped, and inside GROUP BY over ONE
                     * argument — a bare column or a computed Int64 / Float64 expression — on the sort-based route of an unsharded
                     * table; LLKV_UNSUPPORTED otherwise */
  const llkv_expr_token *expr; /* NULL for COUNT(*)                          */
  uint32_t expr_len;
  const char *alias;
} llkv_aggregate_spec;

/* One finalized aggregate cell = the 1-element Arrow array returned by
 * `AggregateAccumulator::finalize` llkv-aggregate/src/lib.rs:1488-1939.      */
typedef struct llkv_value {
  int32_t dtype;   /* LLKV_DT_INT64 / LLKV_DT_FLOAT64 / LLKV_DT_UTF8 / LLKV_DT_DECIMAL128 */
  int32_t is_null;
  int64_t i64;     /* Decimal128: low 64 bits of the raw value                */
  double f64;
  const char *str; /* group keys of Utf8 type; owned by the result object    */
  int64_t i64_hi;  /* Decimal128: high 64 bits of the raw value              */
  int32_t precision, scale; /* Decimal128(precision, scale)                  */
} llkv_value;

/* ------------------------------------------------------------------------- */
/* Device / context                                                           */
/* ------------------------------------------------------------------------- */
/* Bind the calling process to one GPU (one process per GPU).  Also makes the two copy lanes of the staging path and sends one
 * page-locked block through each: the first registration + DMA of a process costs ~23 ms whatever it copies, and the first table
 * staged should not pay it (LLKV_HIP_NO_STAGING_PRIME=1 leaves it to the first staging call). */
llkv_status llkv_hip_init(int32_t device_ordinal);
void llkv_hip_shutdown(void);
int32_t llkv_hip_device_count(void);
uint32_t llkv_hip_abi_version(void);
/* Thread-local message of the last failing call on this thread. */
const char *llkv_hip_last_error(void);

/* ------------------------------------------------------------------------- */
/* Tables — the HBM-resident image of a `Table`'s column chunks.               */
/* Replaces the chunk walk of `ColumnStore` (llkv-column-map/src/store/        */
/* scan/unsorted.rs:40-67); chunks are the sharding unit (131 072 rows per     */
/* 8-byte column, llkv-column-map/src/store/constants.rs:22).                  */
/* ------------------------------------------------------------------------- */
typedef struct llkv_hip_table llkv_hip_table;

/* `global_chunk_rows[n_global_chunks]` describes the whole table; this rank
 * stages only the chunks of its shard (see llkv_hip_table_local_chunks).
 * world == 1 stages everything.  world must divide 8 for results that are
 * bit-identical across GPU counts.                                           */
llkv_status llkv_hip_table_create(uint16_t table_id, const uint64_t *global_chunk_rows,
                                  uint32_t n_global_chunks, uint32_t rank, uint32_t world,
                                  llkv_hip_table **out);
void llkv_hip_table_free(llkv_hip_table *table);
/* Chunk range [first, first+count) of the global table this rank owns. */
llkv_status llkv_hip_table_local_chunks(const llkv_hip_table *table, uint32_t *first,
                                        uint32_t *count);
uint64_t llkv_hip_table_total_rows(const llkv_hip_table *table); /* global   */
uint64_t llkv_hip_table_local_rows(const llkv_hip_table *table);

/* Stage the value buffers of one fixed-width column: `chunk_values[i]` points at
 * the Arrow values buffer of local chunk i (host memory, `local chunk rows[i]`
 * elements).  Copies go pinned-host → HBM with hipMemcpyAsync.                */
llkv_status llkv_hip_table_append_column(llkv_hip_table *table, uint32_t field_id, int32_t dtype,
                                         const void *const *chunk_values, uint32_t n_chunks);
/* Utf8 column given as Arrow offsets(i32)+data per chunk; staged as 1-byte
 * dictionary codes (≤ 256 distinct values, else LLKV_UNSUPPORTED).
 * `dictionary` (dict_size strings) fixes the code of every value; it is REQUIRED
 * when the table is sharded (world > 1) so that all ranks agree on the codes —
 * e.g. the sorted union of the shards' distinct values.  With world == 1 it may
 * be NULL: codes are then assigned in first-appearance order.                  */
llkv_status llkv_hip_table_append_utf8_column(llkv_hip_table *table, uint32_t field_id,
                                              const int32_t *const *chunk_offsets,
                                              const uint8_t *const *chunk_data,
                                              uint32_t n_chunks,
                                              const char *const *dictionary, uint32_t dict_size);
/* --- llkv-column-map chunk format ------------------------- */
/* `ARR0` blob header, llkv-column-map/src/serialization.rs:41-140: 24 bytes
 * (magic, layout, type code, len, extra_a, extra_b) then the payload; no null
 * bitmaps exist on disk.                                                      */
typedef struct llkv_arr0_desc {
  int32_t layout;        /* 0 Primitive, 1 FslFloat32, 2 Varlen, 3 Struct       */
  int32_t type_code;     /* PrimType, serialization.rs:146-166                  */
  int32_t dtype;         /* llkv_dtype, or -1 when the path does not take it    */
  int32_t reserved;
  uint64_t len;          /* element count                                       */
  uint64_t payload_offset; /* = 24                                              */
  uint64_t values_offset;  /* Primitive: = payload; Varlen: after the offsets   */
  uint64_t values_len;
  uint64_t offsets_len;    /* Varlen only                                       */
} llkv_arr0_desc;
llkv_status llkv_hip_arr0_describe(const uint8_t *blob, uint64_t blob_len, llkv_arr0_desc *out);

/* Row-id shadow chunk metadata (ChunkMetadata, store/descriptor.rs:19-84) and the
 * density test of `dense_row_runs` (store/scan/filter.rs:1510-1582): every chunk
 * spans exactly row_count ids and chunks follow one another.  `*is_dense == 0`: hand the
 * table its row ids (llkv_hip_table_set_row_ids below).                          */
typedef struct llkv_chunk_meta {
  uint64_t row_count;
  uint64_t min_val_u64;
  uint64_t max_val_u64;
} llkv_chunk_meta;
llkv_status llkv_hip_dense_row_runs(const llkv_chunk_meta *rowid_chunks, uint32_t n_chunks,
                                    int32_t *is_dense, uint64_t *first_row_id);

/* Row ids that are not dense from 0: the row-id shadow column of the table (one array per local chunk, strictly
 * ascending — llkv-column-map keeps it beside every column, store/descriptor.rs:19-84; gaps appear where rows were
 * removed, gather over such ids: store/gather.rs:764-884).  Everything inside the library works on row POSITIONS
 * (predicates, windows, first-appearance order, probe order: all follow the position, which orders as the ids do); the
 * calls that report row ids — llkv_hip_filter_row_ids, llkv_hip_scan_stream with include_row_ids — translate positions to
 * these ids on the device.  llkv_hip_join_stream (index pairs) answers LLKV_UNSUPPORTED over such a table: its batch cuts
 * are positions and its pairs ids; llkv_hip_join_stream_batches reports no ids and takes it.  Ids that ARE the positions:
 * nothing is kept.  Call before the table is queried; once.                                                           */
llkv_status llkv_hip_table_set_row_ids(llkv_hip_table *table, const uint64_t *const *chunk_row_ids, uint32_t n_chunks);

/* Stage one column straight from its ARR0 chunk blobs (the pager blobs the
 * reference deserializes zero-copy, serialization.rs:438-488).                 */
llkv_status llkv_hip_table_append_arr0_column(llkv_hip_table *table, uint32_t field_id,
                                              const uint8_t *const *chunk_blobs,
                                              const uint64_t *chunk_blob_lens, uint32_t n_chunks,
                                              const char *const *dictionary, uint32_t dict_size);

/* Adopt a buffer that already lives in HBM (all local chunks back to back). */
llkv_status llkv_hip_table_adopt_device_column(llkv_hip_table *table, uint32_t field_id,
                                               int32_t dtype, const void *device_values);

/* Decimal128(precision, scale) column: 16-byte little-endian i128 raw values per row
 * (arrow Decimal128Array).  The reference cannot filter such a column at the leaf
 * (llkv-table/src/table.rs:1160-1167) but aggregates it exactly (i128 checked sums,
 * llkv-aggregate/src/lib.rs:925-967,1236-1284,1720-1804).  When every value of the
 * column fits 64 bits (DECIMAL(15,2) money columns always do) the HBM image is
 * narrowed to 8 B/row at staging — half the traffic, and a sum of < 2^63 rows can
 * no longer leave i128, so the order-dependent overflow check disappears.  A column
 * with a wider value is staged as low and high halves (16 B/row): SUM / TOTAL / AVG
 * (exact, when rows · max|v| ≤ i128::MAX excludes an overflowing prefix), MIN / MAX (when
 * the column's values span less than 2^64), the counts and plain scan projections take
 * it, every other use of it — and such a column in a sharded table — returns
 * LLKV_UNSUPPORTED.                                                                                     */
llkv_status llkv_hip_table_append_decimal128_column(llkv_hip_table *table, uint32_t field_id,
                                                    int32_t precision, int32_t scale,
                                                    const void *const *chunk_values,
                                                    uint32_t n_chunks);

/* Incremental growth — ColumnStore::append (llkv-column-map/src/store/core.rs:787): `n_new_chunks` chunks of `chunk_rows[]` rows
 * follow the table's last chunk, with the new chunks of EVERY staged column (`columns[n_columns]`: fixed-width value buffers,
 * arrow's 16-byte Decimal128 values, or Utf8 offsets + data; optional Arrow validity bitmaps) and, for a table with its own row
 * ids (llkv_hip_table_set_row_ids) or when the new ids are not the dense continuation, `chunk_row_ids` (else NULL).  Only the new
 * chunks cross the host → HBM link; a column image without room moves once on the device into a buffer with headroom.
 * Dictionaries grow by the new strings (beyond 256: LLKV_UNSUPPORTED — re-stage), statistics, validity and row ids follow.
 * Whatever the data can refuse is checked before anything is touched: a refused append leaves the table as it was.
 * Every successful append starts a new GENERATION of the image: a query prepared before it answers LLKV_INVALID_ARGUMENT at
 * launch (buffers may have moved, statistics decide lowerings) — prepare it again (lowering + a kernel-cache lookup, no staging).
 * No execution may be in flight during the call.  Unsharded tables only (world = 1).                                        */
typedef struct llkv_column_chunks {
  uint32_t field_id;
  const void *const *values;      /* [n_new_chunks] fixed width: value buffers; Decimal128: 16-byte raw values; Utf8: NULL   */
  const int32_t *const *offsets;  /* [n_new_chunks] Utf8: rows + 1 offsets per chunk                                         */
  const uint8_t *const *data;     /* [n_new_chunks] Utf8: string bytes                                                       */
  const uint8_t *const *validity; /* NULL, or [n_new_chunks] Arrow bitmaps (a NULL entry: every cell of the chunk present)   */
} llkv_column_chunks;
llkv_status llkv_hip_table_append_chunks(llkv_hip_table *table, const uint64_t *chunk_rows, uint32_t n_new_chunks,
                                         const llkv_column_chunks *columns, uint32_t n_columns,
                                         const uint64_t *const *chunk_row_ids);
uint64_t llkv_hip_table_generation(const llkv_hip_table *table); /* number of appends so far */
/* Key images: the 4-byte copies of Int64 key columns whose statistics fit 32 bits that the join → GROUP BY pipeline builds on first
 * use and streams in place of the 8-byte column.  Device memory the table holds beside its columns: 4 B per row
 * and image; an append drops them.  `LLKV_HIP_JOIN_NO_KEY_IMAGE=1` turns them off.                                               */
llkv_status llkv_hip_table_key_images(const llkv_hip_table *table, uint32_t *n_images, uint64_t *device_bytes);

/* Integer column statistics (the analogue of ChunkMetadata.min/max_val_u64, llkv-column-map/src/store/
 * descriptor.rs:19-84).  Plans use them (exact SUM without overflow tracking, dense integer GROUP BY), so every
 * rank of a sharded table must see the SAME, table-wide values: a single-rank table gets them from a reduction
 * at staging; with world > 1 the local values are only readable here, and the binding installs the table-wide
 * ones (the descriptor holds every chunk's min/max; or an all-reduce of the local values) before preparing
 * queries.  Without them the plans fall back to the statistics-free forms.                                   */
llkv_status llkv_hip_table_local_column_stats(const llkv_hip_table *table, uint32_t field_id,
                                              int32_t *has_stats, int64_t *min_value, int64_t *max_value);
llkv_status llkv_hip_table_set_column_stats(llkv_hip_table *table, uint32_t field_id,
                                            int64_t min_value, int64_t max_value);
/* The same for Float64 / Float32 columns: the largest and the smallest non-zero |v| over the column's finite values
 * (0 = the column holds no non-zero value).  They bound aggregate arguments from above and below, which is what lets
 * the shared-image GROUP BY keep its f64 sums exact and order-free (fused_scan.hip.h: SumF64X).                    */
llkv_status llkv_hip_table_local_column_float_stats(const llkv_hip_table *table, uint32_t field_id,
                                                    int32_t *has_stats, double *abs_max, double *abs_min_nonzero);
llkv_status llkv_hip_table_set_column_float_stats(llkv_hip_table *table, uint32_t field_id,
                                                  double abs_max, double abs_min_nonzero);
/* … and whether the column holds no NaN / ±∞ at all (`*all_finite` of this rank's rows; installed table-wide = every
 * rank's rows are): then such a sum is ONE integer lane of grid steps in the kernel's image (SumF64Q).            */
llkv_status llkv_hip_table_local_column_all_finite(const llkv_hip_table *table, uint32_t field_id, int32_t *all_finite);
llkv_status llkv_hip_table_set_column_all_finite(llkv_hip_table *table, uint32_t field_id, int32_t all_finite);

/* Bytes moved host → HBM by the staging calls of this process so far, and the wall
 * time those copies took (pinned ring fill + DMA; the host-side preparation of a
 * column image — dictionary coding, bitmap expansion — is not in it).  Purely
 * informational: staging happens once per resident column, outside every query.   */
void llkv_hip_staging_stats(uint64_t *bytes, double *seconds);
/* Page-locked host memory of the library: bytes resting in its cache of recycled blocks (windows of scan_stream, large id vectors,
 * exchange images; bounded by LLKV_HIP_PINNED_CACHE_MB, default 2048) and bytes handed out and not yet given back (results the
 * caller still holds — llkv_hip_free returns them).                                                                        */
void llkv_hip_pinned_stats(uint64_t *cached_bytes, uint64_t *outstanding_bytes);

/* NULL cells of an already staged column.  In the reference a NULL cell is a row
 * id that is absent from the column's row-id shadow chunks (llkv-table/src/
 * table.rs:1202-1223; gather turns it into an Arrow NULL, llkv-column-map/src/
 * store/projection.rs:929-1352); the binding passes that as one Arrow validity
 * bitmap per local chunk (LSB first, bit = 1 → present; a NULL pointer = the chunk
 * has no NULL cell).  The mask is kept as 1 B/row in HBM next to the values and
 * is read only by plans that touch the column; the values under NULL cells are
 * never observed.  A column without NULL cells stays on the NULL-free fast path. */
llkv_status llkv_hip_table_set_column_validity(llkv_hip_table *table, uint32_t field_id,
                                               const uint8_t *const *chunk_validity,
                                               uint32_t n_chunks);

/* ------------------------------------------------------------------------- */
/* Prepared queries: plan lowering + kernel selection happen once, launches   */
/* are asynchronous on a caller-supplied HIP stream.                          */
/* ------------------------------------------------------------------------- */
typedef struct llkv_hip_query llkv_hip_query;

/* Ungrouped aggregates with optional WHERE — replaces
 * `execute_aggregates` llkv-executor/src/lib.rs:5357-5682 and
 * `compute_aggregate_values` :6087-6665 (scan + accumulate + finalize).      */
llkv_status llkv_hip_query_prepare_aggregate(const llkv_hip_table *table,
                                             const llkv_filter *filters, uint32_t n_filters,
                                             const llkv_eval_op *ops, uint32_t n_ops,
                                             const llkv_aggregate_spec *aggs, uint32_t n_aggs,
                                             llkv_hip_query **out);
/* GROUP BY + aggregates — replaces `execute_group_by_single_table`
 * llkv-executor/src/lib.rs:4405-4542 and `execute_group_by_with_aggregates`
 * :5028-5355.  Groups come back in first-appearance order unless
 * `order_by_keys` (ascending lexsort over the keys, :13762-13868).           */
llkv_status llkv_hip_query_prepare_groupby(const llkv_hip_table *table,
                                           const llkv_filter *filters, uint32_t n_filters,
                                           const llkv_eval_op *ops, uint32_t n_ops,
                                           const uint32_t *key_fields, uint32_t n_keys,
                                           const llkv_aggregate_spec *aggs, uint32_t n_aggs,
                                           int32_t order_by_keys, llkv_hip_query **out);
void llkv_hip_query_free(llkv_hip_query *query);

/* Enqueue one execution on `hip_stream` (a hipStream_t, NULL = the library's
 * own stream).                                                               */
llkv_status llkv_hip_query_launch(llkv_hip_query *query, void *hip_stream);
/* Device buffer of `len` int64 lanes holding this rank's partial aggregate
 * state of the oldest execution awaiting submission (slot s of the ring lives
 * at base + s·len, executions use slots round-robin), zero where another rank
 * owns the lane.  With world > 1 the caller
 * all-reduces it (ncclSum over ncclInt64 — exact for every lane type because
 * exactly one rank contributes non-zero bits per lane) before finish.        */
llkv_status llkv_hip_query_exchange_buffer(llkv_hip_query *query, void **device_ptr,
                                           uint64_t *len_i64);
/* Copy the (combined) state to the host, fold it in canonical order and
 * finalize.  Blocks until done.                                              */
llkv_status llkv_hip_query_finish(llkv_hip_query *query, void *hip_stream);

/* Pipelined form (several executions of one prepared query in flight, e.g. many
 * concurrent sessions): set_depth(D) allows D launches before the oldest must be
 * collected.  Per execution: launch → [caller's all-reduce of exchange_buffer] →
 * submit (enqueue the copy-out) … later collect (wait + fold + finalize the
 * OLDEST submitted execution; results readable until the next collect).
 * finish() = submit + collect everything outstanding.                         */
llkv_status llkv_hip_query_set_depth(llkv_hip_query *query, uint32_t depth);
/* In steady state an execution is ONE kernel launch: the scan of execution i also folds
 * the tile partials of execution i-1 (its first workgroups), so the exchange image of an
 * execution completes with the NEXT launch — or with a small standalone fold when its
 * result is requested first.  wait_folded makes `hip_stream` wait for the image of the
 * oldest execution not yet submitted (call it before an all-reduce issued on a
 * communication stream); submit(hip_stream) orders that execution's copy-out after
 * prior work on `hip_stream` (NULL = the stream it was launched on).                   */
llkv_status llkv_hip_query_wait_folded(llkv_hip_query *query, void *hip_stream);
llkv_status llkv_hip_query_submit(llkv_hip_query *query, void *hip_stream);
llkv_status llkv_hip_query_collect(llkv_hip_query *query);

/* Blocking copy of the latest launch's exchange image (len_i64 lanes) to host memory —
 * for hosts that run the collective themselves, and for tests.                        */
llkv_status llkv_hip_query_read_exchange(llkv_hip_query *query, uint64_t *out, uint64_t len_i64);

/* Same, from an exchange image the caller already holds on the host (for hosts
 * that run the collective themselves; `len_i64` must match exchange_buffer). */
llkv_status llkv_hip_query_finish_from_host(llkv_hip_query *query, const uint64_t *exchange,
                                            uint64_t len_i64);

/* Results of the last finish(). Aggregate queries have exactly one group.    */
uint32_t llkv_hip_query_num_groups(const llkv_hip_query *query);
uint32_t llkv_hip_query_num_keys(const llkv_hip_query *query);
uint32_t llkv_hip_query_num_aggregates(const llkv_hip_query *query);
llkv_status llkv_hip_query_group_key(const llkv_hip_query *query, uint32_t group, uint32_t key,
                                     llkv_value *out);
llkv_status llkv_hip_query_value(const llkv_hip_query *query, uint32_t group, uint32_t agg,
                                 llkv_value *out);

/* DISTINCT aggregates over a SHARDED table (ungrouped COUNT / SUM / TOTAL / AVG): after launch + finish every rank
 * exports the distinct values of its rows in order of first appearance (64-bit images: i64 values or f64 bit
 * patterns), the binding all-gathers them, and llkv_hip_query_merge_distinct computes the aggregate over their union
 * in rank order = the table's order of first appearance (checked i64 sums, f64 sums 0.0 then += — the accumulator's
 * own order, so the merged f64 sums are bit-exact with the reference's).  llkv_hip_query_value then returns it.      */
llkv_status llkv_hip_query_distinct_partial(llkv_hip_query *query, uint32_t agg, const uint64_t **values,
                                            uint64_t *n_values);
llkv_status llkv_hip_query_merge_distinct(llkv_hip_query *query, uint32_t agg, uint32_t world,
                                          const uint64_t *rank_counts, const uint64_t *const *rank_values);

/* GROUP BY of any cardinality over a SHARDED table (the sort-based route; the dense route combines through the
 * exchange image above).  Every rank runs the query over its own chunks — launch, finish — and then holds partial
 * groups: `key_values[n_keys][n]` (integers; Utf8 as codes of the table-wide dictionary), `key_valid[n_keys][n]`
 * and `lanes[n][lanes_per_group]` (the accumulator lanes), valid until the next launch.  The binding all-gathers
 * the three arrays of every rank and installs the table-wide groups with llkv_hip_query_merge_groups (rank order
 * = row order: states of one key are combined lane by lane in that order; the groups come out in key order or in
 * first-appearance order, as on one device); group_key / value then read the merged result.  Integer results are
 * those of one device bit for bit, f64 sums within the 1e-9 of the contract (one more level of association).      */
llkv_status llkv_hip_query_partial_groups(const llkv_hip_query *query, uint64_t *n_groups, uint32_t *n_keys,
                                          uint32_t *lanes_per_group, const int64_t **key_values,
                                          const uint8_t **key_valid, const uint64_t **lanes);
llkv_status llkv_hip_query_merge_groups(llkv_hip_query *query, uint32_t world, const uint64_t *rank_groups,
                                        const int64_t *const *key_values, const uint8_t *const *key_valid,
                                        const uint64_t *const *lanes);
/* Status a finalize step produced for one aggregate (e.g. "integer overflow"
 * is LLKV_INVALID_ARGUMENT, llkv-aggregate/src/lib.rs:816-829).              */

/* Measurement hooks (bench only): HIP-event time of the dominant kernel,
 * summed over launches since the last reset, and the launch count.           */
/* enabled: 0 off, 1 every launch, n > 1 every n-th launch (fewer event packets
 * between back-to-back kernels).                                               */
llkv_status llkv_hip_query_set_profiling(llkv_hip_query *query, int32_t enabled);
llkv_status llkv_hip_query_kernel_time(llkv_hip_query *query, double *total_ms,
                                       uint64_t *launches, const char **kernel_name);
/* Algorithmic bytes one launch reads on this rank (value buffers, once). */
uint64_t llkv_hip_query_algorithmic_bytes(const llkv_hip_query *query);
/* Name of the compiled kernel variant chosen for this plan. */
const char *llkv_hip_query_kernel_signature(const llkv_hip_query *query);
/* Which kernel family serves the plan — register accumulators, per-thread accumulator columns in LDS (≤ 64 groups),
 * one shared accumulator image per workgroup (hundreds … thousands of groups, order-free lanes), the sort-based
 * route — and, when a cheaper family declined, why.                                                              */
const char *llkv_hip_query_route_note(const llkv_hip_query *query);

/* One-shot conveniences (prepare + launch + finish + copy out + free). */
llkv_status llkv_hip_aggregate(const llkv_hip_table *table, const llkv_filter *filters,
                               uint32_t n_filters, const llkv_eval_op *ops, uint32_t n_ops,
                               const llkv_aggregate_spec *aggs, uint32_t n_aggs,
                               llkv_value *out_values);

/* ------------------------------------------------------------------------- */
/* Streaming scan — `StorageTable::scan_stream`                               */
/* llkv-executor/src/types/storage.rs:20-50 → `execute_scan`                   */
/* llkv-scan/src/execute.rs:47-295: filter → selection vector → 65 536-row     */
/* windows → gathered projections → on_batch, ascending row-id order, never an */
/* empty batch.                                                               */
/* ------------------------------------------------------------------------- */
typedef struct llkv_projection {
  int32_t computed;            /* 0 = ScanProjection::Column, 1 = ::Computed  */
  uint32_t field_id;           /* column projection                           */
  const llkv_expr_token *expr; /* computed projection (postfix)               */
  uint32_t expr_len;
  const char *alias;
} llkv_projection;

typedef enum llkv_order_transform { /* ScanOrderTransform llkv-scan/src/lib.rs:41-46 */
  LLKV_ORDER_IDENTITY_INT64 = 0,
  LLKV_ORDER_IDENTITY_INT32 = 1,
  LLKV_ORDER_IDENTITY_UTF8 = 2,
  LLKV_ORDER_CAST_UTF8_TO_INTEGER = 3
} llkv_order_transform;

typedef struct llkv_scan_options {
  int32_t include_nulls;   /* ScanStreamOptions.include_nulls                 */
  int32_t include_row_ids; /* ScanStreamOptions.include_row_ids               */
  /* ScanStreamOptions.order (ScanOrderSpec, llkv-scan/src/lib.rs:48-55; sort_row_ids_with_order llkv-scan/src/
   * ordering.rs:16-140): the selected rows are sorted by one column before they are cut into windows.  Rows
   * with equal keys keep row-id order (arrow's sort leaves ties unspecified).                                  */
  int32_t order_enabled;
  uint32_t order_field;
  int32_t order_descending;
  int32_t order_nulls_first;
  int32_t order_transform; /* llkv_order_transform                            */
} llkv_scan_options;

typedef struct llkv_column_view {
  int32_t dtype;
  const void *values;      /* host memory, valid during the callback only     */
  const uint8_t *validity; /* Arrow validity bitmap or NULL (all valid)       */
  const char *const *dictionary; /* Utf8: code → string, else NULL            */
  int32_t precision, scale;      /* Decimal128: 16-byte little-endian values  */
} llkv_column_view;

typedef struct llkv_batch_view {
  uint64_t num_rows;
  uint32_t num_columns;
  const llkv_column_view *columns;
  const uint64_t *row_ids; /* NULL unless include_row_ids                     */
} llkv_batch_view;

typedef void (*llkv_on_batch)(const llkv_batch_view *batch, void *user);

/* Arrow C Data Interface (https://arrow.apache.org/docs/format/CDataInterface.html), the standard definitions.  */
#ifndef ARROW_C_DATA_INTERFACE
#define ARROW_C_DATA_INTERFACE
struct ArrowSchema {
  const char *format;
  const char *name;
  const char *metadata;
  int64_t flags;
  int64_t n_children;
  struct ArrowSchema **children;
  struct ArrowSchema *dictionary;
  void (*release)(struct ArrowSchema *);
  void *private_data;
};
struct ArrowArray {
  int64_t length;
  int64_t null_count;
  int64_t offset;
  int64_t n_buffers;
  int64_t n_children;
  const void **buffers;
  struct ArrowArray **children;
  struct ArrowArray *dictionary;
  void (*release)(struct ArrowArray *);
  void *private_data;
};
#endif

/* One scan batch as an Arrow RecordBatch (a struct array: one child per column, then "rowid" when the view has
 * row ids).  The reference passes batches to `on_batch` by value — Arc-backed buffers the callee may keep
 * (llkv-executor/src/types/storage.rs:20-50) — while a view lives only during the callback: this copies the view
 * into buffers the consumer owns and frees through the release callbacks (arrow-rs `from_ffi`, pyarrow
 * `_import_from_c`, …).  Fields are nullable as in llkv-scan/src/execute.rs:166-181; Utf8 columns are
 * materialised from their dictionary codes.  `column_names` may be NULL ("c0", "c1", …).  Host only.           */
llkv_status llkv_hip_batch_export_arrow(const llkv_batch_view *batch, const char *const *column_names,
                                        struct ArrowArray *out_array, struct ArrowSchema *out_schema);

/* Stage one column from Arrow arrays, one per local chunk (Int32/64, UInt32/64, Float32/64, Date32, Boolean,
 * Decimal128, Utf8; offsets and validity bitmaps honoured) — the arrays `deserialize_array` yields for the
 * reference's chunk blobs (llkv-column-map/src/serialization.rs:438-488).  `dictionary` as for
 * llkv_hip_table_append_utf8_column.  The arrays are only read; they stay the caller's.                          */
llkv_status llkv_hip_table_append_arrow_column(llkv_hip_table *table, uint32_t field_id,
                                               const struct ArrowSchema *schema,
                                               const struct ArrowArray *const *chunks, uint32_t n_chunks,
                                               const char *const *dictionary, uint32_t dict_size);

llkv_status llkv_hip_scan_stream(const llkv_hip_table *table, const llkv_projection *projections,
                                 uint32_t n_projections, const llkv_filter *filters,
                                 uint32_t n_filters, const llkv_eval_op *ops, uint32_t n_ops,
                                 const llkv_scan_options *options, llkv_on_batch on_batch,
                                 void *user);
/* `StorageTable::filter_row_ids` (storage.rs:34-37): matching row ids, ascending.
 * `*out_row_ids` belongs to the library (large vectors live in recycled pinned
 * host memory, written by the device at PCIe speed); release it with
 * llkv_hip_free, never with free().                                          */
llkv_status llkv_hip_filter_row_ids(const llkv_hip_table *table, const llkv_filter *filters,
                                    uint32_t n_filters, const llkv_eval_op *ops, uint32_t n_ops,
                                    uint64_t **out_row_ids, uint64_t *out_len);
void llkv_hip_free(void *ptr);

/* ------------------------------------------------------------------------- */
/* Hash join — `TableJoinExt::join_stream` llkv-join/src/lib.rs:240-282.        */
/* Build = right, probe = left; output = matching (left row, right row) index  */
/* pairs in probe order × build insertion order, in the reference's batches.   */
/* · one key pair of one integer type (Int32/Int64/UInt32/UInt64) on both      */
/*   sides: the integer fast path, hash_join.rs:955-1417 (null_equals_null =   */
/*   per-type sentinel; a batch ends after the probe row that brings it to     */
/*   ≥ batch_size pairs and at the end of every 65 536-row probe scan batch);  */
/* · any other key list of 1..4 pairs: the generic typed-key path, :200-335,   */
/*   :377-505 (all parts equal; NULL equals nothing or, under                  */
/*   null_equals_null, the marker Utf8("<NULL>"); floats by bit pattern;       */
/*   values of two different types never equal; a value of a type              */
/*   extract_key_value does not list — Date32, Boolean, Decimal128 — takes the */
/*   row out; the probe is cut into slices of batch_size rows first);          */
/* · no key pair: the Cartesian product, :1500-1599.                           */
/* ------------------------------------------------------------------------- */
typedef enum llkv_join_type {
  LLKV_JOIN_INNER = 0,
  LLKV_JOIN_LEFT = 1,
  LLKV_JOIN_RIGHT = 2, /* unimplemented in the reference (hash_join.rs:328-332) */
  LLKV_JOIN_FULL = 3,  /* idem                                                 */
  LLKV_JOIN_SEMI = 4,
  LLKV_JOIN_ANTI = 5
} llkv_join_type;

typedef struct llkv_join_key {
  uint32_t left_field;
  uint32_t right_field;
  int32_t null_equals_null;
} llkv_join_key;

/* Whose key rules a join follows.                                              */
typedef enum llkv_join_key_rules {
  /* llkv-join's TableJoinExt (the two paths described above)                   */
  LLKV_JOIN_KEYS_TABLE = 0,
  /* the executor's SQL joins — hash_join_table_batches / normalize_join_column */
  /* / build_join_match_indices / build_left_join_match_indices,                */
  /* llkv-executor/src/lib.rs:12218-12581: key columns are normalised first     */
  /* (Boolean and every integer type → Int64 — a UInt64 ≥ 2^63 becomes NULL —,  */
  /* Float32 → Float64), then compared as arrow-row bytes: equal only inside    */
  /* one class (Int64, Float64 by bit pattern, Utf8, Date32, Decimal128 by raw   */
  /* value); a NULL in any key part never matches; INNER and LEFT only (other   */
  /* types: Internal, "use llkv-join"); the reference materialises ONE batch,   */
  /* here the pairs arrive in probe order over several callbacks (batch_size is */
  /* not looked at).                                                            */
  LLKV_JOIN_KEYS_EXECUTOR = 1
} llkv_join_key_rules;

typedef struct llkv_join_options {
  int32_t join_type;   /* llkv_join_type                                      */
  uint64_t batch_size; /* JoinOptions.batch_size (default 8192); 0 = error    */
  int32_t key_rules;   /* llkv_join_key_rules                                 */
} llkv_join_options;

/* Index-pair batch: right_rows[i] == UINT64_MAX marks a NULL-padded right side
 * (LEFT join).  SEMI/ANTI deliver left rows only (right_rows == NULL).       */
typedef void (*llkv_on_join_batch)(const uint64_t *left_rows, const uint64_t *right_rows,
                                   uint64_t n_pairs, void *user);

llkv_status llkv_hip_join_stream(const llkv_hip_table *left, const llkv_hip_table *right,
                                 const llkv_join_key *keys, uint32_t n_keys,
                                 const llkv_join_options *options, llkv_on_join_batch on_batch,
                                 void *user);

/* The same join delivering what the reference's `on_batch` receives: the joined RecordBatch, gathered on the    */
/* device (emit_joined_batch / emit_left_joined_batch / emit_semi_batch llkv-join/src/hash_join.rs:715-772,     */
/* cross_join_pair llkv-join/src/cartesian.rs:22-110, synthesize_left_join_nulls :1468-1497; the executor's     */
/* hash_join_table_batches llkv-executor/src/lib.rs:12218-12392 under LLKV_JOIN_KEYS_EXECUTOR).                 */
/* · `left_columns` / `right_columns` are the USER columns of the two tables in schema order with their names   */
/*   (build_user_projections :782-806: every schema field that carries a field id; the row-id column never is   */
/*   one), so the batch holds the left columns, then the right columns — SEMI / ANTI: the left columns only     */
/*   (build_output_schema :877-943).  A right name already taken gets the suffix "_1" (:913-939; the executor   */
/*   rules keep the names as given, :12237-12244).  `column_names` of the callback are the output names.        */
/* · Both sides are read as `scan_stream(all user columns, ScanStreamOptions::default())` reads them            */
/*   (:203-240,:346-375): a row that is NULL in EVERY user column is dropped by the scan's DropNulls gather      */
/*   (llkv-column-map/src/store/projection.rs:1326-1330) — it joins nothing and is not padded by a LEFT join    */
/*   (LLKV_JOIN_KEYS_TABLE only; the generic typed-key path over a probe side that holds such rows answers      */
/*   LLKV_UNSUPPORTED: its slices count the rows that survive).                                                  */
/* · A LEFT join pads the right columns of an unmatched row with NULLs; every right column of a LEFT join       */
/*   carries a validity bitmap.  Batches are cut exactly where the reference cuts them (see above); rows keep   */
/*   probe order × build insertion order.  `batch->row_ids` is NULL.  An empty column list on either side:      */
/*   the reference scans nothing there (:211-221,:226) — no left columns, no batches.                           */
/* The view and the names are valid during the callback; llkv_hip_batch_export_arrow makes an owned RecordBatch. */
typedef struct llkv_join_column {
  uint32_t field_id;
  const char *name;
} llkv_join_column;

typedef struct llkv_join_output {
  const llkv_join_column *left_columns;
  uint32_t n_left;
  const llkv_join_column *right_columns;
  uint32_t n_right;
} llkv_join_output;

typedef void (*llkv_on_join_record_batch)(const llkv_batch_view *batch, const char *const *column_names,
                                          void *user);

llkv_status llkv_hip_join_stream_batches(const llkv_hip_table *left, const llkv_hip_table *right,
                                         const llkv_join_key *keys, uint32_t n_keys,
                                         const llkv_join_options *options, const llkv_join_output *output,
                                         llkv_on_join_record_batch on_batch, void *user);

/* build_output_schema alone (hash_join.rs:877-943): writes the n_left (+ n_right unless SEMI / ANTI) output     */
/* names into `names` (each a malloc'ed string the caller frees with llkv_hip_free) and their count.  Host only. */
llkv_status llkv_hip_join_output_names(const llkv_join_output *output, int32_t join_type, int32_t key_rules,
                                       char **names, uint32_t *n_names);

/* ------------------------------------------------------------------------- */
/* Join → GROUP BY → ORDER BY … LIMIT for the TPC-H Q3 shape — the executor's   */
/* multi-table route: try_execute_hash_join llkv-executor/src/lib.rs:3780-4052, */
/* hash_join_table_batches :12218-12392, post-join filter mask :1629-1646,      */
/* execute_group_by_from_batches, sort_record_batch_with_order :13762-13868,    */
/* LIMIT :10925-10955.                                                          */
/*   fact ⋈ dim [⋉ dim2]  GROUP BY dim.key, payload…   SUM(fact expr)           */
/*   ORDER BY sum DESC, payload[0] ASC   LIMIT k                                */
/* dim.key must be unique (a primary key); groups are then identified by the    */
/* dim row.  Sums add the fact rows of a group in scan order, like the          */
/* reference, so they are bit-exact with it.                                    */
/* ------------------------------------------------------------------------- */
typedef struct llkv_join_side {
  const llkv_hip_table *table;
  const llkv_filter *filters; /* conjunction (Expr::all_of)                      */
  uint32_t n_filters;
  uint32_t key_field;         /* integer join key                                */
} llkv_join_side;

typedef struct llkv_join_group_row {
  int64_t key;        /* dim key (= fact key) of the group                       */
  double sum;
  uint64_t count;     /* fact rows in the group                                  */
  int64_t payload[4]; /* dim payload columns (integers / Date32)                 */
  uint64_t group_index; /* position of the dim row among the qualifying dim rows (row order): the final
                         * tie-break of the ordering, identical on every rank                            */
} llkv_join_group_row;

llkv_status llkv_hip_join_groupby_topk(const llkv_join_side *fact, const llkv_join_side *dim,
                                       uint32_t dim_fk_field, const llkv_join_side *dim2 /* may be NULL */,
                                       const uint32_t *payload_fields, uint32_t n_payload,
                                       const llkv_expr_token *sum_expr, uint32_t sum_expr_len,
                                       uint32_t limit, llkv_join_group_row *out_rows, uint32_t *out_n,
                                       uint64_t *out_total_groups);

/* The same pipeline for a fact table sharded over ranks (one process per GPU):
 * the dimension tables are staged whole (world = 1) on every rank, the fact table by chunk.
 *   1. prepare            local build, probe, per-group sums and row counts (group = qualifying dim row)
 *   2. counts_buffer      int64[len] in HBM: the binding all-reduces it (SUM) in place — the one sizeable
 *                         collective (8 B per qualifying dim row)
 *   3. straddlers         this rank's raw (group, value) pairs, in row order, of the groups whose rows are
 *                         spread over several ranks; the binding all-gathers them (few for a fact table
 *                         clustered by the key) and concatenates in rank order
 *   4. fold_straddlers    host only: exact left-to-right sums of those groups in global row order
 *   5. candidates         the top `limit` groups this rank reports (held alone, or straddlers held first)
 *   6. merge              host only: ORDER BY / LIMIT over the all-gathered candidates
 * Results are bit-identical to the single-GPU call for every rank count.  The three tables must outlive the
 * handle (its later phases still read their HBM images).                                                */
typedef struct llkv_hip_join_agg llkv_hip_join_agg;
llkv_status llkv_hip_join_agg_prepare(const llkv_join_side *fact, const llkv_join_side *dim,
                                      uint32_t dim_fk_field, const llkv_join_side *dim2 /* may be NULL */,
                                      const uint32_t *payload_fields, uint32_t n_payload,
                                      const llkv_expr_token *sum_expr, uint32_t sum_expr_len,
                                      llkv_hip_join_agg **out);
void llkv_hip_join_agg_free(llkv_hip_join_agg *h);
llkv_status llkv_hip_join_agg_counts_buffer(llkv_hip_join_agg *h, void **device_ptr, uint64_t *len_i64);
llkv_status llkv_hip_join_agg_straddlers(llkv_hip_join_agg *h, const uint32_t **groups,
                                         const double **values, uint64_t *n);
llkv_status llkv_hip_join_agg_fold_straddlers(const uint32_t *groups, const double *values,
                                              const uint64_t *rank_offsets /* [world + 1] */, uint32_t world,
                                              uint32_t *out_groups, double *out_sums, uint64_t *out_counts,
                                              uint32_t *out_first_rank, uint64_t *n_out /* capacity in, count out */);
llkv_status llkv_hip_join_agg_candidates(llkv_hip_join_agg *h, const uint32_t *folded_groups,
                                         const double *folded_sums, const uint64_t *folded_counts,
                                         const uint32_t *folded_first_rank, uint64_t n_folded, uint32_t rank,
                                         uint32_t limit, llkv_join_group_row *out_rows, uint32_t *out_n,
                                         uint64_t *out_groups /* groups this rank reports */);
llkv_status llkv_hip_join_agg_merge(const llkv_join_group_row *rows, uint32_t n, uint32_t n_payload,
                                    uint32_t limit, llkv_join_group_row *out_rows, uint32_t *out_n);

/* The RANGE form of the sharded pipeline, for a fact table clustered by the join key (lineitem by l_orderkey) and a
 * dimension whose key column is in ascending row order with a statistics-bounded range (orders by o_orderkey).  A rank
 * then needs the dimension rows of its own key range only: the dimension scan leaves at once every tile whose keys lie
 * outside [min, max] of the rank's fact keys (1/world of the dimension work instead of all of it), group ids are local
 * (rank of the key in the rank's own bitmap) and nothing is exchanged per group: a group can straddle two ranks only as
 * the LAST group of one rank's pair stream and the FIRST of the next one's, so every rank publishes those two runs as raw
 * values (`boundary`: a few hundred bytes), folds the shared ones in rank order = global row order, and reports its
 * candidates as before.  Per query and rank: ~1 KB of boundary runs + `limit` candidate rows, instead of 8 B per
 * qualifying dimension row.  Results are bit-identical to the single-GPU call.
 *   prepare_ranged   as prepare; LLKV_UNSUPPORTED when the shape does not qualify (take prepare)
 *   boundary         this rank's block for the all-gather (valid until the handle is freed)
 *   finish_ranged    with every rank's block (rank order; `offsets[world + 1]`): folds the shared boundary groups and
 *                    returns this rank's candidates and how many groups it reports; LLKV_UNSUPPORTED — identically on
 *                    every rank — when the pair streams turn out not to be in key order across the ranks, or a boundary
 *                    run is longer than the block holds (64 rows): free the handle and take the general form
 *   merge            as above, over the all-gathered candidates
 * llkv_hip_join_agg_finish_sharded runs either form's collectives; counts_buffer / straddlers / candidates refuse a
 * ranged handle.                                                                                                    */
llkv_status llkv_hip_join_agg_prepare_ranged(const llkv_join_side *fact, const llkv_join_side *dim,
                                             uint32_t dim_fk_field, const llkv_join_side *dim2 /* may be NULL */,
                                             const uint32_t *payload_fields, uint32_t n_payload,
                                             const llkv_expr_token *sum_expr, uint32_t sum_expr_len,
                                             llkv_hip_join_agg **out);
llkv_status llkv_hip_join_agg_boundary(llkv_hip_join_agg *h, const void **block, uint64_t *bytes);
/* Measurement hook: bytes the collectives of the last llkv_hip_join_agg_finish_sharded call moved (all ranks' shares). */
uint64_t llkv_hip_join_agg_exchange_bytes(const llkv_hip_join_agg *h);
llkv_status llkv_hip_join_agg_finish_ranged(llkv_hip_join_agg *h, const void *blocks, const uint64_t *offsets /* [world + 1] */,
                                            uint32_t world, uint32_t rank, uint32_t limit, llkv_join_group_row *out_rows,
                                            uint32_t *out_n, uint64_t *out_groups /* groups this rank reports */);

/* ------------------------------------------------------------------------- */
/* Join → GROUP BY with ANY aggregate list — the general form of the star shape    */
/* above (execute_group_by_from_batches llkv-executor/src/lib.rs:4544-4755 over     */
/* the batches of try_execute_hash_join :3780-4052; ORDER BY :13762-13868,          */
/* LIMIT :10925-10955):                                                             */
/*   fact ⋈ dim [⋉ dim2]   GROUP BY dim.key [, payload …]                           */
/*   COUNT(*) / COUNT / SUM / TOTAL / AVG / MIN / MAX over fact-side expressions,   */
/*   several of them; ORDER BY any of the aggregates, the payload columns and the   */
/*   key, ASC / DESC, NULLS FIRST / LAST; LIMIT or none.                            */
/* `prepare` builds the key set of the qualifying dimension rows (dim.key must be   */
/* unique among them: LLKV_UNSUPPORTED otherwise, and for key columns without a     */
/* statistics-bounded range) and returns a prepared GROUP BY of the fact key over   */
/* the fact rows that pass `filters AND key IN (key set)`: it runs like any other   */
/* prepared query — llkv_hip_query_launch + llkv_hip_query_finish, or               */
/* llkv_hip_query_finish_sharded for a fact table sharded over ranks (dimension     */
/* tables replicated; the ranks' partial groups are merged lane by lane in rank     */
/* order) — and its groups can be read through llkv_hip_query_group_key / _value.   */
/* Aggregate arguments follow the GROUP BY (PlanValue) semantics, decimals          */
/* included.  `rows` looks every group's dimension row up (payload cells: integer / */
/* Date32 columns of dim), applies ORDER BY — ties broken by the dimension row's    */
/* position, so every rank count gives the same rows — and LIMIT (UINT64_MAX: all). */
/* llkv_hip_join_groupby_topk stays the hand-tuned form of the single-SUM shape.    */
/* ------------------------------------------------------------------------- */
typedef enum llkv_join_order_kind {
  LLKV_JOIN_ORDER_AGGREGATE = 0, /* index = position in the aggregate list */
  LLKV_JOIN_ORDER_PAYLOAD = 1,   /* index = position in the payload list   */
  LLKV_JOIN_ORDER_KEY = 2        /* the group key (dim.key)                */
} llkv_join_order_kind;

typedef struct llkv_join_order_key {
  int32_t kind;  /* llkv_join_order_kind */
  uint32_t index;
  int32_t descending;
  int32_t nulls_first;
} llkv_join_order_key;

typedef struct llkv_hip_join_rows llkv_hip_join_rows;

llkv_status llkv_hip_join_groupby_prepare(const llkv_join_side *fact, const llkv_join_side *dim, uint32_t dim_fk_field,
                                          const llkv_join_side *dim2 /* may be NULL */, const llkv_aggregate_spec *aggs,
                                          uint32_t n_aggs, llkv_hip_query **out);
llkv_status llkv_hip_join_groupby_rows(llkv_hip_query *query, const uint32_t *payload_fields, uint32_t n_payload,
                                       const llkv_join_order_key *order, uint32_t n_order, uint64_t limit,
                                       llkv_hip_join_rows **out);
uint64_t llkv_hip_join_rows_len(const llkv_hip_join_rows *rows);
uint64_t llkv_hip_join_rows_total_groups(const llkv_hip_join_rows *rows); /* groups before LIMIT */
/* Row i: the group key, payload[n_payload] (+ payload_is_null), the dimension row's position among the qualifying rows, and a
 * pointer to its n_aggs finalized cells (owned by `rows`).  Any output pointer may be NULL.                                */
llkv_status llkv_hip_join_rows_get(const llkv_hip_join_rows *rows, uint64_t i, int64_t *key, int64_t *payload,
                                   uint8_t *payload_is_null, uint64_t *group_index, const llkv_value **values);
void llkv_hip_join_rows_free(llkv_hip_join_rows *rows);

/* ------------------------------------------------------------------------- */
/* Multi-GPU combine, host pieces (no device needed).  The chunk list is cut    */
/* into 8 canonical octants (boundaries floor(j·C/8)); rank r of `world` owns   */
/* octants [r·8/world, (r+1)·8/world).  Partial aggregate state is exchanged    */
/* per octant and folded in octant order, so 1/2/4/8-GPU runs agree bit for bit.*/
/* ------------------------------------------------------------------------- */
#define LLKV_HIP_OCTANTS 8
llkv_status llkv_hip_shard_layout(uint32_t n_chunks, uint32_t world,
                                  uint32_t *octant_chunk_begin /* [9] */,
                                  uint32_t *octant_owner /* [8] */);
/* Lane combine ops of a prepared query (one byte per lane: 0 add f64, 1 add i64,
 * 2 min i64, 3 max i64, 4 max u64). `ops_out` may be NULL to query the count.  */
llkv_status llkv_hip_query_lane_ops(const llkv_hip_query *query, uint8_t *ops_out, uint32_t *lanes);
/* Canonical fold of an exchange buffer [8][lanes] → state[lanes] (what finish()
 * does after the copy-out).                                                    */
llkv_status llkv_hip_fold_exchange(const uint64_t *exchange, const uint8_t *lane_ops,
                                   uint32_t lanes, uint64_t *state_out);

/* ------------------------------------------------------------------------- */
/* Collectives behind the boundary (one process per GPU).  The reference has no */
/* counterpart (single process; its only concurrency boundary is the Rayon pool, */
/* llkv-threading/src/lib.rs:75-82): these calls are what a multi-GPU            */
/* `QueryExecutor` shim adds around the sharded forms of the queries above.       */
/* Transport: RCCL over xGMI (ncclCommInitRank on the bound device), or functions */
/* the host supplies (MPI, a socket layer, gloo in the tests) that see HOST       */
/* memory.  One communicator per process; every rank makes the same calls in the  */
/* same order.                                                                    */
/* ------------------------------------------------------------------------- */
#define LLKV_HIP_COMM_ID_BYTES 128
/* Rank 0: a fresh ncclUniqueId (128 bytes) to hand to the other ranks over the host's own channel. */
llkv_status llkv_hip_comm_unique_id(uint8_t id_out[LLKV_HIP_COMM_ID_BYTES]);
/* Every rank, after llkv_hip_init: join the RCCL communicator `id` names. */
llkv_status llkv_hip_comm_init(const uint8_t id[LLKV_HIP_COMM_ID_BYTES], uint32_t rank, uint32_t world);
/* Host-supplied transport.  Both functions return 0 on success and are called from the thread that calls into the
 * library: all_reduce_sum_i64 sums `n` int64 lanes over the ranks in place; all_gather concatenates every rank's
 * `bytes_per_rank` bytes in rank order into `recv` (world · bytes_per_rank bytes).                                  */
typedef struct llkv_comm_transport {
  int32_t (*all_reduce_sum_i64)(int64_t *host_buf, uint64_t n, void *user);
  int32_t (*all_gather)(const void *send, void *recv, uint64_t bytes_per_rank, void *user);
  void *user;
} llkv_comm_transport;
llkv_status llkv_hip_comm_init_custom(const llkv_comm_transport *transport, uint32_t rank, uint32_t world);
void llkv_hip_comm_destroy(void);
uint32_t llkv_hip_comm_rank(void);
uint32_t llkv_hip_comm_world(void); /* 0 = no communicator */
/* What the communicator is: backend 0 = none, 1 = RCCL, 2 = host transport; `ranks` = what the communicator itself reports
 * (RCCL: ncclCommCount) — a benchmark line quotes these instead of what it asked for. */
llkv_status llkv_hip_comm_describe(int32_t *backend, uint32_t *ranks);

/* In-place SUM of a device buffer of int64 lanes over the ranks, ordered on `hip_stream` (RCCL: ncclAllReduce,
 * nothing blocks on the host).                                                                                   */
llkv_status llkv_hip_comm_all_reduce_i64(void *device_buf, uint64_t n, void *hip_stream);
/* Variable-length all-gather of host bytes: `*out` (release with llkv_hip_free) holds the ranks' contributions in
 * rank order, `offsets_out[world + 1]` their bounds.                                                             */
llkv_status llkv_hip_comm_all_gather_v(const void *send, uint64_t bytes, void **out, uint64_t *offsets_out);
/* Sorted union of the ranks' string lists — the table-wide dictionary a sharded Utf8 column is staged with
 * (llkv_hip_table_append_utf8_column).  `*out` is one block: `*n_out` pointers followed by the characters;
 * release with llkv_hip_free.                                                                                     */
llkv_status llkv_hip_comm_union_strings(const char *const *local, uint32_t n_local, char ***out, uint32_t *n_out);

/* Table metadata plans depend on must be the same on every rank of a sharded table BEFORE queries are prepared:
 * integer statistics (min / max over all shards, see llkv_hip_table_set_column_stats) and whether a column has NULL
 * cells (a rank whose chunks hold none still gets an all-present mask, so that every rank lowers the same plan: same
 * lanes, same exchange image).  All-gathers the local values of every staged column and installs the agreement.    */
llkv_status llkv_hip_table_share_metadata(llkv_hip_table *table);

/* The one collective of a dense aggregate / GROUP BY query: make `hip_stream` wait for the exchange image of the
 * oldest execution not yet submitted (llkv_hip_query_wait_folded), then sum it over the ranks in place on that
 * stream.  Per execution: launch → all_reduce → submit → collect.                                                 */
llkv_status llkv_hip_query_all_reduce(llkv_hip_query *query, void *hip_stream);
/* finish() of any query over a sharded table, collectives included: dense plans — all_reduce + submit + collect of
 * everything outstanding; the sort-based GROUP BY — all-gather of the ranks' partial groups and their merge
 * (llkv_hip_query_merge_groups); DISTINCT aggregates — all-gather of the ranks' distinct values and their merge
 * (llkv_hip_query_merge_distinct).  Every rank ends with the table-wide result.                                    */
llkv_status llkv_hip_query_finish_sharded(llkv_hip_query *query, void *hip_stream);
/* The join → GROUP BY → top-k pipeline over a sharded fact table in one call (steps 2–6 of the phased form below:
 * counts all-reduce on the device, straddler and candidate all-gathers): every rank ends with the same rows.      */
llkv_status llkv_hip_join_agg_finish_sharded(llkv_hip_join_agg *h, uint32_t limit, llkv_join_group_row *out_rows,
                                             uint32_t *out_n, uint64_t *out_total_groups);

/* `configured_thread_count` of the reference's shared pool (llkv-threading/src/lib.rs:13-31): LLKV_MAX_THREADS when
 * it parses to a positive number, else the detected parallelism.  The library's own host threads (staging lanes,
 * per-chunk preparation of column images) are bounded by it.                                                       */
uint32_t llkv_hip_max_threads(void);

/* Planning option, process wide, read when a query is prepared: every f64 SUM / AVG / TOTAL becomes the correctly rounded
 * EXACT sum of the rows' values instead of a fixed-order tree of rounded additions.  The reference adds sequentially
 * (llkv-aggregate/src/lib.rs:870-888): its 15th–16th digit is a property of that order, which no parallel sum shares —
 * and which the qualification rule of llkv-tpch compares (`sum` columns as Decimal::from_f64, 15 significant digits,
 * qualification.rs:672-706).  The exact sum is the order-free answer both approximate; it costs one more state lane per
 * sum.  Needs column statistics that bound the argument (no NaN / ±inf, a smallest non-zero magnitude, |max| / |min
 * non-zero| < 2^9 for the two-lane form); a plan whose argument they do not bound answers LLKV_UNSUPPORTED while the
 * option is on.  Default off.                                                                                        */
void llkv_hip_set_exact_f64_sums(int32_t on);
int32_t llkv_hip_exact_f64_sums(void);

/* Route selection — `QueryExecutor::execute_select_with_filter` llkv-executor/src/lib.rs:523-563: which executor
 * route a SELECT of this shape takes, and whether the GPU path has an entry point for it.                          */
typedef enum llkv_route {
  LLKV_ROUTE_COMPOUND = 1,          /* :531-533 execute_compound_select          — CPU                              */
  LLKV_ROUTE_NO_TABLE = 2,          /* :534-536 execute_select_without_table     — CPU                              */
  LLKV_ROUTE_GROUP_BY = 3,          /* :541-543 execute_group_by_single_table    → llkv_hip_query_prepare_groupby  */
  LLKV_ROUTE_CROSS_PRODUCT = 4,     /* :537-539,544-546 execute_cross_product    → llkv_hip_join_groupby_topk /
                                                                                    llkv_hip_join_stream            */
  LLKV_ROUTE_AGGREGATES = 5,        /* :552-554 execute_aggregates               → llkv_hip_query_prepare_aggregate */
  LLKV_ROUTE_COMPUTED_AGGREGATES = 6, /* :555-557 execute_computed_aggregates    → llkv_hip_query_prepare_aggregate */
  LLKV_ROUTE_PROJECTION = 7         /* :558-560 execute_projection               → llkv_hip_scan_stream            */
} llkv_route;
/* The fields of `SelectPlan` (llkv-plan/src/plans.rs:801-829) the dispatch looks at. */
typedef struct llkv_select_shape {
  int32_t has_compound;                /* plan.compound.is_some()                                                   */
  uint32_t n_tables;                   /* plan.tables.len()                                                         */
  uint32_t n_group_by;                 /* plan.group_by.len()                                                       */
  uint32_t n_aggregates;               /* plan.aggregates.len() (plain `agg(col)` forms)                            */
  int32_t has_computed_aggregates;     /* Self::has_computed_aggregates(&plan): an aggregate inside a projection    */
  uint32_t n_joins;                    /* plan.joins.len(): explicit JOIN … ON between the tables                   */
  int32_t has_having, has_distinct, has_scalar_subqueries; /* SQL breadth the GPU path leaves to the CPU routes     */
} llkv_select_shape;
/* `*route_out`: the route the reference takes.  Returns LLKV_OK when the GPU path serves it, LLKV_UNSUPPORTED when
 * the caller keeps its CPU route (message: why), LLKV_INVALID_ARGUMENT for a shape the reference rejects.          */
llkv_status llkv_hip_select_route(const llkv_select_shape *shape, int32_t *route_out);

/* ------------------------------------------------------------------------- */
/* Plan inspection (host only, no device needed): lowers a plan exactly as the */
/* prepare calls do and returns the kernel plan type.  Used by the build to    */
/* pre-compile the benchmark plans and by tests of the typing rules.           */
/* ------------------------------------------------------------------------- */
typedef struct llkv_column_desc {
  uint32_t field_id;
  int32_t dtype;
  uint64_t rows;      /* global table rows                                   */
  int32_t has_stats;  /* integer min/max known                               */
  int64_t min_i, max_i;
  uint32_t dict_size; /* LLKV_DT_UTF8                                        */
  const char *const *dictionary;
  int32_t nullable;   /* the column has NULL cells                           */
  int32_t precision, scale; /* LLKV_DT_DECIMAL128                            */
  int32_t has_fstats; /* Float64 / Float32: the two statistics below are known */
  double f_absmax;    /* largest |v| over the column's finite values         */
  double f_absmin_nz; /* smallest non-zero |v| over them (0: none)           */
  int32_t f_all_finite; /* … and the column holds no NaN / ±∞                */
} llkv_column_desc;

/* `grouped`: 0 = ungrouped aggregates, 1 = GROUP BY (groups in first-appearance
 * order), 3 = GROUP BY with ORDER BY on the keys (no first-row tracking); + 4 =
 * lower for the shared-image GROUP BY kernel (hundreds … thousands of groups); + 8
 * (with 4) = that lowering in the partitioned route's form (up to 2^24 dense group ids).  */
llkv_status llkv_plan_lower(const llkv_column_desc *cols, uint32_t n_cols,
                            const llkv_filter *filters, uint32_t n_filters,
                            const llkv_eval_op *ops, uint32_t n_ops, const uint32_t *key_fields,
                            uint32_t n_keys, const llkv_aggregate_spec *aggs, uint32_t n_aggs,
                            int32_t grouped, char *type_string_out, uint64_t type_string_cap,
                            uint32_t *lanes_out, uint64_t *bytes_per_row_out);
const char *llkv_plan_last_error(void);
/* The number a Utf8 value counts as under the reference's SQLite-style coercion of aggregate inputs
 * (`s.trim().parse::<f64>().unwrap_or(0.0)`, llkv-aggregate/src/lib.rs:426-434) — what SUM / AVG / TOTAL / MIN / MAX over
 * a dictionary-coded Utf8 column accumulate on the GPU path.  Host only.                                            */
double llkv_plan_parse_numeric(const char *text);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* LLKV_HIP_H */
