// plan.hpp — host-side lowering of the reference's plan vocabulary (filters, predicate
// program, postfix expressions, aggregate specs; see include/llkv_hip.h) to a kernel plan:
// a C++ type string naming an instantiation of fused_scan_kernel<Plan<...>>, the literal
// banks, the column slots and the lane layout the host needs to finalize.
//
// The lowering restates, on the host, the typing decisions of the reference:
//   literal → native casts            llkv-types/src/literal.rs:364-520
//   leaf predicate typing             llkv-expr/src/typed_predicate.rs:253-312
//   computed projection typing        llkv-compute/src/kernels.rs:179-242, fast_numeric.rs:69-121
//   aggregate admission / accumulators llkv-executor/src/lib.rs:5946-5988, llkv-aggregate/src/lib.rs:463-748
//   GROUP BY argument semantics       llkv-executor/src/lib.rs:7193-7389
#pragma once

#include "llkv_hip.h"

#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace llkv {

struct ColumnInfo {
  uint32_t field_id = 0;
  int32_t dtype = LLKV_DT_NULL;
  uint64_t rows = 0;                    // global row count of the table
  bool has_stats = false;               // integer min/max known (staging statistics)
  int64_t min_i = 0, max_i = 0;
  std::vector<std::string> dictionary;  // LLKV_DT_UTF8: code → string
  int32_t precision = 0, scale = 0;     // LLKV_DT_DECIMAL128 (device image: the raw values narrowed to i64)
  bool nullable = false;                // some cell is NULL (row id absent from the column): a 1 B/row validity mask is staged
  bool f_all_finite = false;            // … and no NaN / ±∞ among the values (unsharded tables; agreed by share_metadata)
  bool f_no_neg_zero = false;           // … and no −0.0: with both, MIN / MAX over the bare column need no row-order lanes (one order-key lane)
  bool has_fstats = false;              // Float64 / Float32 columns, over the finite values (staging statistics):
  double f_absmax = 0.0;                //   largest |v|
  double f_absmin_nz = 0.0;             //   smallest non-zero |v| (0: none / unknown)
  bool ascending = false;               // integer column of an unsharded table: the rows are in strictly ascending value order
  // LLKV_DT_DECIMAL128 with values beyond 64 bits: the device image is two 8 B/row buffers (low halves, high halves) and
  // only SUM / TOTAL / AVG (and the counts, which read no value) take the column; largest |v| as (hi, lo)
  bool wide128 = false;
  uint64_t wide_absmax_hi = 0, wide_absmax_lo = 0;
  uint64_t wide_min_hi = 0, wide_min_lo = 0, wide_max_hi = 0, wide_max_lo = 0; // i128 smallest / largest value of a wide128 column (every cell, NULL ones too)
};

using ColumnResolver = std::function<const ColumnInfo *(uint32_t field_id)>;

enum class AggFinal : int {
  CountRows,   // COUNT(*) / COUNT(col) on a NULL-free column
  SumI64,      // 3 lanes: lo32, hi32, max|v|
  SumI64Fast,  // 1 wrapping lane, overflow excluded by statistics
  SumF64,      // 1 lane
  TotalF64,    // SumF64 lane, never NULL
  AvgI64,      // SumI64 lanes + rows
  AvgI64Fast,
  AvgF64,
  MinI64, MaxI64,
  MinF64, MaxF64,
  CountNullsZero, // COUNT_NULLS on a NULL-free column = 0
  SumDec, TotalDec, AvgDec, MinDec, MaxDec, // Decimal128 column narrowed to i64: SumI64 / SumI64Fast / MinI64 / MaxI64 lanes
  CountValid,     // COUNT(x), x nullable: the valid-row lane
  CountNulls      // COUNT_NULLS(x), x nullable: rows − valid rows
};

struct AggOut {
  AggFinal fin;
  int lane = -1; // first lane of its lane group, relative to the group's lane block
  bool typed_by_first_value = false; // GROUP BY computed argument: the group's temp column takes the type of its first
                                     // non-NULL value, Int64 when there is none (llkv-executor/src/lib.rs:298-406)
  bool fast_sum = false;        // decimal sums: one wrapping lane (statistics exclude i64 overflow) instead of the 96-bit split
  bool wide = false;            // decimal sums over values beyond 64 bits: four lanes, the sums of the 32-bit limbs (SumDecWide)
  bool plain_minmax = false;    // f64 MIN / MAX as ONE order-key lane (the column holds no NaN and no −0.0: no leading NaN to stick, no ±0 tie to break)
  bool null_without_values = false; // SumDec over DISTINCT values: NULL (not 0) when the group has none (SumDistinctDecimal128 finalize)
  int wide_delta = 0;           // MIN / MAX over such values: one MAX_U64 lane of (v − column min) [1] or (column max − v) [2] (MaxWideDelta)
  uint64_t wide_base_hi = 0, wide_base_lo = 0; // … and that column min / max
  int32_t precision = 0, scale = 0; // Decimal128 results
  // GROUP BY computed DECIMAL argument: lane (relative to the group's lane block) whose MIN is (first non-NULL row << 6·n |
  // digit fields), and this argument's field — the digit count of the group's first non-NULL value is the precision of the
  // temp column the reference builds (plan_values_to_arrow_array llkv-executor/src/lib.rs:298-330); a scale above it is an error
  int digits_lane = -1, digits_shift = 0;
  int count_lane = -1; // nullable argument: lane holding the number of non-NULL argument rows (else the group's row lane)
  int exact_levels = 0; // f64 sum kept as exact grid-level lanes (SumF64X, 2 or 3 of them): value = smallest level first, summed
  bool fixed_point = false; // f64 sum kept as an integer count of grid steps 2^fixed_exp (SumF64Q): lanes = low 32 bits, high part
  int fixed_exp = 0;
};

struct LoweredPlan {
  std::string type_string;            // "Plan<Cols<...>,<pred>,Keys<...>,Aggs<...>,U>"
  std::vector<uint32_t> slot_fields;  // slot → field id
  bool late_columns = false;          // register-state plan whose argument-only columns are read for the passing rows (Plan::EARLY)
  std::vector<int32_t> slot_dtypes;   // slot → llkv_dtype (UTF8 = 1-byte codes)
  std::vector<uint8_t> slot_is_valid; // what of the field the slot reads: 0 its values, 1 its validity mask (1 B/row), 2 the high halves of a wide Decimal128 column
  std::vector<int64_t> lit_i;
  std::vector<double> lit_f;
  std::vector<uint32_t> key_fields, key_slots, key_strides, key_cards;
  std::vector<int64_t> key_bases;   // integer keys: code = value - base (column minimum); unused for Utf8
  std::vector<uint8_t> key_is_int;
  std::vector<uint8_t> key_nullable; // the key column has NULL cells: code == cardinality − 1 is the NULL group
  uint32_t ng = 1;
  bool grouped = false;
  bool track_first = false; // lane [1] = first row id of the group (first-appearance order)
  bool acc_lds = false;     // accumulators in per-thread LDS slots (grouped plans)
  bool acc_image = false;   // ONE accumulator image per workgroup in LDS, shared by its threads (hundreds … thousands of groups)
  bool acc_part = false;    // the shared-image lowering for the partitioned route (group_part.cpp): up to 2^24 dense groups, no LDS bound
  int image_passes = 1;     // … the groups cut into this many slices, one scan of the table each
  bool image_cell32 = false; // … with 4-byte cells (counts and bounded integer sums only: Plan::CELL32)
  // shared-image plans: the kernel's image has k_image lanes per group; the fold expands them into the k lanes of the
  // exchange image — exchange lane j of a group = xf(kernel lane image_src[j]): 0 as is, 1 low 32 bits, 2 high part (>> 32)
  int k_image = 0;
  std::vector<uint8_t> image_src, image_xf;
  uint32_t image_min_grid = 0; // workgroups below which an image lane could overflow (fixed-point sums)
  int64_t distinct_field = -1; // reduce plans (sort-based GROUP BY): the column every DISTINCT aggregate is over (-1: none)
  uint32_t distinct_numeric = 0; // what its lanes add: 0 the Int64 / Float64 cell, 1 the numeric image of a Utf8 code, 2 a Boolean's 1.0 / 0.0, 3 a Date32's day number (join.hpp: hj_launch_distinct_heads)
  std::vector<double> distinct_dict_num; // (1) array_value_to_numeric over the dictionary
  // … or a COMPUTED argument (distinct_field stays −1): the projection plan that evaluates it — PlanValue semantics — for the selected
  // rows ("ProjPlan<Cols<…>,Outs<E>>", one Int64 / Float64 output, its validity when some row can be NULL); the values are one more
  // sort key exactly as a column's cells are
  std::shared_ptr<LoweredPlan> distinct_proj;
  std::string distinct_node;
  std::vector<llkv_expr_token> distinct_tokens; // (the argument itself: two DISTINCT aggregates share the sort key only over the same expression)
  bool has_distinct() const { return distinct_field >= 0 || distinct_proj != nullptr; }
  int k = 1;      // lanes per group
  int lanes = 2;  // ng * k + 1
  int unroll = 2;
  std::vector<uint8_t> lane_ops; // size lanes
  std::vector<AggOut> aggs;      // one per requested aggregate
  uint64_t bytes_per_row = 0;    // algorithmic bytes (value buffers, once)
  bool always_false = false;     // predicate folded to FALSE on the host
  bool always_true = false;      // selection plans: predicate folded to TRUE
  std::vector<int32_t> out_dtypes; // projection plans: storage dtype of each output
  std::vector<int32_t> out_fields; // projection plans: source field of a passthrough column, else -1
  std::vector<uint8_t> out_nullable; // projection plans: the output carries a validity bitmap
  // slots whose dictionary codes some aggregate reads as numbers (DictNum<slot>): slot → 256 values (code → f64)
  std::vector<std::pair<int, std::vector<double>>> dict_num;
};

// `str.trim().parse::<f64>().unwrap_or(0.0)` of the reference's numeric coercion (llkv-aggregate/src/lib.rs:426-434):
// Rust's float grammar — sign, decimal digits with an optional point, optional exponent, or inf / infinity / nan in
// any case; no hexadecimal forms, nothing else around the number.
double parse_numeric_or_zero(const std::string &s);

// Lowers a plan.  `grouped` selects the GROUP BY argument semantics (PlanValue
// interpreter) instead of the computed-projection fast path; `track_first` keeps each
// group's first row id (needed unless the output is ordered by the keys).  Returns an llkv_status;
// on failure `err` holds the message.
// `image`: lower for the shared-image GROUP BY kernel (image_scan_body) instead of the per-thread accumulator kernel:
// up to kMaxImageGroups groups as long as the image fits the LDS, every lane order-free — f64 sums need a bound on
// |argument| from the column statistics (else LLKV_UNSUPPORTED: the sort-based route takes the query).
// Planning option: f64 SUM / AVG / TOTAL as the correctly rounded exact sum of the rows' values (llkv_hip_set_exact_f64_sums).
void plan_set_exact_f64_sums(bool on);
bool plan_exact_f64_sums();

int lower_plan(const ColumnResolver &resolve, const llkv_filter *filters, uint32_t n_filters,
               const llkv_eval_op *ops, uint32_t n_ops, const uint32_t *key_fields, uint32_t n_keys,
               const llkv_aggregate_spec *aggs, uint32_t n_aggs, bool grouped, bool track_first,
               LoweredPlan *out, std::string *err, bool image = false, bool partitioned = false);

// Predicate only → "SelPlan<Cols<…>,pred>" (selection-vector kernels, select.hip.h).
// `drop_null_fields`: GatherNullPolicy::DropNulls — rows whose listed fields are ALL NULL are not selected.
int lower_selection(const ColumnResolver &resolve, const llkv_filter *filters, uint32_t n_filters,
                    const llkv_eval_op *ops, uint32_t n_ops, const uint32_t *drop_null_fields, uint32_t n_drop_null_fields,
                    LoweredPlan *out, std::string *err);
// The same with one more conjunct: the integer column `key_field` (no NULL cells) must be in the key set the launch
// binds (InKeySet, fused_scan.hip.h) → "SelPlan<Cols<…>,AndThen<pred,InKeySet<key>>>".
int lower_selection_in_set(const ColumnResolver &resolve, const llkv_filter *filters, uint32_t n_filters, uint32_t key_field,
                           LoweredPlan *out, std::string *err);
// Scan projections (ScanProjection::{Column,Computed}, llkv-scan/src/lib.rs:59-65) →
// "ProjPlan<Cols<…>,Outs<…>>" (window gather + computed expressions).
// `pad_rows`: a row index of ~0 yields a NULL in every output (the build side of a LEFT join)
int lower_projection(const ColumnResolver &resolve, const llkv_projection *projections, uint32_t n_projections,
                     LoweredPlan *out, std::string *err, bool pad_rows = false);

// Fact side of a join → aggregate pipeline → "ProbePlan<Cols<…>,pred,KeyExpr,ValExpr>" (select.hip.h).
// The aggregate argument follows the GROUP BY (PlanValue) semantics and must be Float64.
int lower_probe(const ColumnResolver &resolve, const llkv_filter *filters, uint32_t n_filters, uint32_t key_field,
                const llkv_expr_token *expr, uint32_t expr_len, LoweredPlan *out, std::string *err, bool emit_keybit = false);

// Selected argument values in row order → "EmitPlan<Cols<…>,pred,ValExpr>" (exact SUM(Int64) overflow check).
// `allow_f64`: Float64 arguments are emitted as bit images (DISTINCT aggregates); *is_f64 reports the type.
int lower_emit(const ColumnResolver &resolve, const llkv_filter *filters, uint32_t n_filters, const llkv_eval_op *ops,
               uint32_t n_ops, const llkv_expr_token *expr, uint32_t expr_len, LoweredPlan *out, std::string *err,
               bool allow_f64 = false, bool *is_f64 = nullptr, const uint32_t *in_set_field = nullptr, int32_t *key_dtype = nullptr,
               bool int32_value = false); // (int32_value: a bare Int32 column is emitted sign-extended — the key-bits scan over a key image)

// Aggregates of a sort-based GROUP BY (any number of groups, any state width): "ReducePlan<Cols<…>,Aggs<…>>",
// one wave per group over the group's rows (select.hip.h: group_reduce_body).  Lane layout per group:
// [rows][first row id][aggregate lanes…]; GROUP BY (PlanValue) argument semantics.
int lower_reduce(const ColumnResolver &resolve, const llkv_aggregate_spec *aggs, uint32_t n_aggs, LoweredPlan *out, std::string *err);

// Typed literal cast used by leaf predicates (shared with the selection path).
struct NativeLit {
  bool is_float = false, is_unsigned = false;
  int64_t i = 0;
  double f = 0;
};
int cast_literal_for_column(const llkv_literal &lit, int32_t dtype, NativeLit *out, std::string *err);

const char *dtype_name(int32_t dtype);
const char *dtype_tag(int32_t dtype); // "I64", "F64", ...
uint32_t dtype_width(int32_t dtype);     // bytes per row of the device image
uint32_t dtype_out_width(int32_t dtype); // bytes per row handed to the caller (Decimal128: 16)

} // namespace llkv
