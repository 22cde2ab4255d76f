/*
 * llkv_tpch_gen.h — deterministic TPC-H-shaped synthetic data (harness side).
 *
 * Stands in for the `tpchgen`-based loader of the reference harness
 * (llkv-tpch/src/lib.rs:243-374), which cannot run offline.  Column types follow
 * the reference's SQL→Arrow mapping (llkv-sql/src/lib.rs:25-28) with the
 * Int64/Float64 column choice of SURVEY.md §8 (Decimal128 cannot be leaf-filtered,
 * llkv-table/src/table.rs:1160-1167).
 *
 * Every value is a pure function of (seed, global row index), so any shard of any
 * table can be generated independently on any rank, and host and device see the
 * same bits.  Pass NULL for columns that are not needed.
 */
#ifndef LLKV_TPCH_GEN_H
#define LLKV_TPCH_GEN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LLKV_TPCH_DEFAULT_SEED 20240607ULL

/* Exact lineitem row counts used by the configs (SF0.01 / SF1 / SF10). */
#define LLKV_TPCH_LINEITEM_SF001 60175ULL
#define LLKV_TPCH_LINEITEM_SF1 6001215ULL
#define LLKV_TPCH_LINEITEM_SF10 59986052ULL

/* Date32 day numbers used by the queries. */
#define LLKV_TPCH_DATE_1992_01_01 8035
#define LLKV_TPCH_DATE_1994_01_01 8766
#define LLKV_TPCH_DATE_1995_01_01 9131
#define LLKV_TPCH_DATE_1995_03_15 9204
#define LLKV_TPCH_DATE_1995_06_17 9298
#define LLKV_TPCH_DATE_1998_09_02 10471

/* Number of orders referenced by the first `lineitem_rows` lineitem rows. */
uint64_t llkv_tpch_orders_for_lineitems(uint64_t lineitem_rows);
/* Customers for a scale factor (150 000 × SF, at least 1). */
uint64_t llkv_tpch_customers_for_scale(double scale);

void llkv_tpch_gen_lineitem(uint64_t seed, double scale, uint64_t row_begin, uint64_t rows,
                            int64_t *l_orderkey, int64_t *l_partkey, int64_t *l_suppkey,
                            int64_t *l_linenumber, int64_t *l_quantity, double *l_extendedprice,
                            double *l_discount, double *l_tax, int32_t *l_shipdate,
                            int32_t *l_commitdate, int32_t *l_receiptdate, uint8_t *l_returnflag,
                            uint8_t *l_linestatus, int32_t threads);

void llkv_tpch_gen_orders(uint64_t seed, double scale, uint64_t row_begin, uint64_t rows,
                          int64_t *o_orderkey, int64_t *o_custkey, int32_t *o_orderdate,
                          int64_t *o_shippriority, int32_t threads);

/* c_mktsegment is a 1-byte code 0..4 into llkv_tpch_segment_name(). */
void llkv_tpch_gen_customer(uint64_t seed, double scale, uint64_t row_begin, uint64_t rows,
                            int64_t *c_custkey, uint8_t *c_mktsegment, int32_t threads);
const char *llkv_tpch_segment_name(uint32_t code);

#ifdef __cplusplus
}
#endif
#endif /* LLKV_TPCH_GEN_H */
