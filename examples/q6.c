/* examples/q6.c — TPC-H Q6 through the C ABI alone (no Python, no C++): what a cgo / Rust / JNI binding does.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/q6.c -Lrust-llkv_amd -lllkv_hip -lllkv_tpch -Wl,-rpath,$PWD/rust-llkv_amd -o q6
 *   ./q6 [rows]          (default: SF1, 6 001 215 rows)
 *
 * SELECT sum(l_extendedprice * l_discount) FROM lineitem
 * WHERE l_shipdate >= 8766 AND l_shipdate < 9131 AND l_discount >= 0.05 AND l_discount <= 0.07 AND l_quantity < 24
 */
#include "llkv_hip.h"
#include "llkv_tpch_gen.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(call)                                                                                  \
  do {                                                                                               \
    llkv_status rc_ = (call);                                                                        \
    if (rc_ != LLKV_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, (int)rc_, llkv_hip_last_error()); return 1; } \
  } while (0)

enum { F_QUANTITY = 5, F_EXTENDEDPRICE = 6, F_DISCOUNT = 7, F_SHIPDATE = 11, CHUNK = 131072 };

static llkv_literal lit_i(long long v) { llkv_literal l; memset(&l, 0, sizeof l); l.tag = LLKV_LIT_INT128; l.lo = (uint64_t)v; l.hi = v < 0 ? -1 : 0; return l; }
static llkv_literal lit_f(double v) { llkv_literal l; memset(&l, 0, sizeof l); l.tag = LLKV_LIT_FLOAT64; l.f64 = v; return l; }

int main(int argc, char **argv) {
  const uint64_t rows = argc > 1 ? strtoull(argv[1], NULL, 10) : LLKV_TPCH_LINEITEM_SF1;
  const uint32_t n_chunks = (uint32_t)((rows + CHUNK - 1) / CHUNK);
  uint64_t *chunk_rows = malloc(n_chunks * sizeof *chunk_rows);
  int64_t *qty = malloc(rows * 8);
  double *price = malloc(rows * 8), *disc = malloc(rows * 8);
  int32_t *ship = malloc(rows * 4);
  const void **p_qty = malloc(n_chunks * sizeof(void *)), **p_price = malloc(n_chunks * sizeof(void *));
  const void **p_disc = malloc(n_chunks * sizeof(void *)), **p_ship = malloc(n_chunks * sizeof(void *));
  if (!chunk_rows || !qty || !price || !disc || !ship || !p_qty || !p_price || !p_disc || !p_ship) return 2;
  llkv_tpch_gen_lineitem(LLKV_TPCH_DEFAULT_SEED, (double)rows / 6001215.0, 0, rows, NULL, NULL, NULL, NULL, qty, price, disc, NULL, ship, NULL,
                         NULL, NULL, NULL, 8);
  for (uint32_t c = 0; c < n_chunks; ++c) { /* chunks are views into the column buffers: nothing is copied here */
    const uint64_t lo = (uint64_t)c * CHUNK;
    chunk_rows[c] = rows - lo < CHUNK ? rows - lo : CHUNK;
    p_qty[c] = qty + lo; p_price[c] = price + lo; p_disc[c] = disc + lo; p_ship[c] = ship + lo;
  }

  CHECK(llkv_hip_init(0));
  llkv_hip_table *t = NULL;
  CHECK(llkv_hip_table_create(1, chunk_rows, n_chunks, 0, 1, &t));
  CHECK(llkv_hip_table_append_column(t, F_QUANTITY, LLKV_DT_INT64, p_qty, n_chunks));
  CHECK(llkv_hip_table_append_column(t, F_EXTENDEDPRICE, LLKV_DT_FLOAT64, p_price, n_chunks));
  CHECK(llkv_hip_table_append_column(t, F_DISCOUNT, LLKV_DT_FLOAT64, p_disc, n_chunks));
  CHECK(llkv_hip_table_append_column(t, F_SHIPDATE, LLKV_DT_DATE32, p_ship, n_chunks));

  llkv_filter f[3];
  memset(f, 0, sizeof f);
  f[0].field_id = F_SHIPDATE; f[0].op = LLKV_OP_RANGE;
  f[0].lower_kind = LLKV_BOUND_INCLUDED; f[0].lower = lit_i(LLKV_TPCH_DATE_1994_01_01);
  f[0].upper_kind = LLKV_BOUND_EXCLUDED; f[0].upper = lit_i(LLKV_TPCH_DATE_1995_01_01);
  f[1].field_id = F_DISCOUNT; f[1].op = LLKV_OP_RANGE;
  f[1].lower_kind = LLKV_BOUND_INCLUDED; f[1].lower = lit_f(0.05);
  f[1].upper_kind = LLKV_BOUND_INCLUDED; f[1].upper = lit_f(0.07);
  f[2].field_id = F_QUANTITY; f[2].op = LLKV_OP_LT; f[2].value = lit_i(24);

  llkv_expr_token e[3]; /* l_extendedprice l_discount * */
  memset(e, 0, sizeof e);
  e[0].kind = LLKV_TOK_COLUMN; e[0].field_id = F_EXTENDEDPRICE;
  e[1].kind = LLKV_TOK_COLUMN; e[1].field_id = F_DISCOUNT;
  e[2].kind = LLKV_TOK_BINARY; e[2].binop = LLKV_BIN_MUL;
  llkv_aggregate_spec agg[2];
  memset(agg, 0, sizeof agg);
  agg[0].kind = LLKV_AGG_SUM; agg[0].expr = e; agg[0].expr_len = 3; agg[0].alias = "revenue";
  agg[1].kind = LLKV_AGG_COUNT_STAR; agg[1].alias = "rows";

  llkv_hip_query *q = NULL;
  CHECK(llkv_hip_query_prepare_aggregate(t, f, 3, NULL, 0, agg, 2, &q)); /* no ops: the conjunction of the filters */
  CHECK(llkv_hip_query_launch(q, NULL));
  CHECK(llkv_hip_query_finish(q, NULL));
  llkv_value revenue, n;
  CHECK(llkv_hip_query_value(q, 0, 0, &revenue));
  CHECK(llkv_hip_query_value(q, 0, 1, &n));

  double want = 0.0; /* the same sum on the host, left to right */
  long long want_n = 0;
  for (uint64_t r = 0; r < rows; ++r)
    if (ship[r] >= LLKV_TPCH_DATE_1994_01_01 && ship[r] < LLKV_TPCH_DATE_1995_01_01 && disc[r] >= 0.05 && disc[r] <= 0.07 && qty[r] < 24) {
      want += price[r] * disc[r];
      ++want_n;
    }
  printf("rows %llu  selected %lld (host %lld)  revenue %.6f (host %.6f)\n", (unsigned long long)rows, (long long)n.i64, want_n, revenue.f64, want);
  const double diff = revenue.f64 > want ? revenue.f64 - want : want - revenue.f64;
  const int ok = n.i64 == want_n && diff <= 1e-9 * (want > 0 ? want : 1.0);
  llkv_hip_query_free(q);
  llkv_hip_table_free(t);
  llkv_hip_shutdown();
  free(chunk_rows); free(qty); free(price); free(disc); free(ship); free(p_qty); free(p_price); free(p_disc); free(p_ship);
  puts(ok ? "ok" : "MISMATCH");
  return ok ? 0 : 3;
}
