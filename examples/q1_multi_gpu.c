/* examples/q1_multi_gpu.c — TPC-H Q1 over a lineitem table sharded across GPUs, one process per GPU, through the C ABI
 * alone: the collectives (RCCL over xGMI) are inside the library; the host only ships the 128-byte communicator id.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/q1_multi_gpu.c -Lrust-llkv_amd -lllkv_hip -lllkv_tpch -Wl,-rpath,$PWD/rust-llkv_amd -o q1mg
 *   for r in 0 1 2 3 4 5 6 7; do ./q1mg $r 8 /tmp/q1mg.id 59986052 & done; wait        (device ordinal = rank)
 *   ./q1mg 0 1 /tmp/q1mg.id 2000000                                                      (one rank: still goes through RCCL)
 *
 * SELECT l_returnflag, l_linestatus, sum(l_quantity), sum(l_extendedprice), sum(l_extendedprice * (1 - l_discount)),
 *        sum(l_extendedprice * (1 - l_discount) * (1 + l_tax)), avg(l_quantity), avg(l_extendedprice), avg(l_discount), count(*)
 * FROM lineitem WHERE l_shipdate <= DATE '1998-09-02' GROUP BY l_returnflag, l_linestatus ORDER BY 1, 2
 *
 * Every rank prints nothing but rank 0, which prints the groups; all ranks hold the identical result.  The row count of
 * the groups is checked against the ranks' own host loops (gathered with llkv_hip_comm_all_gather_v).
 */
#define _POSIX_C_SOURCE 200809L /* nanosleep */
#include "llkv_hip.h"
#include "llkv_tpch_gen.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define CHECK(call)                                                                                  \
  do {                                                                                               \
    llkv_status rc_ = (call);                                                                        \
    if (rc_ != LLKV_OK) { fprintf(stderr, "[rank %u] %s -> %d: %s\n", rank, #call, (int)rc_, llkv_hip_last_error()); return 1; } \
  } while (0)

enum { F_QUANTITY = 5, F_EXTENDEDPRICE = 6, F_DISCOUNT = 7, F_TAX = 8, F_RETURNFLAG = 9, F_LINESTATUS = 10, F_SHIPDATE = 11, CHUNK = 131072 };

static llkv_literal lit_i(long long v) { llkv_literal l; memset(&l, 0, sizeof l); l.tag = LLKV_LIT_INT128; l.lo = (uint64_t)v; l.hi = v < 0 ? -1 : 0; return l; }
static llkv_expr_token col(uint32_t f) { llkv_expr_token t; memset(&t, 0, sizeof t); t.kind = LLKV_TOK_COLUMN; t.field_id = f; return t; }
static llkv_expr_token num(long long v) { llkv_expr_token t; memset(&t, 0, sizeof t); t.kind = LLKV_TOK_LITERAL; t.literal = lit_i(v); return t; }
static llkv_expr_token op(int b) { llkv_expr_token t; memset(&t, 0, sizeof t); t.kind = LLKV_TOK_BINARY; t.binop = b; return t; }

/* the communicator id travels over whatever the host has: here a file */
static int ship_id(uint8_t id[LLKV_HIP_COMM_ID_BYTES], uint32_t rank, const char *path) {
  char tmp[4096];
  if (rank == 0) {
    snprintf(tmp, sizeof tmp, "%s.tmp", path);
    FILE *f = fopen(tmp, "wb");
    if (!f || fwrite(id, 1, LLKV_HIP_COMM_ID_BYTES, f) != LLKV_HIP_COMM_ID_BYTES) return 1;
    fclose(f);
    return rename(tmp, path);
  }
  for (int tries = 0; tries < 6000; ++tries) { /* ≤ 60 s */
    FILE *f = fopen(path, "rb");
    if (f) {
      const size_t n = fread(id, 1, LLKV_HIP_COMM_ID_BYTES, f);
      fclose(f);
      if (n == LLKV_HIP_COMM_ID_BYTES) return 0;
    }
    struct timespec ts = {0, 10 * 1000 * 1000};
    nanosleep(&ts, NULL);
  }
  return 1;
}

/* one-character Utf8 column as Arrow offsets + data per chunk, the distinct characters seen into `seen` */
static void utf8_chunks(const uint8_t *codes, const uint64_t *chunk_rows, uint32_t n_chunks, int32_t *offsets, const int32_t **p_off, const uint8_t **p_data,
                        int seen[256]) {
  uint64_t lo = 0;
  for (uint32_t c = 0; c < n_chunks; ++c) {
    int32_t *off = offsets + lo + c; /* chunk c has rows + 1 offsets */
    for (uint64_t r = 0; r <= chunk_rows[c]; ++r) off[r] = (int32_t)r;
    for (uint64_t r = 0; r < chunk_rows[c]; ++r) seen[codes[lo + r]] = 1;
    p_off[c] = off;
    p_data[c] = codes + lo;
    lo += chunk_rows[c];
  }
}

int main(int argc, char **argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s rank world id-file [rows [device]]\n", argv[0]); return 2; }
  const uint32_t rank = (uint32_t)atoi(argv[1]), world = (uint32_t)atoi(argv[2]);
  const uint64_t rows = argc > 4 ? strtoull(argv[4], NULL, 10) : LLKV_TPCH_LINEITEM_SF1;
  const int device = argc > 5 ? atoi(argv[5]) : (int)rank;

  /* ---- device and communicator ------------------------------------------------------------------ */
  CHECK(llkv_hip_init(device));
  uint8_t id[LLKV_HIP_COMM_ID_BYTES];
  memset(id, 0, sizeof id);
  if (rank == 0) CHECK(llkv_hip_comm_unique_id(id));
  if (ship_id(id, rank, argv[3])) { fprintf(stderr, "[rank %u] communicator id not delivered\n", rank); return 1; }
  CHECK(llkv_hip_comm_init(id, rank, world)); /* ncclCommInitRank on the bound device */

  /* ---- this rank's shard: whole chunks of the global chunk list --------------------------------- */
  const uint32_t n_global = (uint32_t)((rows + CHUNK - 1) / CHUNK);
  uint64_t *global_rows = malloc(n_global * sizeof *global_rows);
  if (!global_rows) return 2;
  for (uint32_t c = 0; c < n_global; ++c) global_rows[c] = rows - (uint64_t)c * CHUNK < CHUNK ? rows - (uint64_t)c * CHUNK : CHUNK;
  llkv_hip_table *t = NULL;
  CHECK(llkv_hip_table_create(1, global_rows, n_global, rank, world, &t));
  uint32_t first = 0, n_chunks = 0;
  CHECK(llkv_hip_table_local_chunks(t, &first, &n_chunks));
  const uint64_t local = llkv_hip_table_local_rows(t), row_begin = (uint64_t)first * CHUNK;
  int64_t *qty = malloc((local + 1) * 8);
  double *price = malloc((local + 1) * 8), *disc = malloc((local + 1) * 8), *tax = malloc((local + 1) * 8);
  int32_t *ship = malloc((local + 1) * 4), *offsets = malloc((local + n_chunks + 1) * 4);
  uint8_t *flag = malloc(local + 1), *status = malloc(local + 1);
  const void **p = malloc((size_t)(n_chunks + 1) * 8 * sizeof(void *));
  if (!qty || !price || !disc || !tax || !ship || !offsets || !flag || !status || !p) return 2;
  llkv_tpch_gen_lineitem(LLKV_TPCH_DEFAULT_SEED, (double)rows / 6001215.0, row_begin, local, NULL, NULL, NULL, NULL, qty, price, disc, tax, ship, NULL, NULL,
                         flag, status, 8);
  const void **p_qty = p, **p_price = p + n_chunks, **p_disc = p + 2 * n_chunks, **p_tax = p + 3 * n_chunks, **p_ship = p + 4 * n_chunks;
  const int32_t **p_off = (const int32_t **)(p + 5 * n_chunks);
  const uint8_t **p_flag = (const uint8_t **)(p + 6 * n_chunks), **p_status = (const uint8_t **)(p + 7 * n_chunks);
  uint64_t lo = 0;
  for (uint32_t c = 0; c < n_chunks; ++c) {
    p_qty[c] = qty + lo; p_price[c] = price + lo; p_disc[c] = disc + lo; p_tax[c] = tax + lo; p_ship[c] = ship + lo;
    lo += global_rows[first + c];
  }
  CHECK(llkv_hip_table_append_column(t, F_QUANTITY, LLKV_DT_INT64, p_qty, n_chunks));
  CHECK(llkv_hip_table_append_column(t, F_EXTENDEDPRICE, LLKV_DT_FLOAT64, p_price, n_chunks));
  CHECK(llkv_hip_table_append_column(t, F_DISCOUNT, LLKV_DT_FLOAT64, p_disc, n_chunks));
  CHECK(llkv_hip_table_append_column(t, F_TAX, LLKV_DT_FLOAT64, p_tax, n_chunks));
  CHECK(llkv_hip_table_append_column(t, F_SHIPDATE, LLKV_DT_DATE32, p_ship, n_chunks));
  /* Utf8 group keys: every rank must code the strings alike → the table-wide dictionary is the union of the shards' */
  const uint8_t *codes[2] = {flag, status};
  const uint8_t **p_data[2] = {p_flag, p_status};
  const uint32_t fields[2] = {F_RETURNFLAG, F_LINESTATUS};
  for (int k = 0; k < 2; ++k) {
    int seen[256];
    memset(seen, 0, sizeof seen);
    utf8_chunks(codes[k], global_rows + first, n_chunks, offsets, p_off, p_data[k], seen);
    char words[256][2];
    const char *mine[256];
    uint32_t n_mine = 0;
    for (int ch = 0; ch < 256; ++ch)
      if (seen[ch]) { words[n_mine][0] = (char)ch; words[n_mine][1] = 0; mine[n_mine] = words[n_mine]; ++n_mine; }
    char **dict = NULL;
    uint32_t n_dict = 0;
    CHECK(llkv_hip_comm_union_strings(mine, n_mine, &dict, &n_dict));
    CHECK(llkv_hip_table_append_utf8_column(t, fields[k], p_off, p_data[k], n_chunks, (const char *const *)dict, n_dict));
    llkv_hip_free(dict);
  }
  CHECK(llkv_hip_table_share_metadata(t)); /* statistics and nullability agreed by all ranks: the same plan everywhere */

  /* ---- Q1 ------------------------------------------------------------------------------------------ */
  llkv_filter f;
  memset(&f, 0, sizeof f);
  f.field_id = F_SHIPDATE; f.op = LLKV_OP_LE; f.value = lit_i(LLKV_TPCH_DATE_1998_09_02);
  llkv_expr_token e_qty[1] = {col(F_QUANTITY)}, e_price[1] = {col(F_EXTENDEDPRICE)}, e_disc[1] = {col(F_DISCOUNT)};
  llkv_expr_token e_dp[5] = {col(F_EXTENDEDPRICE), num(1), col(F_DISCOUNT), op(LLKV_BIN_SUB), op(LLKV_BIN_MUL)};
  llkv_expr_token e_ch[9] = {col(F_EXTENDEDPRICE), num(1), col(F_DISCOUNT), op(LLKV_BIN_SUB), op(LLKV_BIN_MUL), num(1), col(F_TAX), op(LLKV_BIN_ADD), op(LLKV_BIN_MUL)};
  llkv_aggregate_spec a[8];
  memset(a, 0, sizeof a);
  a[0].kind = LLKV_AGG_SUM; a[0].expr = e_qty; a[0].expr_len = 1;
  a[1].kind = LLKV_AGG_SUM; a[1].expr = e_price; a[1].expr_len = 1;
  a[2].kind = LLKV_AGG_SUM; a[2].expr = e_dp; a[2].expr_len = 5;
  a[3].kind = LLKV_AGG_SUM; a[3].expr = e_ch; a[3].expr_len = 9;
  a[4].kind = LLKV_AGG_AVG; a[4].expr = e_qty; a[4].expr_len = 1;
  a[5].kind = LLKV_AGG_AVG; a[5].expr = e_price; a[5].expr_len = 1;
  a[6].kind = LLKV_AGG_AVG; a[6].expr = e_disc; a[6].expr_len = 1;
  a[7].kind = LLKV_AGG_COUNT_STAR;
  llkv_hip_query *q = NULL;
  CHECK(llkv_hip_query_prepare_groupby(t, &f, 1, NULL, 0, fields, 2, a, 8, 1, &q));
  CHECK(llkv_hip_query_launch(q, NULL));
  CHECK(llkv_hip_query_finish_sharded(q, NULL)); /* one ncclAllReduce(int64, sum) of the exchange image, then the host fold */

  /* ---- check: Σ count(*) over the groups = rows that pass the filter on all ranks ----------------------- */
  uint64_t mine_pass = 0;
  for (uint64_t r = 0; r < local; ++r) mine_pass += ship[r] <= LLKV_TPCH_DATE_1998_09_02;
  void *all = NULL;
  uint64_t *bounds = malloc((world + 1) * sizeof *bounds);
  if (!bounds) return 2;
  CHECK(llkv_hip_comm_all_gather_v(&mine_pass, sizeof mine_pass, &all, bounds));
  uint64_t want = 0, got = 0;
  for (uint32_t r = 0; r < world; ++r) want += ((const uint64_t *)all)[r];
  llkv_hip_free(all);
  const uint32_t groups = llkv_hip_query_num_groups(q);
  for (uint32_t g = 0; g < groups; ++g) {
    llkv_value k0, k1, v[8];
    CHECK(llkv_hip_query_group_key(q, g, 0, &k0));
    CHECK(llkv_hip_query_group_key(q, g, 1, &k1));
    for (uint32_t i = 0; i < 8; ++i) CHECK(llkv_hip_query_value(q, g, i, &v[i]));
    got += (uint64_t)v[7].i64;
    if (rank == 0)
      printf("%s %s  sum_qty %lld  sum_price %.2f  sum_disc_price %.4f  sum_charge %.6f  avg_qty %.6f  avg_price %.6f  avg_disc %.6f  count %lld\n", k0.str, k1.str,
             (long long)v[0].i64, v[1].f64, v[2].f64, v[3].f64, v[4].f64, v[5].f64, v[6].f64, (long long)v[7].i64);
  }
  llkv_hip_query_free(q);
  llkv_hip_table_free(t);
  llkv_hip_comm_destroy();
  if (got != want) { fprintf(stderr, "[rank %u] %llu rows in the groups, %llu pass the filter\n", rank, (unsigned long long)got, (unsigned long long)want); return 1; }
  if (rank == 0) printf("%u rank(s), %llu rows, %u groups: ok\n", world, (unsigned long long)rows, groups);
  return 0;
}
