/* examples/q3_join.c — the join boundary through the C ABI alone (no Python, no C++): what a cgo / Rust / JNI binding does.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/q3_join.c -Lrust-llkv_amd -lllkv_hip -lllkv_tpch -Wl,-rpath,$PWD/rust-llkv_amd -o q3_join
 *   ./q3_join [lineitem rows]          (default 600 000: scale 0.1)
 *
 * 1. TableJoinExt::join_stream (llkv-join/src/lib.rs:240-282) delivering the reference's joined RecordBatches:
 *      lineitem ⋈ orders ON l_orderkey = o_orderkey, columns (l_orderkey, l_extendedprice | o_orderkey, o_orderdate)
 *    — every batch is checked on the host (keys equal, the date is the order's), the totals against a host join.
 * 2. TPC-H Q3 (llkv_hip_join_groupby_topk): customer(BUILDING) ⋉ orders(< 1995-03-15) ⋈ lineitem(> 1995-03-15),
 *    GROUP BY the order, SUM(l_extendedprice · (1 − l_discount)), top 10 — against the same query on the host,
 *    revenue bit for bit (both sides add an order's lines in row order).
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

enum { L_ORDERKEY = 1, L_EXTENDEDPRICE = 6, L_DISCOUNT = 7, L_SHIPDATE = 11, O_ORDERKEY = 1, O_CUSTKEY = 2, O_ORDERDATE = 3, O_SHIPPRIORITY = 4,
       C_CUSTKEY = 1, C_MKTSEGMENT = 2, CHUNK = 65536, DATE_1995_03_15 = 9204 };

static llkv_literal lit_i(long long v) { llkv_literal l; memset(&l, 0, sizeof l); l.tag = LLKV_LIT_INT128; l.lo = (uint64_t)v; l.hi = v < 0 ? -1 : 0; return l; }

/* one table of `rows` rows in chunks of CHUNK: the chunk list and, per column, the chunk pointers into one buffer */
typedef struct chunks { uint32_t n; uint64_t *rows; } chunks;
static chunks make_chunks(uint64_t rows) {
  chunks c;
  c.n = (uint32_t)((rows + CHUNK - 1) / CHUNK);
  c.rows = malloc((c.n ? c.n : 1) * sizeof *c.rows);
  for (uint32_t i = 0; i < c.n; ++i) c.rows[i] = rows - (uint64_t)i * CHUNK < CHUNK ? rows - (uint64_t)i * CHUNK : CHUNK;
  return c;
}
static const void **views(const chunks *c, const void *base, size_t width) {
  const void **p = malloc((c->n ? c->n : 1) * sizeof *p);
  for (uint32_t i = 0; i < c->n; ++i) p[i] = (const char *)base + (size_t)i * CHUNK * width;
  return p;
}

/* ---- 1. the joined batches ---- */
typedef struct seen { uint64_t rows, batches, bad; double price_sum; const int64_t *o_key; const int32_t *o_date; uint64_t n_orders; } seen;
static void on_batch(const llkv_batch_view *b, const char *const *names, void *user) {
  seen *s = user;
  s->batches += 1;
  s->rows += b->num_rows;
  if (b->num_columns != 4 || strcmp(names[0], "l_orderkey") || strcmp(names[2], "o_orderkey") || strcmp(names[3], "o_orderdate")) { s->bad += 1; return; }
  const int64_t *lk = b->columns[0].values, *ok = b->columns[2].values;
  const double *price = b->columns[1].values;
  const int32_t *date = b->columns[3].values;
  for (uint64_t r = 0; r < b->num_rows; ++r) {
    s->price_sum += price[r];
    /* the generator's order keys ascend: the order of a key by bisection */
    uint64_t lo = 0, hi = s->n_orders;
    while (hi - lo > 1) { const uint64_t mid = (lo + hi) / 2; if (s->o_key[mid] <= lk[r]) lo = mid; else hi = mid; }
    if (lk[r] != ok[r] || s->o_key[lo] != lk[r] || s->o_date[lo] != date[r]) s->bad += 1;
  }
}

typedef struct host_group { int64_t key; double sum; int32_t date; } host_group;
static int by_revenue(const void *a, const void *b) {
  const host_group *x = a, *y = b;
  if (x->sum != y->sum) return x->sum > y->sum ? -1 : 1;
  if (x->date != y->date) return x->date < y->date ? -1 : 1;
  return x->key < y->key ? -1 : x->key > y->key;
}

int main(int argc, char **argv) {
  const uint64_t n_li = argc > 1 ? strtoull(argv[1], NULL, 10) : 600000;
  const double scale = (double)n_li / 6001215.0;
  const uint64_t n_ord = llkv_tpch_orders_for_lineitems(n_li), n_cust = llkv_tpch_customers_for_scale(scale);
  int64_t *l_key = malloc(n_li * 8), *o_key = malloc(n_ord * 8), *o_cust = malloc(n_ord * 8), *o_prio = malloc(n_ord * 8), *c_key = malloc(n_cust * 8);
  double *l_price = malloc(n_li * 8), *l_disc = malloc(n_li * 8);
  int32_t *l_ship = malloc(n_li * 4), *o_date = malloc(n_ord * 4);
  uint8_t *c_seg = malloc(n_cust);
  if (!l_key || !o_key || !o_cust || !o_prio || !c_key || !l_price || !l_disc || !l_ship || !o_date || !c_seg) return 2;
  llkv_tpch_gen_lineitem(LLKV_TPCH_DEFAULT_SEED, scale, 0, n_li, l_key, NULL, NULL, NULL, NULL, l_price, l_disc, NULL, l_ship, NULL, NULL, NULL, NULL, 8);
  llkv_tpch_gen_orders(LLKV_TPCH_DEFAULT_SEED, scale, 0, n_ord, o_key, o_cust, o_date, o_prio, 8);
  llkv_tpch_gen_customer(LLKV_TPCH_DEFAULT_SEED, scale, 0, n_cust, c_key, c_seg, 8);

  CHECK(llkv_hip_init(0));
  chunks cl = make_chunks(n_li), co = make_chunks(n_ord), cc = make_chunks(n_cust);
  llkv_hip_table *lt = NULL, *ot = NULL, *ct = NULL;
  CHECK(llkv_hip_table_create(1, cl.rows, cl.n, 0, 1, &lt));
  CHECK(llkv_hip_table_create(2, co.rows, co.n, 0, 1, &ot));
  CHECK(llkv_hip_table_create(3, cc.rows, cc.n, 0, 1, &ct));
  const void **p;
  CHECK(llkv_hip_table_append_column(lt, L_ORDERKEY, LLKV_DT_INT64, p = views(&cl, l_key, 8), cl.n)); free(p);
  CHECK(llkv_hip_table_append_column(lt, L_EXTENDEDPRICE, LLKV_DT_FLOAT64, p = views(&cl, l_price, 8), cl.n)); free(p);
  CHECK(llkv_hip_table_append_column(lt, L_DISCOUNT, LLKV_DT_FLOAT64, p = views(&cl, l_disc, 8), cl.n)); free(p);
  CHECK(llkv_hip_table_append_column(lt, L_SHIPDATE, LLKV_DT_DATE32, p = views(&cl, l_ship, 4), cl.n)); free(p);
  CHECK(llkv_hip_table_append_column(ot, O_ORDERKEY, LLKV_DT_INT64, p = views(&co, o_key, 8), co.n)); free(p);
  CHECK(llkv_hip_table_append_column(ot, O_CUSTKEY, LLKV_DT_INT64, p = views(&co, o_cust, 8), co.n)); free(p);
  CHECK(llkv_hip_table_append_column(ot, O_ORDERDATE, LLKV_DT_DATE32, p = views(&co, o_date, 4), co.n)); free(p);
  CHECK(llkv_hip_table_append_column(ot, O_SHIPPRIORITY, LLKV_DT_INT64, p = views(&co, o_prio, 8), co.n)); free(p);
  CHECK(llkv_hip_table_append_column(ct, C_CUSTKEY, LLKV_DT_INT64, p = views(&cc, c_key, 8), cc.n)); free(p);
  { /* c_mktsegment as a Utf8 column: Arrow offsets + data per chunk, the segment names as the strings */
    const int32_t **offs = malloc(cc.n * sizeof *offs);
    const uint8_t **data = malloc(cc.n * sizeof *data);
    for (uint32_t i = 0; i < cc.n; ++i) {
      int32_t *o = malloc((cc.rows[i] + 1) * 4);
      size_t bytes = 0;
      for (uint64_t r = 0; r < cc.rows[i]; ++r) bytes += strlen(llkv_tpch_segment_name(c_seg[(uint64_t)i * CHUNK + r]));
      uint8_t *d = malloc(bytes ? bytes : 1);
      o[0] = 0;
      for (uint64_t r = 0; r < cc.rows[i]; ++r) {
        const char *s = llkv_tpch_segment_name(c_seg[(uint64_t)i * CHUNK + r]);
        memcpy(d + o[r], s, strlen(s));
        o[r + 1] = o[r] + (int32_t)strlen(s);
      }
      offs[i] = o; data[i] = d;
    }
    CHECK(llkv_hip_table_append_utf8_column(ct, C_MKTSEGMENT, offs, data, cc.n, NULL, 0));
    for (uint32_t i = 0; i < cc.n; ++i) { free((void *)offs[i]); free((void *)data[i]); }
    free(offs); free(data);
  }

  /* ---- 1. lineitem ⋈ orders as joined RecordBatches ---- */
  const llkv_join_key key = {L_ORDERKEY, O_ORDERKEY, 0};
  llkv_join_options opt;
  memset(&opt, 0, sizeof opt);
  opt.join_type = LLKV_JOIN_INNER;
  opt.batch_size = 8192;
  const llkv_join_column lcols[2] = {{L_ORDERKEY, "l_orderkey"}, {L_EXTENDEDPRICE, "l_extendedprice"}};
  const llkv_join_column rcols[2] = {{O_ORDERKEY, "o_orderkey"}, {O_ORDERDATE, "o_orderdate"}};
  const llkv_join_output out = {lcols, 2, rcols, 2};
  seen s;
  memset(&s, 0, sizeof s);
  s.o_key = o_key; s.o_date = o_date; s.n_orders = n_ord;
  CHECK(llkv_hip_join_stream_batches(lt, ot, &key, 1, &opt, &out, on_batch, &s));
  double want_price = 0.0;
  for (uint64_t r = 0; r < n_li; ++r) want_price += l_price[r]; /* every lineitem has its order: the join keeps them all, in row order */
  printf("join: %llu rows in %llu batches, %llu bad cells, price sum %.2f (host %.2f)\n", (unsigned long long)s.rows, (unsigned long long)s.batches,
         (unsigned long long)s.bad, s.price_sum, want_price);
  int ok = s.rows == n_li && s.bad == 0 && s.price_sum == want_price;

  /* ---- 2. Q3 ---- */
  llkv_filter ff, fd, fc;
  memset(&ff, 0, sizeof ff); memset(&fd, 0, sizeof fd); memset(&fc, 0, sizeof fc);
  ff.field_id = L_SHIPDATE; ff.op = LLKV_OP_GT; ff.value = lit_i(DATE_1995_03_15);
  fd.field_id = O_ORDERDATE; fd.op = LLKV_OP_LT; fd.value = lit_i(DATE_1995_03_15);
  fc.field_id = C_MKTSEGMENT; fc.op = LLKV_OP_EQUALS; fc.value.tag = LLKV_LIT_STRING; fc.value.str = "BUILDING";
  const llkv_join_side fact = {lt, &ff, 1, L_ORDERKEY}, dim = {ot, &fd, 1, O_ORDERKEY}, dim2 = {ct, &fc, 1, C_CUSTKEY};
  llkv_expr_token e[5]; /* l_extendedprice 1 l_discount − × */
  memset(e, 0, sizeof e);
  e[0].kind = LLKV_TOK_COLUMN; e[0].field_id = L_EXTENDEDPRICE;
  e[1].kind = LLKV_TOK_LITERAL; e[1].literal = lit_i(1);
  e[2].kind = LLKV_TOK_COLUMN; e[2].field_id = L_DISCOUNT;
  e[3].kind = LLKV_TOK_BINARY; e[3].binop = LLKV_BIN_SUB;
  e[4].kind = LLKV_TOK_BINARY; e[4].binop = LLKV_BIN_MUL;
  const uint32_t payload[2] = {O_ORDERDATE, O_SHIPPRIORITY};
  llkv_join_group_row top[10];
  uint32_t n_top = 0;
  uint64_t groups = 0;
  CHECK(llkv_hip_join_groupby_topk(&fact, &dim, O_CUSTKEY, &dim2, payload, 2, e, 5, 10, top, &n_top, &groups));

  /* the same on the host: customers of the segment → orders before the date → their lines shipped after it, in row order */
  uint8_t *building = calloc(n_cust + 2, 1);
  for (uint64_t r = 0; r < n_cust; ++r) if (!strcmp(llkv_tpch_segment_name(c_seg[r]), "BUILDING")) building[c_key[r]] = 1; /* customer keys are 1 … n */
  host_group *hg = malloc(n_ord * sizeof *hg);
  uint64_t n_hg = 0, li = 0;
  for (uint64_t o = 0; o < n_ord; ++o) { /* lineitem is clustered by the ascending order key */
    while (li < n_li && l_key[li] < o_key[o]) ++li;
    const int wanted = o_date[o] < DATE_1995_03_15 && o_cust[o] >= 1 && (uint64_t)o_cust[o] <= n_cust && building[o_cust[o]];
    double sum = 0.0;
    uint64_t cnt = 0;
    for (; li < n_li && l_key[li] == o_key[o]; ++li)
      if (wanted && l_ship[li] > DATE_1995_03_15) { sum += l_price[li] * (1 - l_disc[li]); ++cnt; }
    if (cnt) { hg[n_hg].key = o_key[o]; hg[n_hg].sum = sum; hg[n_hg].date = o_date[o]; ++n_hg; }
  }
  qsort(hg, n_hg, sizeof *hg, by_revenue);
  printf("q3: %llu groups (host %llu); top %u\n", (unsigned long long)groups, (unsigned long long)n_hg, n_top);
  ok = ok && groups == n_hg && n_top == (n_hg < 10 ? n_hg : 10);
  for (uint32_t i = 0; i < n_top && ok; ++i) {
    printf("  %10lld  %.4f  %d\n", (long long)top[i].key, top[i].sum, (int)top[i].payload[0]);
    ok = top[i].key == hg[i].key && memcmp(&top[i].sum, &hg[i].sum, 8) == 0 && top[i].payload[0] == hg[i].date;
  }

  /* ---- the same star with an aggregate LIST — sum(revenue), count(*), min(l_extendedprice) — ordered by count desc, key asc:
   *      llkv_hip_join_groupby_prepare → launch → finish → llkv_hip_join_groupby_rows (the general route; the call above is the
   *      hand-tuned single-SUM shape).  Counts, minima, keys and payload are exact; the f64 sum is a tree per group (1e-9). */
  llkv_expr_token price[1];
  memset(price, 0, sizeof price);
  price[0].kind = LLKV_TOK_COLUMN; price[0].field_id = L_EXTENDEDPRICE;
  llkv_aggregate_spec aggs[3];
  memset(aggs, 0, sizeof aggs);
  aggs[0].kind = LLKV_AGG_SUM; aggs[0].expr = e; aggs[0].expr_len = 5;
  aggs[1].kind = LLKV_AGG_COUNT_STAR;
  aggs[2].kind = LLKV_AGG_MIN; aggs[2].expr = price; aggs[2].expr_len = 1;
  llkv_hip_query *jq = NULL;
  CHECK(llkv_hip_join_groupby_prepare(&fact, &dim, O_CUSTKEY, &dim2, aggs, 3, &jq));
  CHECK(llkv_hip_query_launch(jq, NULL));
  CHECK(llkv_hip_query_finish(jq, NULL));
  llkv_join_order_key order[2];
  memset(order, 0, sizeof order);
  order[0].kind = LLKV_JOIN_ORDER_AGGREGATE; order[0].index = 1; order[0].descending = 1;
  order[1].kind = LLKV_JOIN_ORDER_KEY;
  llkv_hip_join_rows *jr = NULL;
  CHECK(llkv_hip_join_groupby_rows(jq, payload, 2, order, 2, 5, &jr));
  ok = ok && llkv_hip_join_rows_total_groups(jr) == n_hg && llkv_hip_join_rows_len(jr) == (n_hg < 5 ? n_hg : 5);
  for (uint64_t i = 0; i < llkv_hip_join_rows_len(jr) && ok; ++i) {
    int64_t key, pay[4];
    uint8_t pay_null[4];
    uint64_t pos;
    const llkv_value *v;
    CHECK(llkv_hip_join_rows_get(jr, i, &key, pay, pay_null, &pos, &v));
    /* the host's figures for this order: count, minimum price, sum of the revenue */
    uint64_t cnt = 0, h = 0;
    double mn = 0.0, sum = 0.0;
    while (h < n_hg && hg[h].key != key) ++h;
    for (uint64_t r = 0; r < n_li; ++r)
      if (l_key[r] == key && l_ship[r] > DATE_1995_03_15) { mn = cnt == 0 || l_price[r] < mn ? l_price[r] : mn; sum += l_price[r] * (1 - l_disc[r]); ++cnt; }
    printf("  general: %10lld  count %llu  min %.2f  sum %.4f  date %d\n", (long long)key, (unsigned long long)v[1].i64, v[2].f64, v[0].f64, (int)pay[0]);
    ok = h < n_hg && (uint64_t)v[1].i64 == cnt && v[2].f64 == mn && !pay_null[0] && pay[0] == hg[h].date &&
         (v[0].f64 - sum <= 1e-9 * sum && sum - v[0].f64 <= 1e-9 * sum);
  }
  llkv_hip_join_rows_free(jr);
  llkv_hip_query_free(jq);

  llkv_hip_table_free(lt); llkv_hip_table_free(ot); llkv_hip_table_free(ct);
  llkv_hip_shutdown();
  puts(ok ? "ok" : "MISMATCH");
  return ok ? 0 : 3;
}
