/*
 * llkv_oracle_fast.c — "best-effort parallel" CPU mode of the oracle (BASELINE.md §2,
 * mode 2).  TEST / BASELINE INFRASTRUCTURE ONLY.
 *
 * Same arithmetic per row as llkv_oracle.c (typed predicates of
 * llkv-expr/src/typed_predicate.rs:75-146, the final-type evaluation order of
 * llkv-compute/src/fast_numeric.rs:69-121, the accumulators of
 * llkv-aggregate/src/lib.rs:759-1477) but fused and chunk-parallel: each worker walks
 * whole chunks of 131 072 rows (llkv-column-map/src/store/constants.rs:22) in blocks of
 * 2 048 rows, partial states are combined in chunk order.  Restricted to what the
 * benchmark queries need: conjunction of leaf filters on non-NULL columns,
 * SUM/COUNT/AVG/MIN/MAX over i64/f64 expressions, ungrouped or grouped by up to two
 * one-character Utf8 columns (TPC-H Q1's flags: dense group ids from the byte values).
 */
#define _GNU_SOURCE
#include "llkv_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define CHUNK_ROWS 131072u
#define BLOCK 2048u

typedef __int128 i128;
typedef unsigned __int128 u128;

typedef struct fpred {
  const void *values;
  int32_t dtype;
  int has_lo, lo_incl, has_hi, hi_incl, is_eq;
  int64_t ilo, ihi;
  double dlo, dhi;
} fpred;

typedef struct fagg {
  int32_t kind;
  int is_f64;
  const llkv_expr_token *expr;
  uint32_t n;
} fagg;

typedef struct fstate { /* per aggregate partial */
  double fsum;
  i128 isum;
  int64_t count;
  double fmin, fmax;
  int64_t imin, imax;
  int has;
} fstate;

typedef struct fjob {
  const orc_table *t;
  const fpred *preds;
  uint32_t n_preds;
  const fagg *aggs;
  uint32_t n_aggs;
  uint64_t n_chunks;
  /* grouped: key k of row r has the dense code code[k][key_data[k][r]]; group = code0 * card1 + code1 */
  uint32_t n_keys, ng, card1;
  const uint8_t *key_data[2];
  uint8_t code[2][256];
  fstate *partials; /* [n_chunks][ng][n_aggs] */
  uint64_t next_chunk;
  pthread_mutex_t mu;
} fjob;

static const orc_column *fcol(const orc_table *t, uint32_t fid) {
  for (uint32_t i = 0; i < t->n_cols; ++i) if (t->cols[i].field_id == fid) return &t->cols[i];
  return NULL;
}
static i128 flit(const llkv_literal *l) { return (i128)(((u128)(uint64_t)l->hi << 64) | (u128)l->lo); }

static void pred_block(const fpred *p, uint64_t base, uint32_t n, uint8_t *mask) {
#define LOOP(T, LO, HI, EQ)                                                                       \
  do {                                                                                            \
    const T *v = (const T *)p->values + base;                                                     \
    for (uint32_t i = 0; i < n; ++i) {                                                            \
      T x = v[i];                                                                                 \
      uint8_t ok = 1;                                                                             \
      if (p->is_eq) ok = x == (T)(EQ);                                                            \
      if (p->has_lo) ok &= p->lo_incl ? x >= (T)(LO) : x > (T)(LO);                               \
      if (p->has_hi) ok &= p->hi_incl ? x <= (T)(HI) : x < (T)(HI);                               \
      mask[i] &= ok;                                                                              \
    }                                                                                             \
  } while (0)
  switch (p->dtype) {
  case LLKV_DT_INT64: LOOP(int64_t, p->ilo, p->ihi, p->ilo); break;
  case LLKV_DT_INT32: case LLKV_DT_DATE32: LOOP(int32_t, p->ilo, p->ihi, p->ilo); break;
  case LLKV_DT_FLOAT64: LOOP(double, p->dlo, p->dhi, p->dlo); break;
  default: memset(mask, 0, n); break;
  }
#undef LOOP
}

/* evaluate a postfix program over a block in the final type (f64 or i64) */
static void eval_block(const orc_table *t, const fagg *a, uint64_t base, uint32_t n, double (*fst)[BLOCK], int64_t (*ist)[BLOCK]) {
  uint32_t sp = 0;
  for (uint32_t k = 0; k < a->n; ++k) {
    const llkv_expr_token *e = &a->expr[k];
    if (e->kind == LLKV_TOK_COLUMN) {
      const orc_column *c = fcol(t, e->field_id);
      if (a->is_f64) {
        double *d = fst[sp];
        if (c->dtype == LLKV_DT_FLOAT64) memcpy(d, (const double *)c->values + base, n * 8);
        else if (c->dtype == LLKV_DT_INT64) for (uint32_t i = 0; i < n; ++i) d[i] = (double)((const int64_t *)c->values)[base + i];
        else for (uint32_t i = 0; i < n; ++i) d[i] = (double)((const int32_t *)c->values)[base + i];
      } else {
        int64_t *d = ist[sp];
        if (c->dtype == LLKV_DT_INT64) memcpy(d, (const int64_t *)c->values + base, n * 8);
        else for (uint32_t i = 0; i < n; ++i) d[i] = ((const int32_t *)c->values)[base + i];
      }
      ++sp;
    } else if (e->kind == LLKV_TOK_LITERAL) {
      if (a->is_f64) { double v = e->literal.tag == LLKV_LIT_FLOAT64 ? e->literal.f64 : (double)flit(&e->literal); for (uint32_t i = 0; i < n; ++i) fst[sp][i] = v; }
      else { int64_t v = e->literal.tag == LLKV_LIT_FLOAT64 ? (int64_t)e->literal.f64 : (int64_t)flit(&e->literal); for (uint32_t i = 0; i < n; ++i) ist[sp][i] = v; }
      ++sp;
    } else {
      --sp;
      if (a->is_f64) {
        double *l = fst[sp - 1], *r = fst[sp];
        switch (e->binop) {
        case LLKV_BIN_ADD: for (uint32_t i = 0; i < n; ++i) l[i] = l[i] + r[i]; break;
        case LLKV_BIN_SUB: for (uint32_t i = 0; i < n; ++i) l[i] = l[i] - r[i]; break;
        case LLKV_BIN_MUL: for (uint32_t i = 0; i < n; ++i) l[i] = l[i] * r[i]; break;
        default: break;
        }
      } else {
        int64_t *l = ist[sp - 1], *r = ist[sp];
        switch (e->binop) { /* wrapping here; the sequential oracle is the overflow authority */
        case LLKV_BIN_ADD: for (uint32_t i = 0; i < n; ++i) l[i] = (int64_t)((uint64_t)l[i] + (uint64_t)r[i]); break;
        case LLKV_BIN_SUB: for (uint32_t i = 0; i < n; ++i) l[i] = (int64_t)((uint64_t)l[i] - (uint64_t)r[i]); break;
        case LLKV_BIN_MUL: for (uint32_t i = 0; i < n; ++i) l[i] = (int64_t)((uint64_t)l[i] * (uint64_t)r[i]); break;
        default: break;
        }
      }
    }
  }
}

static void *worker(void *arg) {
  fjob *j = arg;
  double (*fst)[BLOCK] = malloc(sizeof(double) * BLOCK * 8);
  int64_t (*ist)[BLOCK] = malloc(sizeof(int64_t) * BLOCK * 8);
  uint8_t *mask = malloc(BLOCK);
  uint16_t *gid = malloc(BLOCK * sizeof(uint16_t));
  for (;;) {
    pthread_mutex_lock(&j->mu);
    uint64_t c = j->next_chunk++;
    pthread_mutex_unlock(&j->mu);
    if (c >= j->n_chunks) break;
    fstate *st0 = &j->partials[c * j->ng * j->n_aggs];
    uint64_t r0 = c * CHUNK_ROWS, r1 = r0 + CHUNK_ROWS < j->t->rows ? r0 + CHUNK_ROWS : j->t->rows;
    for (uint64_t b = r0; b < r1; b += BLOCK) {
      uint32_t n = (uint32_t)(r1 - b < BLOCK ? r1 - b : BLOCK);
      memset(mask, 1, n);
      for (uint32_t p = 0; p < j->n_preds; ++p) pred_block(&j->preds[p], b, n, mask);
      uint32_t any = 0;
      for (uint32_t i = 0; i < n; ++i) any += mask[i];
      if (!any) continue;
      if (j->n_keys) {
        for (uint32_t i = 0; i < n; ++i) {
          uint32_t g = j->code[0][j->key_data[0][b + i]];
          if (j->n_keys > 1) g = g * j->card1 + j->code[1][j->key_data[1][b + i]];
          gid[i] = (uint16_t)g;
        }
      }
      const uint32_t na = j->n_aggs;
      for (uint32_t a = 0; a < na; ++a) {
        const fagg *ag = &j->aggs[a];
        if (ag->kind == LLKV_AGG_COUNT_STAR) {
          if (!j->n_keys) st0[a].count += any;
          else for (uint32_t i = 0; i < n; ++i) if (mask[i]) st0[gid[i] * na + a].count++;
          continue;
        }
        eval_block(j->t, ag, b, n, fst, ist);
        if (ag->is_f64) {
          const double *v = fst[0];
          for (uint32_t i = 0; i < n; ++i) if (mask[i]) {
            fstate *s = &st0[(j->n_keys ? gid[i] : 0) * na + a];
            double x = v[i];
            s->fsum += x; s->count++;
            if (!s->has) { s->fmin = s->fmax = x; s->has = 1; } else { if (x < s->fmin) s->fmin = x; if (x > s->fmax) s->fmax = x; }
          }
        } else {
          const int64_t *v = ist[0];
          for (uint32_t i = 0; i < n; ++i) if (mask[i]) {
            fstate *s = &st0[(j->n_keys ? gid[i] : 0) * na + a];
            int64_t x = v[i];
            s->isum += x; s->fsum += (double)x; s->count++;
            if (!s->has) { s->imin = s->imax = x; s->has = 1; } else { if (x < s->imin) s->imin = x; if (x > s->imax) s->imax = x; }
          }
        }
      }
    }
  }
  free(fst); free(ist); free(mask); free(gid);
  return NULL;
}

static __thread char g_ferr[256];

static int32_t run_parallel(const orc_table *t, const llkv_filter *filters, uint32_t n_filters, const uint32_t *key_fields, uint32_t n_keys,
                            const llkv_aggregate_spec *aggs, uint32_t n_aggs, llkv_value *out_values, uint8_t *out_keys, uint32_t *out_groups,
                            uint32_t max_groups, int32_t threads) {
  (void)g_ferr;
  fpred *preds = calloc(n_filters ? n_filters : 1, sizeof(fpred));
  fagg *fa = calloc(n_aggs ? n_aggs : 1, sizeof(fagg));
  for (uint32_t i = 0; i < n_filters; ++i) {
    const llkv_filter *f = &filters[i];
    const orc_column *c = fcol(t, f->field_id);
    if (!c || c->validity) { free(preds); free(fa); return LLKV_UNSUPPORTED; }
    fpred *p = &preds[i];
    p->values = c->values; p->dtype = c->dtype;
    int isf = c->dtype == LLKV_DT_FLOAT64;
#define LITF(l) ((l).tag == LLKV_LIT_FLOAT64 ? (l).f64 : (l).tag == LLKV_LIT_DECIMAL128 ? (double)flit(&(l)) / pow(10.0, (l).scale) : (double)flit(&(l)))
#define SETLO(l, incl) do { p->has_lo = 1; p->lo_incl = incl; if (isf) p->dlo = LITF(l); else p->ilo = (int64_t)flit(&(l)); } while (0)
#define SETHI(l, incl) do { p->has_hi = 1; p->hi_incl = incl; if (isf) p->dhi = LITF(l); else p->ihi = (int64_t)flit(&(l)); } while (0)
    switch (f->op) {
    case LLKV_OP_EQUALS: p->is_eq = 1; if (isf) p->dlo = LITF(f->value); else p->ilo = (int64_t)flit(&f->value); break;
    case LLKV_OP_GT: SETLO(f->value, 0); break;
    case LLKV_OP_GE: SETLO(f->value, 1); break;
    case LLKV_OP_LT: SETHI(f->value, 0); break;
    case LLKV_OP_LE: SETHI(f->value, 1); break;
    case LLKV_OP_RANGE:
      if (f->lower_kind) SETLO(f->lower, f->lower_kind == LLKV_BOUND_INCLUDED);
      if (f->upper_kind) SETHI(f->upper, f->upper_kind == LLKV_BOUND_INCLUDED);
      break;
    default: free(preds); free(fa); return LLKV_UNSUPPORTED;
    }
  }
  for (uint32_t a = 0; a < n_aggs; ++a) {
    fa[a].kind = aggs[a].kind; fa[a].expr = aggs[a].expr; fa[a].n = aggs[a].expr_len;
    for (uint32_t k = 0; k < aggs[a].expr_len; ++k) {
      const llkv_expr_token *e = &aggs[a].expr[k];
      if (e->kind == LLKV_TOK_COLUMN) { const orc_column *c = fcol(t, e->field_id); if (!c || c->validity) { free(preds); free(fa); return LLKV_UNSUPPORTED; } if (c->dtype == LLKV_DT_FLOAT64) fa[a].is_f64 = 1; }
      else if (e->kind == LLKV_TOK_LITERAL && e->literal.tag == LLKV_LIT_FLOAT64) fa[a].is_f64 = 1;
      else if (e->kind == LLKV_TOK_BINARY && (e->binop == LLKV_BIN_DIV || e->binop == LLKV_BIN_MOD)) { free(preds); free(fa); return LLKV_UNSUPPORTED; }
    }
  }
  fjob j;
  memset(&j, 0, sizeof j);
  j.t = t; j.preds = preds; j.n_preds = n_filters; j.aggs = fa; j.n_aggs = n_aggs;
  j.n_chunks = (t->rows + CHUNK_ROWS - 1) / CHUNK_ROWS;
  j.ng = 1;
  uint8_t values[2][256];
  uint32_t card[2] = {1, 1};
  if (n_keys > 2) { free(preds); free(fa); return LLKV_UNSUPPORTED; }
  for (uint32_t k = 0; k < n_keys; ++k) { /* one-character strings: Arrow offsets 0..n, the data bytes are the keys */
    const orc_column *c = fcol(t, key_fields[k]);
    if (!c || c->dtype != LLKV_DT_UTF8 || c->validity || (t->rows && (c->offsets[t->rows] != (int32_t)t->rows))) { free(preds); free(fa); return LLKV_UNSUPPORTED; }
    uint8_t seen[256] = {0};
    for (uint64_t r = 0; r < t->rows; ++r) seen[c->data[r]] = 1;
    card[k] = 0;
    for (int v = 0; v < 256; ++v) if (seen[v]) { j.code[k][v] = (uint8_t)card[k]; values[k][card[k]++] = (uint8_t)v; }
    if (card[k] == 0) card[k] = 1;
    j.key_data[k] = c->data;
  }
  j.n_keys = n_keys;
  j.card1 = card[1];
  j.ng = card[0] * card[1];
  j.partials = calloc((j.n_chunks ? j.n_chunks : 1) * j.ng * (n_aggs ? n_aggs : 1), sizeof(fstate));
  pthread_mutex_init(&j.mu, NULL);
  int nt = threads > 0 ? threads : 1;
  pthread_t *th = malloc(sizeof(pthread_t) * nt);
  for (int i = 0; i < nt; ++i) pthread_create(&th[i], NULL, worker, &j);
  for (int i = 0; i < nt; ++i) pthread_join(th[i], NULL);
  uint32_t n_out = 0;
  for (uint32_t g = 0; g < j.ng; ++g) {
  uint64_t group_rows = 0;
  for (uint64_t c = 0; c < j.n_chunks && n_aggs; ++c) {
    const fstate *p = &j.partials[(c * j.ng + g) * n_aggs];
    for (uint32_t a = 0; a < n_aggs; ++a) group_rows += (uint64_t)p[a].count;
  }
  if (n_keys && group_rows == 0) continue; /* a key combination without a selected row is no group */
  if (n_out == max_groups) break;
  if (out_keys) { out_keys[2 * n_out] = values[0][g / j.card1]; out_keys[2 * n_out + 1] = n_keys > 1 ? values[1][g % j.card1] : 0; }
  for (uint32_t a = 0; a < n_aggs; ++a) { /* combine in chunk order */
    fstate s;
    memset(&s, 0, sizeof s);
    for (uint64_t c = 0; c < j.n_chunks; ++c) {
      const fstate *p = &j.partials[(c * j.ng + g) * n_aggs + a];
      s.fsum += p->fsum; s.isum += p->isum; s.count += p->count;
      if (p->has) {
        if (!s.has) { s = (fstate){s.fsum, s.isum, s.count, p->fmin, p->fmax, p->imin, p->imax, 1}; }
        else { if (p->fmin < s.fmin) s.fmin = p->fmin; if (p->fmax > s.fmax) s.fmax = p->fmax; if (p->imin < s.imin) s.imin = p->imin; if (p->imax > s.imax) s.imax = p->imax; }
      }
    }
    llkv_value *o = &out_values[(size_t)n_out * n_aggs + a];
    memset(o, 0, sizeof *o);
    int isf = fa[a].is_f64;
    switch (fa[a].kind) {
    case LLKV_AGG_COUNT_STAR: case LLKV_AGG_COUNT: o->dtype = LLKV_DT_INT64; o->i64 = s.count; break;
    case LLKV_AGG_SUM: o->dtype = isf ? LLKV_DT_FLOAT64 : LLKV_DT_INT64; o->is_null = s.count == 0; o->f64 = s.fsum; o->i64 = (int64_t)s.isum; break;
    case LLKV_AGG_TOTAL: o->dtype = LLKV_DT_FLOAT64; o->f64 = s.fsum; break;
    case LLKV_AGG_AVG: o->dtype = LLKV_DT_FLOAT64; o->is_null = s.count == 0; if (s.count) o->f64 = (isf ? s.fsum : (double)(int64_t)s.isum) / (double)s.count; break;
    case LLKV_AGG_MIN: o->dtype = isf ? LLKV_DT_FLOAT64 : LLKV_DT_INT64; o->is_null = !s.has; o->f64 = s.fmin; o->i64 = s.imin; break;
    case LLKV_AGG_MAX: o->dtype = isf ? LLKV_DT_FLOAT64 : LLKV_DT_INT64; o->is_null = !s.has; o->f64 = s.fmax; o->i64 = s.imax; break;
    default: break;
    }
  }
  ++n_out;
  }
  if (out_groups) *out_groups = n_out;
  pthread_mutex_destroy(&j.mu);
  free(th); free(j.partials); free(preds); free(fa);
  return LLKV_OK;
}

int32_t orc_aggregate_parallel(const orc_table *t, const llkv_filter *filters, uint32_t n_filters,
                               const llkv_aggregate_spec *aggs, uint32_t n_aggs, llkv_value *out_values, int32_t threads) {
  return run_parallel(t, filters, n_filters, NULL, 0, aggs, n_aggs, out_values, NULL, NULL, 1, threads);
}

/* GROUP BY up to two one-character Utf8 columns; groups come out in key-byte order: out_keys[2g], out_keys[2g+1],
 * out_values[g][n_aggs]; *out_groups ≤ max_groups. */
int32_t orc_groupby_parallel(const orc_table *t, const llkv_filter *filters, uint32_t n_filters, const uint32_t *key_fields, uint32_t n_keys,
                             const llkv_aggregate_spec *aggs, uint32_t n_aggs, llkv_value *out_values, uint8_t *out_keys,
                             uint32_t *out_groups, uint32_t max_groups, int32_t threads) {
  if (n_keys == 0 || !out_keys || !out_groups) return LLKV_INVALID_ARGUMENT;
  return run_parallel(t, filters, n_filters, key_fields, n_keys, aggs, n_aggs, out_values, out_keys, out_groups, max_groups, threads);
}


/* STREAM-like triad a = b + s·c over `threads` threads (context for the CPU baseline: the host's memory bandwidth).
 * Returns GB/s over 3 × 8 bytes per element, best of `reps`. */
typedef struct triad_job { double *a; const double *b, *c; uint64_t lo, hi; } triad_job;
static void *triad_worker(void *arg) {
  triad_job *j = arg;
  for (uint64_t i = j->lo; i < j->hi; ++i) j->a[i] = j->b[i] + 3.0 * j->c[i];
  return NULL;
}
#include <time.h>
double orc_stream_triad(uint64_t n, int32_t threads, int32_t reps) {
  if (threads < 1) threads = 1;
  double *a = malloc(n * 8), *b = malloc(n * 8), *c = malloc(n * 8);
  if (!a || !b || !c) { free(a); free(b); free(c); return 0.0; }
  for (uint64_t i = 0; i < n; ++i) { a[i] = 0.0; b[i] = 1.0; c[i] = 2.0; }
  pthread_t *th = malloc(sizeof(pthread_t) * threads);
  triad_job *jobs = malloc(sizeof(triad_job) * threads);
  double best = 0.0;
  for (int r = 0; r < reps; ++r) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < threads; ++t) {
      jobs[t] = (triad_job){a, b, c, n * t / threads, n * (t + 1) / threads};
      pthread_create(&th[t], NULL, triad_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double sec = (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
    double gbs = 24.0 * (double)n / sec / 1e9;
    if (gbs > best) best = gbs;
  }
  double check = a[n / 2];
  free(a); free(b); free(c); free(th); free(jobs);
  return check == 7.0 ? best : 0.0;
}
