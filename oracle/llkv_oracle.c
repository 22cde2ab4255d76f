/*
 * llkv_oracle.c — CPU restatement of the reference's hot path (see llkv_oracle.h).
 * TEST INFRASTRUCTURE ONLY.  Sequential, same pass structure as the reference:
 * one pass per leaf predicate → row-id sets → set algebra → 65 536-row windows →
 * gather → one temporary per arithmetic operator → strictly sequential accumulate.
 *
 * Every function names the reference file:line it follows (paths relative to the
 * reference tree).  Third-party arithmetic restated from its published semantics:
 * arrow-rs 57.1.0 `numeric::{add,sub,mul,rem}` (checked for integers, IEEE-754 for
 * floats), Rust `partial_cmp`/`checked_add`/`as` casts, compiler-rt `powi`.
 */
#define _GNU_SOURCE
#include "llkv_oracle.h"

#include <errno.h>
#include <ctype.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef __int128 i128;
typedef unsigned __int128 u128;

#define ROW_STREAM_CHUNK_SIZE 65536u /* llkv-scan/src/execute.rs:31 */

/* ------------------------------------------------------------------ errors */
static __thread char g_err[512];

const char *orc_last_error(void) { return g_err; }

static int32_t fail(int32_t code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

void orc_free(void *p) { free(p); }

static void *xmalloc(size_t n) {
  void *p = malloc(n ? n : 1);
  if (!p) abort();
  return p;
}
static void *xcalloc(size_t n, size_t m) {
  void *p = calloc(n ? n : 1, m ? m : 1);
  if (!p) abort();
  return p;
}
static void *xrealloc(void *q, size_t n) {
  void *p = realloc(q, n ? n : 1);
  if (!p) abort();
  return p;
}

/* ------------------------------------------------------------- id vectors */
typedef struct idvec {
  uint64_t *v;
  uint64_t n, cap;
} idvec;

static void idv_push(idvec *a, uint64_t x) {
  if (a->n == a->cap) {
    a->cap = a->cap ? a->cap * 2 : 1024;
    a->v = xrealloc(a->v, a->cap * sizeof(uint64_t));
  }
  a->v[a->n++] = x;
}
static void idv_free(idvec *a) {
  free(a->v);
  a->v = NULL;
  a->n = a->cap = 0;
}
static idvec idv_all(uint64_t rows) {
  idvec r = {xmalloc(rows * sizeof(uint64_t)), rows, rows};
  for (uint64_t i = 0; i < rows; ++i) r.v[i] = i;
  return r;
}
/* Roaring `&`, `|`, `-` on sorted id lists (llkv-scan/src/predicate.rs:109-183). */
static idvec idv_and(const idvec *a, const idvec *b) {
  idvec r = {0};
  uint64_t i = 0, j = 0;
  while (i < a->n && j < b->n) {
    if (a->v[i] < b->v[j]) ++i;
    else if (a->v[i] > b->v[j]) ++j;
    else { idv_push(&r, a->v[i]); ++i; ++j; }
  }
  return r;
}
static idvec idv_or(const idvec *a, const idvec *b) {
  idvec r = {0};
  uint64_t i = 0, j = 0;
  while (i < a->n || j < b->n) {
    if (j >= b->n || (i < a->n && a->v[i] < b->v[j])) idv_push(&r, a->v[i++]);
    else if (i >= a->n || a->v[i] > b->v[j]) idv_push(&r, b->v[j++]);
    else { idv_push(&r, a->v[i]); ++i; ++j; }
  }
  return r;
}
static idvec idv_sub(const idvec *a, const idvec *b) {
  idvec r = {0};
  uint64_t i = 0, j = 0;
  while (i < a->n) {
    while (j < b->n && b->v[j] < a->v[i]) ++j;
    if (j >= b->n || b->v[j] != a->v[i]) idv_push(&r, a->v[i]);
    ++i;
  }
  return r;
}

/* ------------------------------------------------------------ column access */
static const orc_column *find_col(const orc_table *t, uint32_t field_id) {
  for (uint32_t i = 0; i < t->n_cols; ++i)
    if (t->cols[i].field_id == field_id) return &t->cols[i];
  return NULL;
}
static inline int col_valid(const orc_column *c, uint64_t row) {
  return !c->validity || ((c->validity[row >> 3] >> (row & 7)) & 1);
}
static const char *dtype_name(int32_t dt) {
  switch (dt) {
  case LLKV_DT_INT64: return "Int64";
  case LLKV_DT_FLOAT64: return "Float64";
  case LLKV_DT_INT32: return "Int32";
  case LLKV_DT_DATE32: return "Date32";
  case LLKV_DT_UINT64: return "UInt64";
  case LLKV_DT_UINT32: return "UInt32";
  case LLKV_DT_FLOAT32: return "Float32";
  case LLKV_DT_UTF8: return "Utf8";
  case LLKV_DT_BOOLEAN: return "Boolean";
  case LLKV_DT_DECIMAL128: return "Decimal128";
  default: return "Null";
  }
}

/* ----------------------------------------------------------- literal casts */
/* llkv-types/src/literal.rs:364-520 (`FromLiteral`). */
static i128 lit_i128(const llkv_literal *l) { return (i128)(((u128)(uint64_t)l->hi << 64) | (u128)l->lo); }

static const char *lit_kind(const llkv_literal *l) {
  switch (l->tag) {
  case LLKV_LIT_FLOAT64: return "float";
  case LLKV_LIT_BOOLEAN: return "boolean";
  case LLKV_LIT_STRING: return "string";
  case LLKV_LIT_DATE32: return "date";
  case LLKV_LIT_DECIMAL128: return "decimal";
  case LLKV_LIT_NULL: return "null";
  default: return "integer";
  }
}

/* compiler-rt __powidf2: what Rust's f64::powi lowers to. */
static double powi_f64(double a, int b) {
  const int recip = b < 0;
  double r = 1;
  for (;;) {
    if (b & 1) r *= a;
    b /= 2;
    if (b == 0) break;
    a *= a;
  }
  return recip ? 1 / r : r;
}

/* integer targets: literal.rs:368-420 */
static int32_t lit_to_int(const llkv_literal *l, i128 lo, i128 hi, const char *target, i128 *out) {
  i128 v;
  if (l->tag == LLKV_LIT_INT128) v = lit_i128(l);
  else if (l->tag == LLKV_LIT_DECIMAL128 && l->scale == 0) v = lit_i128(l);
  else
    return fail(LLKV_PREDICATE_BUILD, "literal cast error: expected integer, got %s", lit_kind(l));
  if (v < lo || v > hi)
    return fail(LLKV_PREDICATE_BUILD, "literal cast error: value out of range for %s", target);
  *out = v;
  return LLKV_OK;
}
/* f64: literal.rs:487-520; decimal → f64 llkv-types/src/decimal.rs:102-108 */
static int32_t lit_to_f64(const llkv_literal *l, double *out) {
  switch (l->tag) {
  case LLKV_LIT_FLOAT64: *out = l->f64; return LLKV_OK;
  case LLKV_LIT_INT128: *out = (double)lit_i128(l); return LLKV_OK;
  case LLKV_LIT_DECIMAL128: {
    i128 raw = lit_i128(l);
    *out = raw == 0 ? 0.0 : (double)raw / powi_f64(10.0, l->scale);
    return LLKV_OK;
  }
  default:
    return fail(LLKV_PREDICATE_BUILD, "literal cast error: expected float, got %s", lit_kind(l));
  }
}
/* f32: literal.rs:433-485 */
static int32_t lit_to_f32(const llkv_literal *l, float *out) {
  double v;
  int32_t rc = lit_to_f64(l, &v);
  if (rc) return rc;
  float c = (float)v;
  if (!isfinite(c)) return fail(LLKV_PREDICATE_BUILD, "literal cast error: float out of range for f32");
  *out = c;
  return LLKV_OK;
}

/* -------------------------------------------------------- typed predicates */
/* llkv-expr/src/typed_predicate.rs:41-147 (`Predicate<V>::matches`), :253-312. */
enum { VC_I64, VC_U64, VC_F64, VC_F32, VC_STR };
typedef union nat {
  int64_t i;
  uint64_t u;
  double d;
  float f;
  const char *s;
} nat;

typedef struct tpred {
  int kind; /* llkv_operator_kind, or 0 = Predicate::All */
  int vclass;
  nat a;
  int lower_kind, upper_kind;
  nat lo, hi;
  nat *in;
  uint32_t n_in;
  int case_sensitive; /* StartsWith / EndsWith / Contains */
} tpred;

static int vclass_of(int32_t dtype) {
  switch (dtype) {
  case LLKV_DT_INT64: case LLKV_DT_INT32: case LLKV_DT_DATE32: return VC_I64;
  case LLKV_DT_UINT64: case LLKV_DT_UINT32: return VC_U64;
  case LLKV_DT_FLOAT64: return VC_F64;
  case LLKV_DT_FLOAT32: return VC_F32;
  case LLKV_DT_UTF8: return VC_STR;
  default: return -1;
  }
}

static int32_t cast_native(const llkv_literal *l, int32_t dtype, nat *out) {
  i128 v;
  int32_t rc;
  switch (dtype) {
  case LLKV_DT_INT64:
    rc = lit_to_int(l, INT64_MIN, INT64_MAX, "i64", &v); if (rc) return rc; out->i = (int64_t)v; return LLKV_OK;
  case LLKV_DT_INT32: case LLKV_DT_DATE32: /* Date32 filters as i32: table.rs:1160-1167 */
    rc = lit_to_int(l, INT32_MIN, INT32_MAX, "i32", &v); if (rc) return rc; out->i = (int64_t)v; return LLKV_OK;
  case LLKV_DT_UINT64:
    rc = lit_to_int(l, 0, (i128)UINT64_MAX, "u64", &v); if (rc) return rc; out->u = (uint64_t)v; return LLKV_OK;
  case LLKV_DT_UINT32:
    rc = lit_to_int(l, 0, UINT32_MAX, "u32", &v); if (rc) return rc; out->u = (uint64_t)v; return LLKV_OK;
  case LLKV_DT_FLOAT64: return lit_to_f64(l, &out->d);
  case LLKV_DT_FLOAT32: return lit_to_f32(l, &out->f);
  case LLKV_DT_UTF8:
    if (l->tag != LLKV_LIT_STRING)
      return fail(LLKV_PREDICATE_BUILD, "literal cast error: expected string, got %s", lit_kind(l));
    out->s = l->str;
    return LLKV_OK;
  default:
    return fail(LLKV_INTERNAL, "Filtering on type %s is not supported", dtype_name(dtype));
  }
}

static int32_t build_predicate(const llkv_filter *f, int32_t dtype, tpred *p) {
  memset(p, 0, sizeof *p);
  p->vclass = vclass_of(dtype);
  if (p->vclass < 0) return fail(LLKV_INTERNAL, "Filtering on type %s is not supported", dtype_name(dtype));
  p->kind = f->op;
  int32_t rc;
  switch (f->op) {
  case LLKV_OP_EQUALS: case LLKV_OP_GT: case LLKV_OP_GE: case LLKV_OP_LT: case LLKV_OP_LE:
    return cast_native(&f->value, dtype, &p->a);
  case LLKV_OP_RANGE:
    p->lower_kind = f->lower_kind;
    p->upper_kind = f->upper_kind;
    if (f->lower_kind != LLKV_BOUND_UNBOUNDED && (rc = cast_native(&f->lower, dtype, &p->lo))) return rc;
    if (f->upper_kind != LLKV_BOUND_UNBOUNDED && (rc = cast_native(&f->upper, dtype, &p->hi))) return rc;
    if (f->lower_kind == LLKV_BOUND_UNBOUNDED && f->upper_kind == LLKV_BOUND_UNBOUNDED) p->kind = 0;
    return LLKV_OK;
  case LLKV_OP_IN:
    p->in = xmalloc(sizeof(nat) * (f->in_len ? f->in_len : 1));
    p->n_in = f->in_len;
    for (uint32_t i = 0; i < f->in_len; ++i)
      if ((rc = cast_native(&f->in_list[i], dtype, &p->in[i]))) { free(p->in); p->in = NULL; return rc; }
    return LLKV_OK;
  case LLKV_OP_STARTS_WITH: case LLKV_OP_ENDS_WITH: case LLKV_OP_CONTAINS: /* typed_predicate.rs:439-458: strings only */
    if (p->vclass != VC_STR) return fail(LLKV_PREDICATE_BUILD, "unsupported operator for typed predicate: operator lacks typed literal support");
    p->case_sensitive = f->case_sensitive;
    return cast_native(&f->value, dtype, &p->a);
  default:
    return fail(LLKV_PREDICATE_BUILD, "unsupported operator for typed predicate: operator lacks typed literal support");
  }
}

/* String patterns, typed_predicate.rs:186-210: case-insensitive = both sides through to_lowercase (restated for
 * ASCII; other bytes are left alone). */
static int str_pattern(int kind, const char *v, const char *pat, int case_sensitive) {
  size_t nv = strlen(v), np = strlen(pat);
  if (np > nv) return 0;
  char *a = xmalloc(nv + 1), *b = xmalloc(np + 1);
  for (size_t i = 0; i <= nv; ++i) a[i] = (!case_sensitive && v[i] >= 'A' && v[i] <= 'Z') ? (char)(v[i] + 32) : v[i];
  for (size_t i = 0; i <= np; ++i) b[i] = (!case_sensitive && pat[i] >= 'A' && pat[i] <= 'Z') ? (char)(pat[i] + 32) : pat[i];
  int r;
  if (kind == LLKV_OP_STARTS_WITH) r = memcmp(a, b, np) == 0;
  else if (kind == LLKV_OP_ENDS_WITH) r = memcmp(a + nv - np, b, np) == 0;
  else r = strstr(a, b) != NULL;
  free(a); free(b);
  return r;
}

/* Rust partial_cmp: -1 / 0 / 1, or 2 for "None" (NaN involved). */
static inline int pcmp(int vclass, nat v, nat t) {
  switch (vclass) {
  case VC_I64: return v.i < t.i ? -1 : v.i > t.i;
  case VC_U64: return v.u < t.u ? -1 : v.u > t.u;
  case VC_F64: return v.d < t.d ? -1 : v.d > t.d ? 1 : v.d == t.d ? 0 : 2;
  case VC_F32: return v.f < t.f ? -1 : v.f > t.f ? 1 : v.f == t.f ? 0 : 2;
  default: { int c = strcmp(v.s, t.s); return c < 0 ? -1 : c > 0; }
  }
}
static inline int peq(int vclass, nat v, nat t) {
  switch (vclass) {
  case VC_I64: return v.i == t.i;
  case VC_U64: return v.u == t.u;
  case VC_F64: return v.d == t.d;
  case VC_F32: return v.f == t.f;
  default: return strcmp(v.s, t.s) == 0;
  }
}
static int pred_matches(const tpred *p, nat v) {
  int c;
  switch (p->kind) {
  case 0: return 1;
  case LLKV_OP_EQUALS: return peq(p->vclass, v, p->a);
  case LLKV_OP_GT: return pcmp(p->vclass, v, p->a) == 1;
  case LLKV_OP_GE: c = pcmp(p->vclass, v, p->a); return c == 1 || c == 0;
  case LLKV_OP_LT: return pcmp(p->vclass, v, p->a) == -1;
  case LLKV_OP_LE: c = pcmp(p->vclass, v, p->a); return c == -1 || c == 0;
  case LLKV_OP_RANGE:
    if (p->lower_kind == LLKV_BOUND_INCLUDED) { c = pcmp(p->vclass, v, p->lo); if (!(c == 1 || c == 0)) return 0; }
    else if (p->lower_kind == LLKV_BOUND_EXCLUDED) { if (pcmp(p->vclass, v, p->lo) != 1) return 0; }
    if (p->upper_kind == LLKV_BOUND_INCLUDED) { c = pcmp(p->vclass, v, p->hi); if (!(c == -1 || c == 0)) return 0; }
    else if (p->upper_kind == LLKV_BOUND_EXCLUDED) { if (pcmp(p->vclass, v, p->hi) != -1) return 0; }
    return 1;
  case LLKV_OP_IN:
    for (uint32_t i = 0; i < p->n_in; ++i) if (peq(p->vclass, v, p->in[i])) return 1;
    return 0;
  case LLKV_OP_STARTS_WITH: case LLKV_OP_ENDS_WITH: case LLKV_OP_CONTAINS: return str_pattern(p->kind, v.s, p->a.s, p->case_sensitive);
  default: return 0;
  }
}

/* value of (col,row) in the predicate's value class; Utf8 values are copied to a
 * scratch buffer because Arrow strings are not NUL-terminated. */
static nat col_nat(const orc_column *c, uint64_t row, char **scratch, size_t *scratch_cap) {
  nat v;
  memset(&v, 0, sizeof v);
  switch (c->dtype) {
  case LLKV_DT_INT64: v.i = ((const int64_t *)c->values)[row]; break;
  case LLKV_DT_INT32: case LLKV_DT_DATE32: v.i = ((const int32_t *)c->values)[row]; break;
  case LLKV_DT_UINT64: v.u = ((const uint64_t *)c->values)[row]; break;
  case LLKV_DT_UINT32: v.u = ((const uint32_t *)c->values)[row]; break;
  case LLKV_DT_FLOAT64: v.d = ((const double *)c->values)[row]; break;
  case LLKV_DT_FLOAT32: v.f = ((const float *)c->values)[row]; break;
  case LLKV_DT_UTF8: {
    size_t len = (size_t)(c->offsets[row + 1] - c->offsets[row]);
    if (len + 1 > *scratch_cap) { *scratch_cap = len + 64; *scratch = xrealloc(*scratch, *scratch_cap); }
    memcpy(*scratch, c->data + c->offsets[row], len);
    (*scratch)[len] = 0;
    v.s = *scratch;
    break;
  }
  default: break;
  }
  return v;
}

/* Rows where `field` is present (llkv-table/src/table.rs:1202-1223). */
static int32_t field_nonnull_rows(const orc_table *t, uint32_t field_id, idvec *out) {
  const orc_column *c = find_col(t, field_id);
  if (!c) return fail(LLKV_NOT_FOUND, "field %u not found", field_id);
  idvec r = {0};
  for (uint64_t i = 0; i < t->rows; ++i) if (col_valid(c, i)) idv_push(&r, i);
  *out = r;
  return LLKV_OK;
}

/* MVCC row filter as a leaf: RowVersion::is_visible_for llkv-transaction/src/mvcc.rs:283-333 applied row by
 * row as filter_row_ids_for_snapshot does (llkv-transaction/src/helpers.rs:205-244).  TxnIdManager::status
 * (mvcc.rs:157-171): TXN_ID_NONE → None, TXN_ID_AUTO_COMMIT → Committed, listed ids → not committed,
 * anything else → Committed. */
static int32_t mvcc_visible_rows(const orc_table *t, const llkv_filter *f, idvec *out) {
  const orc_column *cc = find_col(t, f->field_id), *dc = find_col(t, (uint32_t)f->value.lo);
  if (!cc || !dc) return fail(LLKV_NOT_FOUND, "MVCC column not found");
  if (cc->dtype != LLKV_DT_UINT64 || dc->dtype != LLKV_DT_UINT64) return fail(LLKV_INVALID_ARGUMENT, "MVCC columns must be UInt64");
  const uint64_t txn = f->lower.lo, snap = f->upper.lo, NONE = UINT64_MAX, AUTO = 1;
  idvec r = {0};
  for (uint64_t i = 0; i < t->rows; ++i) {
    const uint64_t created = ((const uint64_t *)cc->values)[i], deleted = ((const uint64_t *)dc->values)[i];
    int visible;
#define COMMITTED(id, res) do { res = (id) != NONE; if ((id) != AUTO) for (uint32_t k = 0; k < f->in_len; ++k) if (f->in_list[k].lo == (id)) res = 0; } while (0)
    if (created == txn && txn != AUTO) visible = deleted != txn;
    else {
      int c_ok;
      COMMITTED(created, c_ok);
      if (!c_ok) visible = 0;
      else if (created > snap) visible = 0;
      else if (deleted == NONE) visible = 1;
      else if (deleted == txn && txn != AUTO) visible = 0;
      else {
        int d_ok;
        COMMITTED(deleted, d_ok);
        visible = !d_ok ? 1 : deleted > snap;
      }
    }
#undef COMMITTED
    if (visible) idv_push(&r, i);
  }
  *out = r;
  return LLKV_OK;
}

/* Leaf filter: llkv-table/src/table.rs:1117-1171,1225-1247; hot loop
 * llkv-column-map/src/store/scan/filter.rs:937-955 (sequential, push row id). */
static int32_t filter_leaf(const orc_table *t, const llkv_filter *f, idvec *out) {
  const orc_column *c = find_col(t, f->field_id);
  if (!c) return fail(LLKV_NOT_FOUND, "field %u not found", f->field_id);
  if (f->op == LLKV_OP_MVCC_VISIBLE) return mvcc_visible_rows(t, f, out);
  if (f->op == LLKV_OP_IS_NOT_NULL) return field_nonnull_rows(t, f->field_id, out);
  if (f->op == LLKV_OP_IS_NULL) {
    idvec all = idv_all(t->rows), nn;
    int32_t rc = field_nonnull_rows(t, f->field_id, &nn);
    if (rc) { idv_free(&all); return rc; }
    *out = idv_sub(&all, &nn);
    idv_free(&all); idv_free(&nn);
    return LLKV_OK;
  }
  if (f->op == LLKV_OP_RANGE && f->lower_kind == LLKV_BOUND_UNBOUNDED && f->upper_kind == LLKV_BOUND_UNBOUNDED) {
    *out = idv_all(t->rows); /* table.rs:1146-1153: every table row, NULLs included */
    return LLKV_OK;
  }
  if (c->dtype == LLKV_DT_DECIMAL128) /* llkv-table/src/table.rs:1160-1167 (the DataType's Debug form) */
    return fail(LLKV_INTERNAL, "Filtering on type Decimal128(%d, %d) is not supported", c->precision, c->scale);
  tpred p;
  int32_t rc = build_predicate(f, c->dtype, &p);
  if (rc) return rc;
  idvec r = {0};
  char *scratch = NULL;
  size_t cap = 0;
  for (uint64_t i = 0; i < t->rows; ++i) {
    if (!col_valid(c, i)) continue; /* None => false, table.rs:1241-1244 */
    if (pred_matches(&p, col_nat(c, i, &scratch, &cap))) idv_push(&r, i);
  }
  free(scratch);
  free(p.in);
  *out = r;
  return LLKV_OK;
}

static int32_t compare_rows(const orc_table *t, const llkv_filter *f, idvec *rows, idvec *dom);
static int32_t in_list_rows(const orc_table *t, const llkv_filter *f, idvec *rows, idvec *dom);
static int32_t is_null_expr_rows(const orc_table *t, const llkv_filter *f, idvec *rows, idvec *dom);

/* Predicate VM: llkv-scan/src/predicate.rs:32-193.  Each stack entry carries the
 * matching rows and the domain (rows where the sub-expression is determined),
 * which is what the reference's separate DomainProgram computes
 * (llkv-compute/src/program.rs:447-520; predicate.rs:665-777): Pred → non-null
 * rows of the field, And → intersect, Or → union, Literal → all rows. */
typedef struct vm_entry { idvec rows, dom; } vm_entry;

int32_t orc_filter_row_ids(const orc_table *t, const llkv_filter *filters, uint32_t n_filters,
                           const llkv_eval_op *ops, uint32_t n_ops, uint64_t **out_ids, uint64_t *out_len) {
  llkv_eval_op *synth = NULL;
  if (n_ops == 0) { /* Expr::all_of(filters) */
    if (n_filters == 0) {
      idvec all = idv_all(t->rows);
      *out_ids = all.v; *out_len = all.n;
      return LLKV_OK;
    }
    synth = xmalloc(sizeof(llkv_eval_op) * (n_filters + 1));
    for (uint32_t i = 0; i < n_filters; ++i) { synth[i].op = LLKV_EVAL_PUSH_PREDICATE; synth[i].arg = i; }
    n_ops = n_filters;
    if (n_filters > 1) { synth[n_ops].op = LLKV_EVAL_AND; synth[n_ops].arg = n_filters; ++n_ops; }
    ops = synth;
  }
  vm_entry *stack = xcalloc(n_ops + 1, sizeof(vm_entry));
  uint32_t sp = 0;
  int32_t rc = LLKV_OK;
  for (uint32_t k = 0; k < n_ops && rc == LLKV_OK; ++k) {
    const llkv_eval_op *op = &ops[k];
    switch (op->op) {
    case LLKV_EVAL_PUSH_PREDICATE: {
      if (op->arg >= n_filters) { rc = fail(LLKV_INTERNAL, "predicate index out of range"); break; }
      vm_entry e = {{0}, {0}};
      if (filters[op->arg].op == LLKV_OP_COMPARE) rc = compare_rows(t, &filters[op->arg], &e.rows, &e.dom);
      else if (filters[op->arg].op == LLKV_OP_IN_LIST) rc = in_list_rows(t, &filters[op->arg], &e.rows, &e.dom);
      else if (filters[op->arg].op == LLKV_OP_IS_NULL_EXPR) rc = is_null_expr_rows(t, &filters[op->arg], &e.rows, &e.dom);
      else {
        rc = filter_leaf(t, &filters[op->arg], &e.rows);
        if (rc == LLKV_OK) rc = field_nonnull_rows(t, filters[op->arg].field_id, &e.dom);
      }
      if (rc == LLKV_OK) stack[sp++] = e; else { idv_free(&e.rows); idv_free(&e.dom); }
      break;
    }
    case LLKV_EVAL_PUSH_LITERAL: {
      vm_entry e = {{0}, {0}};
      if (op->arg) e.rows = idv_all(t->rows);
      e.dom = idv_all(t->rows);
      stack[sp++] = e;
      break;
    }
    case LLKV_EVAL_AND: case LLKV_EVAL_OR: {
      if (op->arg == 0 || op->arg > sp) { rc = fail(LLKV_INTERNAL, "predicate stack underflow"); break; }
      vm_entry acc = stack[sp - op->arg];
      for (uint32_t c = 1; c < op->arg; ++c) {
        vm_entry nx = stack[sp - op->arg + c];
        idvec r = op->op == LLKV_EVAL_AND ? idv_and(&acc.rows, &nx.rows) : idv_or(&acc.rows, &nx.rows);
        idvec d = op->op == LLKV_EVAL_AND ? idv_and(&acc.dom, &nx.dom) : idv_or(&acc.dom, &nx.dom);
        idv_free(&acc.rows); idv_free(&acc.dom); idv_free(&nx.rows); idv_free(&nx.dom);
        acc.rows = r; acc.dom = d;
      }
      sp -= op->arg;
      stack[sp++] = acc;
      break;
    }
    case LLKV_EVAL_NOT: { /* predicate.rs:167-186: domain(child) − rows(child) */
      if (sp == 0) { rc = fail(LLKV_INTERNAL, "predicate stack underflow"); break; }
      vm_entry *e = &stack[sp - 1];
      idvec r = idv_sub(&e->dom, &e->rows);
      idv_free(&e->rows);
      e->rows = r;
      break;
    }
    default: rc = fail(LLKV_INTERNAL, "unknown predicate opcode %d", op->op);
    }
  }
  if (rc == LLKV_OK && sp != 1) rc = fail(LLKV_INTERNAL, "predicate program left %u entries", sp);
  if (rc == LLKV_OK) {
    *out_ids = stack[0].rows.v;
    *out_len = stack[0].rows.n;
    stack[0].rows.v = NULL;
  }
  for (uint32_t i = 0; i < sp; ++i) { idv_free(&stack[i].rows); idv_free(&stack[i].dom); }
  free(stack);
  free(synth);
  return rc;
}

/* ------------------------------------------------------ computed projections */
/* Arrays as they flow through the scan: i64 / f64 only after the cast step (other
 * integer widths are widened on gather: get_common_type keeps "widest",
 * llkv-compute/src/kernels.rs:179-242). */
typedef struct arr {
  int32_t dtype; /* LLKV_DT_INT64 or LLKV_DT_FLOAT64, or pass-through dtype for plain columns */
  uint64_t n;
  void *values;
  uint8_t *valid;
  char **strings;
  int32_t precision, scale; /* LLKV_DT_DECIMAL128 */
} arr;

static void arr_free(arr *a) {
  free(a->values);
  free(a->valid);
  if (a->strings) { for (uint64_t i = 0; i < a->n; ++i) free(a->strings[i]); free(a->strings); }
  memset(a, 0, sizeof *a);
}

static size_t dtype_width(int32_t dt) {
  switch (dt) {
  case LLKV_DT_INT64: case LLKV_DT_UINT64: case LLKV_DT_FLOAT64: return 8;
  case LLKV_DT_INT32: case LLKV_DT_DATE32: case LLKV_DT_UINT32: case LLKV_DT_FLOAT32: return 4;
  case LLKV_DT_BOOLEAN: return 1;
  case LLKV_DT_DECIMAL128: return 16;
  default: return 0;
  }
}

/* Gather one column for a window of row ids: llkv-column-map/src/store/projection.rs
 * :929-1352, gather.rs:764-884 — out[i] = value_of(row_ids[i]), absent → NULL. */
static arr gather_column(const orc_column *c, const uint64_t *ids, uint64_t n) {
  arr a;
  memset(&a, 0, sizeof a);
  a.dtype = c->dtype;
  a.precision = c->precision; a.scale = c->scale;
  a.n = n;
  a.valid = xmalloc(n);
  if (c->dtype == LLKV_DT_UTF8) {
    a.strings = xcalloc(n, sizeof(char *));
    for (uint64_t i = 0; i < n; ++i) {
      a.valid[i] = (uint8_t)col_valid(c, ids[i]);
      size_t len = a.valid[i] ? (size_t)(c->offsets[ids[i] + 1] - c->offsets[ids[i]]) : 0;
      a.strings[i] = xmalloc(len + 1);
      if (len) memcpy(a.strings[i], c->data + c->offsets[ids[i]], len);
      a.strings[i][len] = 0;
    }
    return a;
  }
  size_t w = dtype_width(c->dtype);
  a.values = xmalloc(n * w);
  for (uint64_t i = 0; i < n; ++i) {
    a.valid[i] = (uint8_t)col_valid(c, ids[i]);
    memcpy((char *)a.values + i * w, (const char *)c->values + ids[i] * w, w);
  }
  return a;
}

static int is_int_dtype(int32_t dt) {
  return dt == LLKV_DT_INT64 || dt == LLKV_DT_INT32 || dt == LLKV_DT_UINT32;
}
/* get_common_type restricted to the numeric types on this path
 * (llkv-compute/src/kernels.rs:179-242): any float → Float64; signed ints → widest;
 * Int32/UInt32 with Int64 → Int64; a 64-bit unsigned side with a signed side → Float64. */
static int32_t common_type(int32_t a, int32_t b) {
  if (a == b) return a;
  if (a == LLKV_DT_FLOAT64 || b == LLKV_DT_FLOAT64 || a == LLKV_DT_FLOAT32 || b == LLKV_DT_FLOAT32) return LLKV_DT_FLOAT64;
  if (a == LLKV_DT_UINT64 || b == LLKV_DT_UINT64) return LLKV_DT_FLOAT64;
  if (is_int_dtype(a) && is_int_dtype(b)) return LLKV_DT_INT64;
  return LLKV_DT_NULL;
}

/* ScalarEvaluator::simplify llkv-compute/src/eval.rs:761-791, applied by the scan to every computed projection first
 * (llkv-scan/src/execute.rs:91): Binary(Literal, Literal) → fold_binary_literals (:1010-1031) = compute_binary
 * (kernels.rs:99-177) over literal_to_array of each side (:521-543: Int128 → Int64 `as i64`, Float64 → Float64, Null →
 * Null array): common type, checked integer kernels, Divide nullifies a zero divisor first (cmp::eq against the cast 0 —
 * totalOrder on floats, so −0.0 is not "zero"), float kernels IEEE; a NULL result → Literal::Null; a kernel error →
 * None, the node stays unfolded (returned as UNSUPPORTED here: neither side evaluates such plans).  Postfix in,
 * postfix out (`out` holds ≥ n tokens). */
static int is_numeric_literal_token(const llkv_expr_token *t) {
  return t->kind == LLKV_TOK_LITERAL && (t->literal.tag == LLKV_LIT_INT128 || t->literal.tag == LLKV_LIT_FLOAT64 || t->literal.tag == LLKV_LIT_NULL);
}
static int32_t simplify_tokens(const llkv_expr_token *e, uint32_t n, llkv_expr_token *out, uint32_t *n_out) {
  uint32_t m = 0;
  for (uint32_t i = 0; i < n; ++i) {
    if (!(e[i].kind == LLKV_TOK_BINARY && m >= 2 && is_numeric_literal_token(&out[m - 1]) && is_numeric_literal_token(&out[m - 2]))) { out[m++] = e[i]; continue; }
    const llkv_literal *l = &out[m - 2].literal, *r = &out[m - 1].literal;
    llkv_literal z;
    memset(&z, 0, sizeof z);
    if (e[i].binop < LLKV_BIN_ADD || e[i].binop > LLKV_BIN_MOD) return fail(LLKV_UNSUPPORTED, "constant sub-expression under this operator");
    if (l->tag == LLKV_LIT_NULL || r->tag == LLKV_LIT_NULL) z.tag = LLKV_LIT_NULL; /* coerced to the other side's type, NULL in → NULL out */
    else if (l->tag == LLKV_LIT_FLOAT64 || r->tag == LLKV_LIT_FLOAT64) { /* get_common_type: Float64 */
      double a = l->tag == LLKV_LIT_FLOAT64 ? l->f64 : (double)(int64_t)lit_i128(l);
      double b = r->tag == LLKV_LIT_FLOAT64 ? r->f64 : (double)(int64_t)lit_i128(r);
      z.tag = LLKV_LIT_FLOAT64;
      switch (e[i].binop) {
      case LLKV_BIN_ADD: z.f64 = a + b; break;
      case LLKV_BIN_SUB: z.f64 = a - b; break;
      case LLKV_BIN_MUL: z.f64 = a * b; break;
      case LLKV_BIN_DIV: if (b == 0.0 && !signbit(b)) z.tag = LLKV_LIT_NULL; else z.f64 = a / b; break;
      case LLKV_BIN_MOD: z.f64 = fmod(a, b); break;
      }
    } else {
      int64_t a = (int64_t)lit_i128(l), b = (int64_t)lit_i128(r), v = 0;
      int err = 0, null = 0;
      switch (e[i].binop) {
      case LLKV_BIN_ADD: err = __builtin_add_overflow(a, b, &v); break;
      case LLKV_BIN_SUB: err = __builtin_sub_overflow(a, b, &v); break;
      case LLKV_BIN_MUL: err = __builtin_mul_overflow(a, b, &v); break;
      case LLKV_BIN_DIV: if (b == 0) null = 1; else if (a == INT64_MIN && b == -1) err = 1; else v = a / b; break;
      case LLKV_BIN_MOD: if (b == 0) err = 1; else v = b == -1 ? 0 : a % b; break;
      }
      if (err) return fail(LLKV_UNSUPPORTED, "constant sub-expression the reference leaves unfolded (its fold errors)");
      if (null) z.tag = LLKV_LIT_NULL;
      else { z.tag = LLKV_LIT_INT128; z.lo = (uint64_t)v; z.hi = v < 0 ? -1 : 0; }
    }
    --m;
    out[m - 1].literal = z;
  }
  *n_out = m;
  return LLKV_OK;
}

static int32_t infer_simplified_type(const orc_table *t, const llkv_expr_token *e, uint32_t n, int *has_div, int32_t *rc_out);
/* Result type of a postfix token program (fast_numeric.rs:249-298, PlanBuilder::visit), after simplify.
 * Returns LLKV_DT_NULL when the program is not numeric. */
static int32_t infer_expr_type(const orc_table *t, const llkv_expr_token *e, uint32_t n, int *has_div, int32_t *rc_out) {
  llkv_expr_token *s = xmalloc((n ? n : 1) * sizeof *s);
  uint32_t m = 0;
  *has_div = 0;
  if ((*rc_out = simplify_tokens(e, n, s, &m))) { free(s); return LLKV_DT_NULL; }
  int32_t dt = infer_simplified_type(t, s, m, has_div, rc_out);
  free(s);
  return dt;
}
static int32_t infer_simplified_type(const orc_table *t, const llkv_expr_token *e, uint32_t n, int *has_div, int32_t *rc_out) {
  int32_t st[64];
  uint32_t sp = 0;
  *has_div = 0;
  *rc_out = LLKV_OK;
  for (uint32_t i = 0; i < n; ++i) {
    if (sp >= 63) { *rc_out = fail(LLKV_INTERNAL, "expression too deep"); return LLKV_DT_NULL; }
    switch (e[i].kind) {
    case LLKV_TOK_COLUMN: {
      const orc_column *c = find_col(t, e[i].field_id);
      if (!c) { *rc_out = fail(LLKV_NOT_FOUND, "field %u not found", e[i].field_id); return LLKV_DT_NULL; }
      st[sp++] = c->dtype;
      break;
    }
    case LLKV_TOK_LITERAL:
      /* fast_numeric.rs:300-309: Int128 / Decimal(raw) → Int, Float64 → Float */
      if (e[i].literal.tag == LLKV_LIT_FLOAT64) st[sp++] = LLKV_DT_FLOAT64;
      else if (e[i].literal.tag == LLKV_LIT_INT128 || e[i].literal.tag == LLKV_LIT_DECIMAL128 || e[i].literal.tag == LLKV_LIT_NULL) st[sp++] = LLKV_DT_INT64;
      else { *rc_out = fail(LLKV_UNSUPPORTED, "non-numeric literal in computed projection"); return LLKV_DT_NULL; }
      break;
    case LLKV_TOK_BINARY: {
      if (sp < 2) { *rc_out = fail(LLKV_INTERNAL, "fast path stack underflow"); return LLKV_DT_NULL; }
      if (e[i].binop == LLKV_BIN_DIV) *has_div = 1;
      int32_t r = common_type(st[sp - 2], st[sp - 1]);
      if (r == LLKV_DT_NULL || r == LLKV_DT_DATE32 || r == LLKV_DT_UTF8) { *rc_out = fail(LLKV_UNSUPPORTED, "unsupported operand types %s, %s", dtype_name(st[sp - 2]), dtype_name(st[sp - 1])); return LLKV_DT_NULL; }
      /* Int32 ⊕ Int32 (UInt32 ⊕ UInt32) stays 32 bits wide — what the ROOT type is decides the fast path's kernels (eval_simplified).
       * (Numeric literal pairs were folded by simplify_tokens; what is left is not numeric.) */
      if (i >= 2 && e[i - 1].kind == LLKV_TOK_LITERAL && e[i - 2].kind == LLKV_TOK_LITERAL) { *rc_out = fail(LLKV_UNSUPPORTED, "constant sub-expression"); return LLKV_DT_NULL; }
      sp -= 2;
      st[sp++] = r;
      break;
    }
    default: *rc_out = fail(LLKV_INTERNAL, "bad token"); return LLKV_DT_NULL;
    }
  }
  if (sp != 1) { *rc_out = fail(LLKV_INTERNAL, "fast path evaluation missing result"); return LLKV_DT_NULL; }
  return st[0];
}

/* arrow `cast` of a numeric column to the program's target type (fast_numeric.rs:80-87). */
static arr cast_to(const arr *src, int32_t target) {
  arr a;
  memset(&a, 0, sizeof a);
  a.dtype = target;
  a.n = src->n;
  a.valid = xmalloc(src->n);
  memcpy(a.valid, src->valid, src->n);
  a.values = xmalloc(src->n * 8);
  for (uint64_t i = 0; i < src->n; ++i) {
    double d = 0;
    int64_t v = 0;
    switch (src->dtype) {
    case LLKV_DT_INT64: v = ((int64_t *)src->values)[i]; d = (double)v; break;
    case LLKV_DT_INT32: v = ((int32_t *)src->values)[i]; d = (double)v; break;
    case LLKV_DT_UINT32: v = ((uint32_t *)src->values)[i]; d = (double)v; break;
    case LLKV_DT_UINT64: d = (double)((uint64_t *)src->values)[i]; v = (int64_t)((uint64_t *)src->values)[i]; break;
    case LLKV_DT_FLOAT64: d = ((double *)src->values)[i]; v = (int64_t)d; break;
    case LLKV_DT_FLOAT32: d = ((float *)src->values)[i]; v = (int64_t)d; break;
    default: break;
    }
    if (target == LLKV_DT_FLOAT64) ((double *)a.values)[i] = d; else ((int64_t *)a.values)[i] = v;
  }
  return a;
}

/* One arrow-arith kernel call = one temporary array (fast_numeric.rs:312-356, kernels.rs:99-177):
 * integers checked (overflow → error; `%` is mod_wrapping after a zero check → "Divide by zero"), floats IEEE (fmod for %); NULL in → NULL out,
 * and a slot is only evaluated where both operands are valid.  Divide is never on the fast path (:273-275);
 * on the generic path zeros of the divisor become NULLs first (kernels.rs:121-135), then arrow `div`
 * (truncating, i64::MIN / -1 overflows). */
/* Fast path whose target type is Int32 / UInt32 (fast_numeric.rs:333-355: binary_prim!(Int32Type) — arrow's checked 32-bit kernels).
 * The temporaries here are 64 bits wide; the operands are within 32 bits, so + − * are exact in 64 and the reference's overflow is
 * "the result does not fit" (1: Int32, 2: UInt32; 0: the 64-bit kernels).  Set around the fast path's loop by eval_simplified. */
static int g_fit32 = 0;
static int32_t binary_kernel(const arr *l, const arr *r, int32_t op, arr *out) {
  arr a;
  memset(&a, 0, sizeof a);
  a.dtype = l->dtype;
  a.n = l->n;
  a.valid = xmalloc(l->n ? l->n : 1);
  a.values = xmalloc((l->n ? l->n : 1) * 8);
  for (uint64_t i = 0; i < l->n; ++i) {
    a.valid[i] = l->valid[i] && r->valid[i];
    ((int64_t *)a.values)[i] = 0;
    if (!a.valid[i]) continue;
    if (l->dtype == LLKV_DT_FLOAT64) {
      double x = ((double *)l->values)[i], y = ((double *)r->values)[i], z = 0;
      switch (op) {
      case LLKV_BIN_ADD: z = x + y; break;
      case LLKV_BIN_SUB: z = x - y; break;
      case LLKV_BIN_MUL: z = x * y; break;
      case LLKV_BIN_DIV: if (y == 0.0 && !signbit(y)) a.valid[i] = 0; else z = x / y; break; /* nullif(eq(rhs, cast(0))): arrow-ord `eq` is totalOrder on floats, −0.0 is not 0.0 */
      case LLKV_BIN_MOD: z = fmod(x, y); break;
      }
      ((double *)a.values)[i] = z;
    } else {
      int64_t x = ((int64_t *)l->values)[i], y = ((int64_t *)r->values)[i], z = 0;
      int ovf = 0;
      switch (op) {
      case LLKV_BIN_ADD: ovf = __builtin_add_overflow(x, y, &z); break;
      case LLKV_BIN_SUB: ovf = __builtin_sub_overflow(x, y, &z); break;
      case LLKV_BIN_MUL: ovf = __builtin_mul_overflow(x, y, &z); break;
      case LLKV_BIN_DIV: if (y == 0) a.valid[i] = 0; else if (x == INT64_MIN && y == -1) ovf = 1; else z = x / y; break;
      case LLKV_BIN_MOD:
        if (y == 0) { arr_free(&a); return fail(LLKV_INTERNAL, "Divide by zero"); }
        z = y == -1 ? 0 : x % y; /* mod_wrapping: i64::MIN % -1 = 0 */
        break;
      }
      if (g_fit32 == 1 && (z < INT32_MIN || z > INT32_MAX)) ovf = 1;
      if (g_fit32 == 2 && (z < 0 || z > (int64_t)UINT32_MAX)) ovf = 1;
      if (ovf) { arr_free(&a); return fail(LLKV_INTERNAL, "Arithmetic overflow: Overflow happened on: %lld %s %lld", (long long)x, op == LLKV_BIN_ADD ? "+" : op == LLKV_BIN_SUB ? "-" : op == LLKV_BIN_MUL ? "*" : op == LLKV_BIN_DIV ? "/" : "%", (long long)y); }
      ((int64_t *)a.values)[i] = z;
    }
  }
  *out = a;
  return LLKV_OK;
}

/* Evaluate a computed projection over gathered columns:
 * llkv-compute/src/eval.rs:555-614 → NumericFastPath::execute fast_numeric.rs:69-121.
 * Every column is cast to the final type first, every literal broadcast in the
 * final type, then one kernel (one temporary) per operator. */
typedef struct gathered {
  uint32_t field_id;
  arr a;
} gathered;

static const arr *find_gathered(const gathered *g, uint32_t n, uint32_t field_id) {
  for (uint32_t i = 0; i < n; ++i) if (g[i].field_id == field_id) return &g[i].a;
  return NULL;
}

static int32_t eval_simplified(const orc_table *t, const llkv_expr_token *e, uint32_t n_tok,
                               const gathered *g, uint32_t n_g, uint64_t len, arr *out);
static int32_t eval_program(const orc_table *t, const llkv_expr_token *e, uint32_t n_tok,
                            const gathered *g, uint32_t n_g, uint64_t len, arr *out) {
  llkv_expr_token *s = xmalloc((n_tok ? n_tok : 1) * sizeof *s);
  uint32_t m = 0;
  int32_t rc = simplify_tokens(e, n_tok, s, &m);
  if (rc == LLKV_OK) rc = eval_simplified(t, s, m, g, n_g, len, out);
  free(s);
  return rc;
}
static int32_t eval_simplified(const orc_table *t, const llkv_expr_token *e, uint32_t n_tok,
                               const gathered *g, uint32_t n_g, uint64_t len, arr *out) {
  int has_div, rc;
  int32_t target = infer_simplified_type(t, e, n_tok, &has_div, &rc);
  if (rc) return rc;
  const int32_t root = target;
  if (target != LLKV_DT_FLOAT64) target = target == LLKV_DT_NULL ? LLKV_DT_NULL : LLKV_DT_INT64;
  if (target == LLKV_DT_NULL) return fail(LLKV_UNSUPPORTED, "non-numeric computed projection");
  /* a root type of 32 bits (every leaf a column of that type) on the fast path: 32-bit checked kernels, a 32-bit result array */
  const int fit32 = (!has_div && n_tok > 1 && root == LLKV_DT_INT32) ? 1 : (!has_div && n_tok > 1 && root == LLKV_DT_UINT32) ? 2 : 0;
  arr st[64];
  uint32_t sp = 0;
  rc = LLKV_OK;
  if (has_div) {
    /* Generic route (try_evaluate_vectorized eval.rs:616-665 → compute_binary kernels.rs:99-177): every Binary
     * node coerces its own operands to their common type.  Restated for Int64 / Float64 operands. */
    for (uint32_t i = 0; i < n_tok && rc == LLKV_OK; ++i) {
      switch (e[i].kind) {
      case LLKV_TOK_COLUMN: {
        const arr *src = find_gathered(g, n_g, e[i].field_id);
        if (!src) { rc = fail(LLKV_INTERNAL, "missing column for field"); break; }
        if (src->dtype != LLKV_DT_INT64 && src->dtype != LLKV_DT_FLOAT64) { rc = fail(LLKV_UNSUPPORTED, "division over %s operands", dtype_name(src->dtype)); break; }
        st[sp++] = cast_to(src, src->dtype);
        break;
      }
      case LLKV_TOK_LITERAL: {
        const llkv_literal *l = &e[i].literal;
        if (l->tag != LLKV_LIT_INT128 && l->tag != LLKV_LIT_FLOAT64) { rc = fail(LLKV_UNSUPPORTED, "literal kind in a computed projection"); break; }
        arr a;
        memset(&a, 0, sizeof a);
        a.dtype = l->tag == LLKV_LIT_FLOAT64 ? LLKV_DT_FLOAT64 : LLKV_DT_INT64; a.n = len; a.valid = xmalloc(len ? len : 1); a.values = xmalloc((len ? len : 1) * 8);
        for (uint64_t k = 0; k < len; ++k) {
          a.valid[k] = 1;
          if (a.dtype == LLKV_DT_FLOAT64) ((double *)a.values)[k] = l->f64; else ((int64_t *)a.values)[k] = (int64_t)lit_i128(l); /* `*i as i64` */
        }
        st[sp++] = a;
        break;
      }
      case LLKV_TOK_BINARY: {
        arr r = st[--sp], l = st[--sp], z;
        if (l.dtype != r.dtype) { /* Int64 ⊕ Float64 → Float64 */
          arr *ia = l.dtype == LLKV_DT_INT64 ? &l : &r;
          arr c = cast_to(ia, LLKV_DT_FLOAT64);
          arr_free(ia);
          *ia = c;
        }
        rc = binary_kernel(&l, &r, e[i].binop, &z);
        arr_free(&l); arr_free(&r);
        if (rc == LLKV_OK) st[sp++] = z;
        break;
      }
      }
    }
    if (rc != LLKV_OK) { for (uint32_t i = 0; i < sp; ++i) arr_free(&st[i]); return rc; }
    *out = st[0];
    return LLKV_OK;
  }
  g_fit32 = fit32;
  for (uint32_t i = 0; i < n_tok && rc == LLKV_OK; ++i) {
    switch (e[i].kind) {
    case LLKV_TOK_COLUMN: {
      const arr *src = find_gathered(g, n_g, e[i].field_id);
      if (!src) { rc = fail(LLKV_INTERNAL, "missing numeric array for fast path"); break; }
      st[sp++] = cast_to(src, target);
      break;
    }
    case LLKV_TOK_LITERAL: { /* make_literal_array fast_numeric.rs:123-240 */
      arr a;
      memset(&a, 0, sizeof a);
      a.dtype = target; a.n = len; a.valid = xmalloc(len); a.values = xmalloc(len * 8);
      const llkv_literal *l = &e[i].literal;
      int isnull = l->tag == LLKV_LIT_NULL;
      for (uint64_t k = 0; k < len; ++k) {
        a.valid[k] = !isnull;
        if (target == LLKV_DT_FLOAT64) ((double *)a.values)[k] = isnull ? 0 : l->tag == LLKV_LIT_FLOAT64 ? l->f64 : (double)lit_i128(l);
        else ((int64_t *)a.values)[k] = isnull ? 0 : l->tag == LLKV_LIT_FLOAT64 ? (int64_t)l->f64 : (int64_t)lit_i128(l);
      }
      st[sp++] = a;
      break;
    }
    case LLKV_TOK_BINARY: {
      arr r = st[--sp], l = st[--sp], z;
      rc = binary_kernel(&l, &r, e[i].binop, &z);
      arr_free(&l); arr_free(&r);
      if (rc == LLKV_OK) st[sp++] = z;
      break;
    }
    }
  }
  g_fit32 = 0;
  if (rc != LLKV_OK) { for (uint32_t i = 0; i < sp; ++i) arr_free(&st[i]); return rc; }
  if (fit32) { /* the result array in its own width */
    arr z = st[0];
    int32_t *narrow = xmalloc((z.n ? z.n : 1) * 4);
    for (uint64_t k = 0; k < z.n; ++k) narrow[k] = (int32_t)((int64_t *)z.values)[k]; /* (UInt32: the same low 32 bits) */
    free(z.values);
    z.values = narrow;
    z.dtype = root;
    st[0] = z;
  }
  *out = st[0];
  return LLKV_OK;
}

/* ------------------------------------------------------------ Expr::Compare */
/* collect_row_ids_for_compare llkv-scan/src/predicate.rs:333-396 and its domain twin :779-818:
 *  - column ⋈ non-NULL literal (not <>) is rewritten to a leaf filter (simple_compare_filter :970-1010);
 *  - otherwise both sides are evaluated over the rows where every referenced field is present
 *    (evaluate_compare_rows :562-663), coerced to get_common_type of their result types and compared with
 *    arrow-ord's cmp kernels (compute_compare llkv-compute/src/kernels.rs:269-297) — integers natively,
 *    floats by IEEE totalOrder; a row is "determined" when both sides are non-NULL. */
static int64_t total_order_key(double v) {
  int64_t b;
  memcpy(&b, &v, 8);
  return b ^ (int64_t)((uint64_t)(b >> 63) >> 1);
}

/* One side over `n` row ids; the array keeps the side's own arrow type. */
static int32_t compare_side(const orc_table *t, const llkv_expr_token *e, uint32_t n_tok, const gathered *g, uint32_t n_g,
                            const uint64_t *ids, uint64_t n, arr *out) {
  for (uint32_t i = 0; i < n_tok; ++i)
    if (e[i].kind == LLKV_TOK_LITERAL && e[i].literal.tag == LLKV_LIT_NULL) return fail(LLKV_UNSUPPORTED, "NULL literal in a comparison");
  if (n_tok == 1 && e[0].kind == LLKV_TOK_COLUMN) { /* VectorizedExpr::Array: the column as gathered */
    const orc_column *c = find_col(t, e[0].field_id);
    if (!c) return fail(LLKV_NOT_FOUND, "field %u not found", e[0].field_id);
    if (dtype_width(c->dtype) == 0 || c->dtype == LLKV_DT_BOOLEAN || c->dtype == LLKV_DT_DATE32) return fail(LLKV_UNSUPPORTED, "comparison over %s", dtype_name(c->dtype));
    *out = gather_column(c, ids, n);
    return LLKV_OK;
  }
  if (n_tok == 1 && e[0].kind == LLKV_TOK_LITERAL) {
    const llkv_literal *l = &e[0].literal;
    arr a;
    memset(&a, 0, sizeof a);
    a.n = n; a.valid = xmalloc(n ? n : 1); a.values = xmalloc((n ? n : 1) * 8);
    memset(a.valid, 1, n);
    if (l->tag == LLKV_LIT_FLOAT64) { a.dtype = LLKV_DT_FLOAT64; for (uint64_t k = 0; k < n; ++k) ((double *)a.values)[k] = l->f64; }
    else if (l->tag == LLKV_LIT_INT128) {
      __int128 v = lit_i128(l);
      if (v < (__int128)INT64_MIN || v > (__int128)INT64_MAX) { arr_free(&a); return fail(LLKV_UNSUPPORTED, "integer literal beyond Int64 in a comparison"); }
      a.dtype = LLKV_DT_INT64;
      for (uint64_t k = 0; k < n; ++k) ((int64_t *)a.values)[k] = (int64_t)v;
    } else { arr_free(&a); return fail(LLKV_UNSUPPORTED, "non-numeric literal in a comparison"); }
    *out = a;
    return LLKV_OK;
  }
  return eval_program(t, e, n_tok, g, n_g, n, out);
}

static int is_unsigned_dtype(int32_t dt) { return dt == LLKV_DT_UINT32 || dt == LLKV_DT_UINT64; }
static int is_64bit_dtype(int32_t dt) { return dt == LLKV_DT_INT64 || dt == LLKV_DT_UINT64; }

static int rel_i(int32_t op, int64_t a, int64_t b) {
  switch (op) { case LLKV_CMP_EQ: return a == b; case LLKV_CMP_NOT_EQ: return a != b; case LLKV_CMP_LT: return a < b;
                case LLKV_CMP_LT_EQ: return a <= b; case LLKV_CMP_GT: return a > b; default: return a >= b; }
}
static int rel_u(int32_t op, uint64_t a, uint64_t b) {
  switch (op) { case LLKV_CMP_EQ: return a == b; case LLKV_CMP_NOT_EQ: return a != b; case LLKV_CMP_LT: return a < b;
                case LLKV_CMP_LT_EQ: return a <= b; case LLKV_CMP_GT: return a > b; default: return a >= b; }
}
static double side_as_f64(const arr *a, uint64_t i) { /* arrow cast → Float64 */
  switch (a->dtype) {
  case LLKV_DT_FLOAT64: return ((double *)a->values)[i];
  case LLKV_DT_FLOAT32: return (double)((float *)a->values)[i];
  case LLKV_DT_INT64: return (double)((int64_t *)a->values)[i];
  case LLKV_DT_UINT64: return (double)((uint64_t *)a->values)[i];
  case LLKV_DT_INT32: return (double)((int32_t *)a->values)[i];
  default: return (double)((uint32_t *)a->values)[i];
  }
}
static int64_t side_as_i64(const arr *a, uint64_t i) {
  switch (a->dtype) {
  case LLKV_DT_INT64: return ((int64_t *)a->values)[i];
  case LLKV_DT_UINT64: return (int64_t)((uint64_t *)a->values)[i];
  case LLKV_DT_INT32: return ((int32_t *)a->values)[i];
  default: return (int64_t)((uint32_t *)a->values)[i];
  }
}

static int32_t compare_rows(const orc_table *t, const llkv_filter *f, idvec *rows, idvec *dom) {
  if (f->cmp_op < LLKV_CMP_EQ || f->cmp_op > LLKV_CMP_GT_EQ) return fail(LLKV_INVALID_ARGUMENT, "unknown compare operator");
  if (!f->cmp_left || !f->cmp_right || !f->cmp_left_len || !f->cmp_right_len) return fail(LLKV_INVALID_ARGUMENT, "compare needs two expressions");
  const llkv_expr_token *l = f->cmp_left, *r = f->cmp_right;
  const int l_col = f->cmp_left_len == 1 && l[0].kind == LLKV_TOK_COLUMN, r_col = f->cmp_right_len == 1 && r[0].kind == LLKV_TOK_COLUMN;
  const int l_lit = f->cmp_left_len == 1 && l[0].kind == LLKV_TOK_LITERAL, r_lit = f->cmp_right_len == 1 && r[0].kind == LLKV_TOK_LITERAL;
  if (f->cmp_op != LLKV_CMP_NOT_EQ && ((l_col && r_lit) || (l_lit && r_col))) {
    const llkv_literal *lit = l_col ? &r[0].literal : &l[0].literal;
    if (lit->tag != LLKV_LIT_NULL) {
      llkv_filter leaf;
      memset(&leaf, 0, sizeof leaf);
      leaf.field_id = l_col ? l[0].field_id : r[0].field_id;
      int32_t op = f->cmp_op;
      if (!l_col) op = op == LLKV_CMP_LT ? LLKV_CMP_GT : op == LLKV_CMP_LT_EQ ? LLKV_CMP_GT_EQ : op == LLKV_CMP_GT ? LLKV_CMP_LT : op == LLKV_CMP_GT_EQ ? LLKV_CMP_LT_EQ : op;
      leaf.op = op == LLKV_CMP_EQ ? LLKV_OP_EQUALS : op == LLKV_CMP_LT ? LLKV_OP_LT : op == LLKV_CMP_LT_EQ ? LLKV_OP_LE : op == LLKV_CMP_GT ? LLKV_OP_GT : LLKV_OP_GE;
      leaf.value = *lit;
      int32_t rc = filter_leaf(t, &leaf, rows);
      if (rc == LLKV_OK) rc = field_nonnull_rows(t, leaf.field_id, dom);
      return rc;
    }
  }
  /* referenced fields, ascending (ordered_fields) */
  uint32_t fields[64], n_fields = 0;
  for (int side = 0; side < 2; ++side) {
    const llkv_expr_token *e = side ? r : l;
    const uint32_t n = side ? f->cmp_right_len : f->cmp_left_len;
    for (uint32_t i = 0; i < n; ++i) {
      if (e[i].kind != LLKV_TOK_COLUMN) continue;
      uint32_t j = 0;
      while (j < n_fields && fields[j] != e[i].field_id) ++j;
      if (j == n_fields) { if (n_fields == 64) return fail(LLKV_INTERNAL, "too many fields"); fields[n_fields++] = e[i].field_id; }
    }
  }
  if (n_fields == 0) {
    /* no field at all (:354-360; domain :791-796): evaluate_constant_compare :887-907 — both sides evaluated once (evaluate_value over
     * an empty array map: a literal is literal_to_array, literal arithmetic compute_binary, i.e. what simplify folds), compared by
     * compute_compare in their common type.  Some(true) → every row of the table; Some(false) → none; both determined everywhere;
     * None (a NULL side) → nothing matched, nothing determined.  Restated for sides that fold to one numeric or NULL literal. */
    llkv_expr_token fl[64], fr[64];
    uint32_t nl = 0, nr = 0;
    if (f->cmp_left_len > 64 || f->cmp_right_len > 64) return fail(LLKV_INTERNAL, "expression too long");
    int32_t frc = simplify_tokens(l, f->cmp_left_len, fl, &nl);
    if (frc == LLKV_OK) frc = simplify_tokens(r, f->cmp_right_len, fr, &nr);
    if (frc) return frc;
    if (nl != 1 || nr != 1 || !is_numeric_literal_token(&fl[0]) || !is_numeric_literal_token(&fr[0]))
      return fail(LLKV_UNSUPPORTED, "constant comparison over sides that do not fold to a numeric literal");
    const llkv_literal *a = &fl[0].literal, *b = &fr[0].literal;
    idvec hit = {0}, all = {0};
    if (a->tag == LLKV_LIT_NULL || b->tag == LLKV_LIT_NULL) { *rows = hit; *dom = all; return LLKV_OK; } /* (both empty) */
    int m;
    if (a->tag == LLKV_LIT_FLOAT64 || b->tag == LLKV_LIT_FLOAT64) {
      const double x = a->tag == LLKV_LIT_FLOAT64 ? a->f64 : (double)(int64_t)lit_i128(a), y = b->tag == LLKV_LIT_FLOAT64 ? b->f64 : (double)(int64_t)lit_i128(b);
      m = rel_i(f->cmp_op, total_order_key(x), total_order_key(y));
    } else m = rel_i(f->cmp_op, (int64_t)lit_i128(a), (int64_t)lit_i128(b));
    for (uint64_t i = 0; i < t->rows; ++i) { idv_push(&all, i); if (m) idv_push(&hit, i); }
    *rows = hit; /* (every row when the compare holds) */
    *dom = all;
    return LLKV_OK;
  }
  idvec d = {0};
  for (uint32_t j = 0; j < n_fields; ++j) {
    idvec nn;
    int32_t rc = field_nonnull_rows(t, fields[j], &nn);
    if (rc) { idv_free(&d); return rc; }
    if (j == 0) d = nn;
    else { idvec x = idv_and(&d, &nn); idv_free(&d); idv_free(&nn); d = x; }
  }
  idvec matched = {0}, determined = {0};
  int32_t rc = LLKV_OK;
  /* a side that IS the NULL literal: literal_to_array gives a NullArray, coerce_types casts it to the other side's type — all NULL:
   * the compare is NULL on every row (nothing matched, nothing determined), the other side is still evaluated (its errors count) */
  const int l_null = l_lit && l[0].literal.tag == LLKV_LIT_NULL, r_null = r_lit && r[0].literal.tag == LLKV_LIT_NULL;
  /* 4096-row chunks of the domain (CHUNK_SIZE), each gathered and evaluated on its own */
  for (uint64_t c0 = 0; c0 < d.n && rc == LLKV_OK; c0 += 4096) {
    const uint64_t n = d.n - c0 < 4096 ? d.n - c0 : 4096;
    gathered g[64];
    for (uint32_t j = 0; j < n_fields; ++j) { g[j].field_id = fields[j]; g[j].a = gather_column(find_col(t, fields[j]), d.v + c0, n); }
    arr la, ra;
    memset(&la, 0, sizeof la); memset(&ra, 0, sizeof ra);
    if (l_null || r_null) {
      if (!l_null) rc = compare_side(t, l, f->cmp_left_len, g, n_fields, d.v + c0, n, &la);
      if (!r_null) rc = compare_side(t, r, f->cmp_right_len, g, n_fields, d.v + c0, n, &ra);
      arr_free(&la); arr_free(&ra);
      for (uint32_t j = 0; j < n_fields; ++j) arr_free(&g[j].a);
      continue;
    }
    rc = compare_side(t, l, f->cmp_left_len, g, n_fields, d.v + c0, n, &la);
    if (rc == LLKV_OK) rc = compare_side(t, r, f->cmp_right_len, g, n_fields, d.v + c0, n, &ra);
    if (rc == LLKV_OK) {
      /* get_common_type (kernels.rs:179-242) */
      const int lf = la.dtype == LLKV_DT_FLOAT64 || la.dtype == LLKV_DT_FLOAT32, rf = ra.dtype == LLKV_DT_FLOAT64 || ra.dtype == LLKV_DT_FLOAT32;
      int mode; /* 0 = Float64, 1 = signed, 2 = unsigned */
      if (lf || rf) mode = 0;
      else if (is_unsigned_dtype(la.dtype) && is_unsigned_dtype(ra.dtype)) mode = 2;
      else if (is_unsigned_dtype(la.dtype) != is_unsigned_dtype(ra.dtype)) mode = (is_64bit_dtype(la.dtype) || is_64bit_dtype(ra.dtype)) ? 0 : 1;
      else mode = 1;
      for (uint64_t i = 0; i < n; ++i) {
        if (!la.valid[i] || !ra.valid[i]) continue; /* NULL compare → NULL: neither matched nor determined */
        idv_push(&determined, d.v[c0 + i]);
        int m;
        if (mode == 0) m = rel_i(f->cmp_op, total_order_key(side_as_f64(&la, i)), total_order_key(side_as_f64(&ra, i)));
        else if (mode == 1) m = rel_i(f->cmp_op, side_as_i64(&la, i), side_as_i64(&ra, i));
        else m = rel_u(f->cmp_op, (uint64_t)side_as_i64(&la, i), (uint64_t)side_as_i64(&ra, i));
        if (m) idv_push(&matched, d.v[c0 + i]);
      }
    }
    arr_free(&la); arr_free(&ra);
    for (uint32_t j = 0; j < n_fields; ++j) arr_free(&g[j].a);
  }
  idv_free(&d);
  if (rc) { idv_free(&matched); idv_free(&determined); return rc; }
  *rows = matched;
  *dom = determined;
  return LLKV_OK;
}

/* ------------------------------------------------- Expr::InList / Expr::IsNull */
static void collect_expr_fields(const llkv_expr_token *e, uint32_t n, uint32_t *fields, uint32_t *n_fields) {
  for (uint32_t i = 0; i < n; ++i) {
    if (e[i].kind != LLKV_TOK_COLUMN) continue;
    uint32_t j = 0;
    while (j < *n_fields && fields[j] != e[i].field_id) ++j;
    if (j == *n_fields && *n_fields < 64) fields[(*n_fields)++] = e[i].field_id;
  }
}

static int dtype_mode(int32_t a, int32_t b) { /* 0 = Float64, 1 = signed, 2 = unsigned: get_common_type of two sides */
  const int af = a == LLKV_DT_FLOAT64 || a == LLKV_DT_FLOAT32, bf = b == LLKV_DT_FLOAT64 || b == LLKV_DT_FLOAT32;
  if (af || bf) return 0;
  if (is_unsigned_dtype(a) && is_unsigned_dtype(b)) return 2;
  if (is_unsigned_dtype(a) != is_unsigned_dtype(b)) return (is_64bit_dtype(a) || is_64bit_dtype(b)) ? 0 : 1;
  return 1;
}

/* evaluate_in_list_over_rows llkv-scan/src/predicate.rs:443-560: rows where every referenced field is present;
 * the target is re-coerced with every item (`target_array = new_target`), `eq` per item (floats by totalOrder),
 * or_kleene over the items, `not` when negated; a NULL result is neither matched nor determined. */
/* A side without a field, as evaluate_value leaves it: one numeric or NULL literal (what simplify folds literal arithmetic to). */
static int32_t constant_side(const llkv_expr_token *e, uint32_t n, llkv_literal *out) {
  llkv_expr_token fl[64];
  uint32_t m = 0;
  if (n > 64) return fail(LLKV_INTERNAL, "expression too long");
  int32_t rc = simplify_tokens(e, n, fl, &m);
  if (rc) return rc;
  if (m != 1 || !is_numeric_literal_token(&fl[0])) return fail(LLKV_UNSUPPORTED, "constant predicate over a side that does not fold to a numeric literal");
  *out = fl[0].literal;
  return LLKV_OK;
}
static void all_rows_of(const orc_table *t, idvec *v) { for (uint64_t i = 0; i < t->rows; ++i) idv_push(v, i); }

static int32_t in_list_rows(const orc_table *t, const llkv_filter *f, idvec *rows, idvec *dom) {
  if (!f->cmp_left || !f->cmp_left_len) return fail(LLKV_INVALID_ARGUMENT, "IN list needs a target expression");
  uint32_t fields[64], n_fields = 0;
  collect_expr_fields(f->cmp_left, f->cmp_left_len, fields, &n_fields);
  for (uint32_t k = 0; k < f->list_len; ++k) collect_expr_fields(f->list_exprs[k], f->list_expr_lens[k], fields, &n_fields);
  if (n_fields == 0) {
    /* evaluate_constant_in_list :909-963 (rows) and collect_in_list_domain_rows :832-836 (its second result): a NULL target matches and
     * determines nothing; the first item equal to the target decides (Some(!negated)); else a NULL item leaves it None; else Some(negated) */
    llkv_literal tgt, item;
    idvec hit = {0}, all = {0};
    int32_t crc = constant_side(f->cmp_left, f->cmp_left_len, &tgt);
    if (crc) return crc;
    if (tgt.tag == LLKV_LIT_NULL) { *rows = hit; *dom = all; return LLKV_OK; }
    int matched = 0, saw_null = 0;
    for (uint32_t k = 0; k < f->list_len && !matched; ++k) {
      if ((crc = constant_side(f->list_exprs[k], f->list_expr_lens[k], &item))) return crc;
      if (item.tag == LLKV_LIT_NULL) { saw_null = 1; continue; }
      if (tgt.tag == LLKV_LIT_FLOAT64 || item.tag == LLKV_LIT_FLOAT64) {
        const double x = tgt.tag == LLKV_LIT_FLOAT64 ? tgt.f64 : (double)(int64_t)lit_i128(&tgt), y = item.tag == LLKV_LIT_FLOAT64 ? item.f64 : (double)(int64_t)lit_i128(&item);
        matched = total_order_key(x) == total_order_key(y);
      } else matched = (int64_t)lit_i128(&tgt) == (int64_t)lit_i128(&item);
    }
    if (!matched && saw_null) { *rows = hit; *dom = all; return LLKV_OK; }
    all_rows_of(t, &all);
    if (matched ? !f->negated : f->negated) all_rows_of(t, &hit);
    *rows = hit; *dom = all;
    return LLKV_OK;
  }
  idvec d = {0};
  for (uint32_t j = 0; j < n_fields; ++j) {
    idvec nn;
    int32_t rc = field_nonnull_rows(t, fields[j], &nn);
    if (rc) { idv_free(&d); return rc; }
    if (j == 0) d = nn;
    else { idvec x = idv_and(&d, &nn); idv_free(&d); idv_free(&nn); d = x; }
  }
  idvec matched = {0}, determined = {0};
  int32_t rc = LLKV_OK;
  for (uint64_t c0 = 0; c0 < d.n && rc == LLKV_OK; c0 += 4096) {
    const uint64_t n = d.n - c0 < 4096 ? d.n - c0 : 4096;
    gathered g[64];
    for (uint32_t j = 0; j < n_fields; ++j) { g[j].field_id = fields[j]; g[j].a = gather_column(find_col(t, fields[j]), d.v + c0, n); }
    arr tgt;
    memset(&tgt, 0, sizeof tgt);
    rc = compare_side(t, f->cmp_left, f->cmp_left_len, g, n_fields, d.v + c0, n, &tgt);
    int8_t *acc = xmalloc(n ? n : 1); /* Kleene: 0 false, 1 true, -1 NULL; starts as "no item yet" = false */
    memset(acc, 0, n);
    int have = 0;
    for (uint32_t k = 0; k < f->list_len && rc == LLKV_OK; ++k) {
      arr it;
      memset(&it, 0, sizeof it);
      rc = compare_side(t, f->list_exprs[k], f->list_expr_lens[k], g, n_fields, d.v + c0, n, &it);
      if (rc) break;
      const int mode = dtype_mode(tgt.dtype, it.dtype);
      if (mode == 0 && tgt.dtype != LLKV_DT_FLOAT64) { /* the coerced target replaces the target */
        arr c = cast_to(&tgt, LLKV_DT_FLOAT64);
        arr_free(&tgt);
        tgt = c;
      }
      for (uint64_t i = 0; i < n; ++i) {
        int8_t eq;
        if (!tgt.valid[i] || !it.valid[i]) eq = -1;
        else if (mode == 0) eq = total_order_key(side_as_f64(&tgt, i)) == total_order_key(side_as_f64(&it, i));
        else eq = side_as_i64(&tgt, i) == side_as_i64(&it, i);
        if (!have) acc[i] = eq;
        else acc[i] = (acc[i] == 1 || eq == 1) ? 1 : (acc[i] == -1 || eq == -1) ? -1 : 0; /* or_kleene */
      }
      have = 1;
      arr_free(&it);
    }
    if (rc == LLKV_OK)
      for (uint64_t i = 0; i < n; ++i) {
        if (acc[i] < 0) continue;
        const int v = f->negated ? !acc[i] : acc[i];
        idv_push(&determined, d.v[c0 + i]);
        if (v) idv_push(&matched, d.v[c0 + i]);
      }
    free(acc);
    arr_free(&tgt);
    for (uint32_t j = 0; j < n_fields; ++j) arr_free(&g[j].a);
  }
  idv_free(&d);
  if (rc) { idv_free(&matched); idv_free(&determined); return rc; }
  *rows = matched;
  *dom = determined;
  return LLKV_OK;
}

/* collect_row_ids_for_is_null llkv-scan/src/predicate.rs:249-331; domain: PushIsNullDomain :744-767. */
static int32_t is_null_expr_rows(const orc_table *t, const llkv_filter *f, idvec *rows, idvec *dom) {
  if (!f->cmp_left || !f->cmp_left_len) return fail(LLKV_INVALID_ARGUMENT, "IS NULL needs an expression");
  if (f->cmp_left_len == 1 && f->cmp_left[0].kind == LLKV_TOK_COLUMN) {
    llkv_filter leaf;
    memset(&leaf, 0, sizeof leaf);
    leaf.field_id = f->cmp_left[0].field_id;
    leaf.op = f->negated ? LLKV_OP_IS_NOT_NULL : LLKV_OP_IS_NULL;
    int32_t rc = filter_leaf(t, &leaf, rows);
    if (rc == LLKV_OK) rc = field_nonnull_rows(t, leaf.field_id, dom);
    return rc;
  }
  uint32_t fields[64], n_fields = 0;
  collect_expr_fields(f->cmp_left, f->cmp_left_len, fields, &n_fields);
  if (n_fields == 0) { /* :276-284: the constant is NULL or not — every row of the table or none; determined everywhere (:746-749) */
    llkv_literal v;
    int32_t crc = constant_side(f->cmp_left, f->cmp_left_len, &v);
    if (crc) return crc;
    idvec hit = {0}, all = {0};
    all_rows_of(t, &all);
    if ((v.tag == LLKV_LIT_NULL) != (f->negated != 0)) all_rows_of(t, &hit);
    *rows = hit; *dom = all;
    return LLKV_OK;
  }
  idvec uni = {0}, inter = {0};
  for (uint32_t j = 0; j < n_fields; ++j) {
    idvec nn;
    int32_t rc = field_nonnull_rows(t, fields[j], &nn);
    if (rc) { idv_free(&uni); idv_free(&inter); return rc; }
    if (j == 0) { uni = nn; inter = idv_and(&nn, &nn); }
    else {
      idvec u = idv_or(&uni, &nn), x = idv_and(&inter, &nn);
      idv_free(&uni); idv_free(&inter); idv_free(&nn);
      uni = u; inter = x;
    }
  }
  idvec res = {0};
  int32_t rc = LLKV_OK;
  for (uint64_t c0 = 0; c0 < uni.n && rc == LLKV_OK; c0 += 4096) {
    const uint64_t n = uni.n - c0 < 4096 ? uni.n - c0 : 4096;
    gathered g[64];
    for (uint32_t j = 0; j < n_fields; ++j) { g[j].field_id = fields[j]; g[j].a = gather_column(find_col(t, fields[j]), uni.v + c0, n); }
    arr v;
    memset(&v, 0, sizeof v);
    rc = eval_program(t, f->cmp_left, f->cmp_left_len, g, n_fields, n, &v);
    if (rc == LLKV_OK)
      for (uint64_t i = 0; i < n; ++i) {
        const int is_null = !v.valid[i];
        if ((is_null && !f->negated) || (!is_null && f->negated)) idv_push(&res, uni.v[c0 + i]);
      }
    arr_free(&v);
    for (uint32_t j = 0; j < n_fields; ++j) arr_free(&g[j].a);
  }
  idv_free(&uni);
  if (rc) { idv_free(&res); idv_free(&inter); return rc; }
  *rows = res;
  *dom = inter;
  return LLKV_OK;
}

/* ----------------------------------------------------------------- scan */
static int is_simple_column(const llkv_expr_token *e, uint32_t n) { return n == 1 && e[0].kind == LLKV_TOK_COLUMN; }

typedef struct proj_plan {
  int computed;
  uint32_t field_id;
  const llkv_expr_token *expr;
  uint32_t expr_len;
} proj_plan;

typedef void (*window_cb)(const arr *cols, uint32_t n_cols, const uint64_t *row_ids, uint64_t n, void *user, int32_t *rc);

/* execute_scan llkv-scan/src/execute.rs:47-295; window materialisation
 * row_stream.rs:369-438,451-623. */
static int32_t order_row_ids(const orc_table *t, const llkv_scan_options *o, uint64_t *ids, uint64_t n);

static int32_t scan_core_ordered(const orc_table *t, const proj_plan *projs, uint32_t n_projs,
                                 const llkv_filter *filters, uint32_t n_filters, const llkv_eval_op *ops,
                                 uint32_t n_ops, int include_nulls, const llkv_scan_options *order, window_cb cb, void *user);
static int32_t scan_core(const orc_table *t, const proj_plan *projs, uint32_t n_projs,
                         const llkv_filter *filters, uint32_t n_filters, const llkv_eval_op *ops,
                         uint32_t n_ops, int include_nulls, window_cb cb, void *user) {
  return scan_core_ordered(t, projs, n_projs, filters, n_filters, ops, n_ops, include_nulls, NULL, cb, user);
}
static int32_t scan_core_ordered(const orc_table *t, const proj_plan *projs, uint32_t n_projs,
                                 const llkv_filter *filters, uint32_t n_filters, const llkv_eval_op *ops,
                                 uint32_t n_ops, int include_nulls, const llkv_scan_options *order, window_cb cb, void *user) {
  /* unique fields to gather (direct columns + expression inputs) */
  uint32_t fields[128], n_fields = 0;
  for (uint32_t p = 0; p < n_projs; ++p) {
    uint32_t cnt = projs[p].computed ? projs[p].expr_len : 1;
    for (uint32_t k = 0; k < cnt; ++k) {
      uint32_t fid;
      if (projs[p].computed) { if (projs[p].expr[k].kind != LLKV_TOK_COLUMN) continue; fid = projs[p].expr[k].field_id; }
      else fid = projs[p].field_id;
      uint32_t j = 0;
      while (j < n_fields && fields[j] != fid) ++j;
      if (j == n_fields) { if (n_fields == 128) return fail(LLKV_INTERNAL, "too many fields"); fields[n_fields++] = fid; }
    }
  }
  for (uint32_t j = 0; j < n_fields; ++j)
    if (!find_col(t, fields[j])) return fail(LLKV_NOT_FOUND, "field %u not found", fields[j]);

  uint64_t *ids = NULL, n_ids = 0;
  int32_t rc = orc_filter_row_ids(t, filters, n_filters, ops, n_ops, &ids, &n_ids); /* execute.rs:219 */
  if (rc) return rc;
  if (order && order->order_enabled && n_ids && (rc = order_row_ids(t, order, ids, n_ids))) { free(ids); return rc; } /* execute.rs:221-237 */

  /* all matching row ids are materialised, then cut into 65 536-row windows
   * (row_stream.rs:243-254,289-299) */
  for (uint64_t w0 = 0; w0 < n_ids && rc == LLKV_OK; w0 += ROW_STREAM_CHUNK_SIZE) {
    uint64_t wn = n_ids - w0 < ROW_STREAM_CHUNK_SIZE ? n_ids - w0 : ROW_STREAM_CHUNK_SIZE;
    uint64_t *wids = xmalloc(wn * sizeof(uint64_t));
    uint64_t kept = 0;
    /* GatherNullPolicy::DropNulls — drop rows whose gathered fields are ALL NULL
     * (store/projection.rs:40-48,1326-1330) */
    for (uint64_t i = 0; i < wn; ++i) {
      uint64_t rid = ids[w0 + i];
      int keep = include_nulls || n_fields == 0;
      for (uint32_t j = 0; j < n_fields && !keep; ++j) keep = col_valid(find_col(t, fields[j]), rid);
      if (keep) wids[kept++] = rid;
    }
    if (kept == 0) { free(wids); continue; } /* empty batches are not emitted, execute.rs:289-291 */
    gathered *g = xcalloc(n_fields ? n_fields : 1, sizeof(gathered));
    for (uint32_t j = 0; j < n_fields; ++j) { g[j].field_id = fields[j]; g[j].a = gather_column(find_col(t, fields[j]), wids, kept); }
    arr *outs = xcalloc(n_projs ? n_projs : 1, sizeof(arr));
    uint8_t *owned = xcalloc(n_projs ? n_projs : 1, 1);
    for (uint32_t p = 0; p < n_projs && rc == LLKV_OK; ++p) {
      if (!projs[p].computed) outs[p] = *find_gathered(g, n_fields, projs[p].field_id);
      else { rc = eval_program(t, projs[p].expr, projs[p].expr_len, g, n_fields, kept, &outs[p]); owned[p] = rc == LLKV_OK; }
    }
    if (rc == LLKV_OK) cb(outs, n_projs, wids, kept, user, &rc);
    for (uint32_t p = 0; p < n_projs; ++p) if (owned[p]) arr_free(&outs[p]);
    for (uint32_t j = 0; j < n_fields; ++j) arr_free(&g[j].a);
    free(outs); free(owned); free(g); free(wids);
  }
  free(ids);
  return rc;
}

typedef struct stream_ctx {
  orc_on_batch cb;
  void *user;
  int include_row_ids;
} stream_ctx;

static void stream_window(const arr *cols, uint32_t n_cols, const uint64_t *row_ids, uint64_t n, void *user, int32_t *rc) {
  (void)rc;
  stream_ctx *s = user;
  orc_batch_column *bc = xcalloc(n_cols ? n_cols : 1, sizeof *bc);
  for (uint32_t i = 0; i < n_cols; ++i) {
    bc[i].dtype = cols[i].dtype;
    bc[i].values = cols[i].values;
    bc[i].valid = cols[i].valid;
    bc[i].strings = (const char *const *)cols[i].strings;
    bc[i].precision = cols[i].precision; bc[i].scale = cols[i].scale;
  }
  orc_batch b = {n, n_cols, bc, s->include_row_ids ? row_ids : NULL};
  s->cb(&b, s->user);
  free(bc);
}

int32_t orc_scan_stream(const orc_table *t, const llkv_projection *projections, uint32_t n_projections,
                        const llkv_filter *filters, uint32_t n_filters, const llkv_eval_op *ops,
                        uint32_t n_ops, const llkv_scan_options *options, orc_on_batch on_batch, void *user) {
  if (n_projections == 0) return fail(LLKV_INVALID_ARGUMENT, "scan requires at least one projection");
  proj_plan *pp = xcalloc(n_projections, sizeof *pp);
  for (uint32_t i = 0; i < n_projections; ++i) {
    pp[i].computed = projections[i].computed;
    pp[i].field_id = projections[i].field_id;
    pp[i].expr = projections[i].expr;
    pp[i].expr_len = projections[i].expr_len;
  }
  stream_ctx s = {on_batch, user, options ? options->include_row_ids : 0};
  int32_t rc = scan_core_ordered(t, pp, n_projections, filters, n_filters, ops, n_ops, options ? options->include_nulls : 0, options, stream_window, &s);
  free(pp);
  return rc;
}

/* sort_row_ids_with_order llkv-scan/src/ordering.rs:16-140: gather the ORDER BY column for the selected rows and
 * arrow sort_to_indices {descending, nulls_first}; ties are left unspecified by arrow — kept in row-id order here. */
typedef struct ord_ent { int is_null; int64_t i; const char *s; uint64_t id; uint64_t pos; } ord_ent;
static int g_ord_desc, g_ord_nulls_first, g_ord_str;
static int ord_cmp(const void *a, const void *b) {
  const ord_ent *x = a, *y = b;
  if (x->is_null != y->is_null) return (x->is_null ? -1 : 1) * (g_ord_nulls_first ? 1 : -1);
  if (!x->is_null) {
    int c = g_ord_str ? strcmp(x->s, y->s) : (x->i < y->i ? -1 : x->i > y->i);
    if (c) return g_ord_desc ? -c : c;
  }
  return x->pos < y->pos ? -1 : x->pos > y->pos;
}
static int32_t order_row_ids(const orc_table *t, const llkv_scan_options *o, uint64_t *ids, uint64_t n) {
  const orc_column *c = find_col(t, o->order_field);
  if (!c) return fail(LLKV_NOT_FOUND, "ORDER BY field %u not found", o->order_field);
  switch (o->order_transform) {
  case LLKV_ORDER_IDENTITY_INT64: if (c->dtype != LLKV_DT_INT64) return fail(LLKV_INVALID_ARGUMENT, "ORDER BY expected INT64 column for IdentityInt64 transform"); break;
  case LLKV_ORDER_IDENTITY_INT32: if (c->dtype != LLKV_DT_INT32) return fail(LLKV_INVALID_ARGUMENT, "ORDER BY expected INT32 column for IdentityInt32 transform"); break;
  case LLKV_ORDER_IDENTITY_UTF8: if (c->dtype != LLKV_DT_UTF8) return fail(LLKV_INVALID_ARGUMENT, "ORDER BY expected UTF8 column for IdentityUtf8 transform"); break;
  case LLKV_ORDER_CAST_UTF8_TO_INTEGER: if (c->dtype != LLKV_DT_UTF8) return fail(LLKV_INVALID_ARGUMENT, "ORDER BY CAST expects a UTF8 column"); break;
  default: return fail(LLKV_INVALID_ARGUMENT, "unknown ORDER BY transform");
  }
  ord_ent *e = xmalloc(n * sizeof *e);
  char **owned = xcalloc(n, sizeof(char *));
  for (uint64_t k = 0; k < n; ++k) {
    const uint64_t r = ids[k];
    e[k].id = r; e[k].pos = k; e[k].i = 0; e[k].s = NULL;
    e[k].is_null = !col_valid(c, r);
    if (e[k].is_null) continue;
    if (c->dtype == LLKV_DT_INT64) e[k].i = ((const int64_t *)c->values)[r];
    else if (c->dtype == LLKV_DT_INT32) e[k].i = ((const int32_t *)c->values)[r];
    else {
      size_t len = (size_t)(c->offsets[r + 1] - c->offsets[r]);
      owned[k] = xmalloc(len + 1);
      memcpy(owned[k], c->data + c->offsets[r], len);
      owned[k][len] = 0;
      e[k].s = owned[k];
      if (o->order_transform == LLKV_ORDER_CAST_UTF8_TO_INTEGER) { /* str::parse::<i64>, failures → NULL */
        char *end;
        errno = 0;
        long long v = strtoll(owned[k], &end, 10);
        int ok = len > 0 && *end == 0 && errno == 0 && !(owned[k][0] == ' ' || owned[k][0] == '\t');
        if (ok) e[k].i = v; else e[k].is_null = 1;
      }
    }
  }
  g_ord_desc = o->order_descending != 0;
  g_ord_nulls_first = o->order_nulls_first != 0;
  g_ord_str = o->order_transform == LLKV_ORDER_IDENTITY_UTF8;
  qsort(e, n, sizeof *e, ord_cmp);
  for (uint64_t k = 0; k < n; ++k) ids[k] = e[k].id;
  for (uint64_t k = 0; k < n; ++k) free(owned[k]);
  free(owned); free(e);
  return LLKV_OK;
}

/* ------------------------------------------------------------ accumulators */
/* llkv-aggregate/src/lib.rs: state :95-249, update :759-1477, finalize :1488-1939. */
enum {
  ACC_COUNT_STAR, ACC_COUNT_COLUMN, ACC_SUM_I64, ACC_SUM_F64, ACC_TOTAL_I64, ACC_TOTAL_F64,
  ACC_AVG_I64, ACC_AVG_F64, ACC_MIN_I64, ACC_MIN_F64, ACC_MAX_I64, ACC_MAX_F64, ACC_COUNT_NULLS,
  ACC_SUM_DEC, ACC_TOTAL_DEC, ACC_AVG_DEC, ACC_MIN_DEC, ACC_MAX_DEC /* Decimal128: i128 state, :925-967,1071-1088,1236-1259,1332-1352,1400-1420 */
};
typedef struct acc {
  int kind;
  int64_t i;      /* count / i64 sum / min / max */
  double f;       /* f64 sum / min / max */
  int64_t count;  /* avg count, count_nulls total rows */
  int has;        /* has_values / saw_value / Some(..) */
  __int128 d;     /* Decimal128 sum / min / max */
  int32_t precision, scale;
  /* DISTINCT variants (:103-204): the `seen` set, kept as the arrival-ordered list of values (Int by value,
   * Float by bit pattern :252-331); first appearances are folded in order at finalize */
  int distinct, distinct_kind, distinct_f64;
  uint64_t *seen;
  uint64_t n_seen, cap_seen;
  int32_t distinct_rc;
  /* … and for Str / Bool / Date / Decimal keys (DistinctKey :252-331) the set itself, in order of first insertion: the key's
   * bytes, its numeric image (what `*sum += v` adds for a new key, :904-921) or raw i128 */
  int32_t distinct_dtype;
  struct dkey { char *bytes; uint32_t len; double num; __int128 raw; } *keys;
  uint64_t n_keys, cap_keys;
  uint64_t *key_slots; /* open addressing over `keys`: index + 1, 0 = empty */
  uint64_t n_slots;
} acc;

static void acc_free_distinct(acc *a) {
  free(a->seen);
  for (uint64_t i = 0; i < a->n_keys; ++i) free(a->keys[i].bytes);
  free(a->keys);
  free(a->key_slots);
  a->seen = NULL; a->keys = NULL; a->key_slots = NULL; a->n_keys = a->cap_keys = a->n_slots = 0;
}
static uint64_t bytes_hash(const char *b, uint32_t n) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (uint32_t i = 0; i < n; ++i) h = (h ^ (unsigned char)b[i]) * 0x100000001b3ull;
  return h;
}
/* seen.insert(key): 1 when the key is new (then *at = its position in insertion order) */
static int acc_insert_key(acc *a, const char *bytes, uint32_t len, uint64_t *at) {
  if ((a->n_keys + 1) * 2 > a->n_slots) {
    const uint64_t ns = a->n_slots ? a->n_slots * 2 : 1024;
    free(a->key_slots);
    a->key_slots = xcalloc(ns, sizeof(uint64_t));
    a->n_slots = ns;
    for (uint64_t i = 0; i < a->n_keys; ++i) {
      uint64_t p = bytes_hash(a->keys[i].bytes, a->keys[i].len) & (ns - 1);
      while (a->key_slots[p]) p = (p + 1) & (ns - 1);
      a->key_slots[p] = i + 1;
    }
  }
  uint64_t p = bytes_hash(bytes, len) & (a->n_slots - 1);
  while (a->key_slots[p]) {
    const struct dkey *k = &a->keys[a->key_slots[p] - 1];
    if (k->len == len && memcmp(k->bytes, bytes, len) == 0) return 0;
    p = (p + 1) & (a->n_slots - 1);
  }
  if (a->n_keys == a->cap_keys) { a->cap_keys = a->cap_keys ? a->cap_keys * 2 : 256; a->keys = xrealloc(a->keys, a->cap_keys * sizeof *a->keys); }
  struct dkey *k = &a->keys[a->n_keys];
  k->bytes = xmalloc(len ? len : 1);
  memcpy(k->bytes, bytes, len);
  k->len = len; k->num = 0.0; k->raw = 0;
  a->key_slots[p] = a->n_keys + 1;
  *at = a->n_keys++;
  return 1;
}

/* `str::trim` + `<f64 as FromStr>::from_str` (Rust core, published grammar): Unicode White_Space is trimmed; a float is
 * Sign? ( "inf" | "infinity" | "nan" | Number ) in any case, Number = ( Digit+ | Digit+ "." Digit* | Digit* "." Digit+ )
 * Exp?, Exp = ("e" | "E") Sign? Digit+ — no hexadecimal forms, nothing else around it.  Anything else → 0.0. */
static size_t ws_len(const char *s, size_t n, size_t i, int backwards) {
  static const char *const multi[] = {"\xC2\x85", "\xC2\xA0", "\xE1\x9A\x80", "\xE2\x80\x80", "\xE2\x80\x81", "\xE2\x80\x82", "\xE2\x80\x83",
                                      "\xE2\x80\x84", "\xE2\x80\x85", "\xE2\x80\x86", "\xE2\x80\x87", "\xE2\x80\x88", "\xE2\x80\x89", "\xE2\x80\x8A",
                                      "\xE2\x80\xA8", "\xE2\x80\xA9", "\xE2\x80\xAF", "\xE2\x81\x9F", "\xE3\x80\x80"};
  const unsigned char c = (unsigned char)s[i];
  if (c == ' ' || (c >= 9 && c <= 13)) return 1;
  for (size_t m = 0; m < sizeof multi / sizeof multi[0]; ++m) {
    const size_t len = strlen(multi[m]);
    if (!backwards) { if (i + len <= n && memcmp(s + i, multi[m], len) == 0) return len; }
    else if (i + 1 >= len && memcmp(s + i + 1 - len, multi[m], len) == 0) return len;
  }
  return 0;
}
static int word_is(const char *s, size_t n, const char *w) {
  if (strlen(w) != n) return 0;
  for (size_t k = 0; k < n; ++k) if (tolower((unsigned char)s[k]) != w[k]) return 0;
  return 1;
}
static double rust_parse_f64_or_zero(const char *text) {
  size_t b = 0, e = strlen(text), len;
  while (b < e && (len = ws_len(text, e, b, 0))) b += len;
  while (e > b && (len = ws_len(text, e, e - 1, 1))) e -= len;
  const char *t = text + b;
  size_t n = e - b, i = 0;
  int neg = 0;
  if (i < n && (t[i] == '+' || t[i] == '-')) neg = t[i++] == '-';
  if (word_is(t + i, n - i, "inf") || word_is(t + i, n - i, "infinity")) return neg ? -INFINITY : INFINITY;
  if (word_is(t + i, n - i, "nan")) return neg ? -NAN : NAN;
  size_t k = i, int_digits = 0, frac_digits = 0, exp_digits = 0;
  while (k < n && isdigit((unsigned char)t[k])) { ++k; ++int_digits; }
  if (k < n && t[k] == '.') { ++k; while (k < n && isdigit((unsigned char)t[k])) { ++k; ++frac_digits; } }
  if (int_digits + frac_digits == 0) return 0.0;
  if (k < n && (t[k] == 'e' || t[k] == 'E')) {
    ++k;
    if (k < n && (t[k] == '+' || t[k] == '-')) ++k;
    while (k < n && isdigit((unsigned char)t[k])) { ++k; ++exp_digits; }
    if (exp_digits == 0) return 0.0;
  }
  if (k != n) return 0.0;
  char buf[512];
  if (n >= sizeof buf) return 0.0; /* (a number of 500 characters: not in any fixture) */
  memcpy(buf, t, n);
  buf[n] = 0;
  return strtod(buf, NULL);
}

/* array_value_to_numeric :400-449 */
static int32_t value_to_numeric(const arr *a, uint64_t i, double *out) {
  switch (a->dtype) {
  case LLKV_DT_INT64: *out = (double)((int64_t *)a->values)[i]; return LLKV_OK;
  case LLKV_DT_FLOAT64: *out = ((double *)a->values)[i]; return LLKV_OK;
  case LLKV_DT_UTF8: *out = rust_parse_f64_or_zero(a->strings[i]); return LLKV_OK; /* s.trim().parse::<f64>().unwrap_or(0.0) :426-434 */
  case LLKV_DT_BOOLEAN: *out = ((uint8_t *)a->values)[i] ? 1.0 : 0.0; return LLKV_OK;
  default: return fail(LLKV_INVALID_ARGUMENT, "Numeric coercion not supported for column type %s", dtype_name(a->dtype));
  }
}

/* kind selection: new_with_projection_index :463-748 after validate_aggregate_type
 * (llkv-executor/src/lib.rs:5946-5988). */
static int32_t acc_new(int32_t agg_kind, int32_t input_dtype, acc *out) {
  memset(out, 0, sizeof *out);
  if (agg_kind == LLKV_AGG_COUNT_STAR) { out->kind = ACC_COUNT_STAR; return LLKV_OK; }
  if (agg_kind == LLKV_AGG_COUNT) { out->kind = ACC_COUNT_COLUMN; return LLKV_OK; }
  if (agg_kind == LLKV_AGG_COUNT_NULLS) { out->kind = ACC_COUNT_NULLS; return LLKV_OK; }
  const char *fn = agg_kind == LLKV_AGG_SUM ? "SUM" : agg_kind == LLKV_AGG_TOTAL ? "TOTAL" : agg_kind == LLKV_AGG_AVG ? "AVG" : agg_kind == LLKV_AGG_MIN ? "MIN" : "MAX";
  int32_t dt;
  if (input_dtype == LLKV_DT_DECIMAL128) { /* (precision, scale) are filled in by the caller */
    out->kind = agg_kind == LLKV_AGG_SUM ? ACC_SUM_DEC : agg_kind == LLKV_AGG_TOTAL ? ACC_TOTAL_DEC : agg_kind == LLKV_AGG_AVG ? ACC_AVG_DEC : agg_kind == LLKV_AGG_MIN ? ACC_MIN_DEC : ACC_MAX_DEC;
    return LLKV_OK;
  }
  switch (input_dtype) {
  case LLKV_DT_INT64: case LLKV_DT_FLOAT64: dt = input_dtype; break;
  case LLKV_DT_UTF8: case LLKV_DT_BOOLEAN: case LLKV_DT_DATE32: case LLKV_DT_NULL: dt = LLKV_DT_FLOAT64; break;
  default: return fail(LLKV_INVALID_ARGUMENT, "%s aggregate not supported for column type %s", fn, dtype_name(input_dtype));
  }
  int is_i = dt == LLKV_DT_INT64;
  switch (agg_kind) {
  case LLKV_AGG_SUM: out->kind = is_i ? ACC_SUM_I64 : ACC_SUM_F64; break;
  case LLKV_AGG_TOTAL: out->kind = is_i ? ACC_TOTAL_I64 : ACC_TOTAL_F64; break;
  case LLKV_AGG_AVG: out->kind = is_i ? ACC_AVG_I64 : ACC_AVG_F64; break;
  case LLKV_AGG_MIN: out->kind = is_i ? ACC_MIN_I64 : ACC_MIN_F64; break;
  case LLKV_AGG_MAX: out->kind = is_i ? ACC_MAX_I64 : ACC_MAX_F64; break;
  default: return fail(LLKV_UNSUPPORTED, "aggregate kind %d", agg_kind);
  }
  return LLKV_OK;
}

static int32_t acc_new_distinct(int32_t agg_kind, int32_t input_dtype, acc *out) {
  memset(out, 0, sizeof *out);
  /* DistinctKey::from_array :261-331: Int64, Float64, Utf8, Boolean, Date32, Decimal128 */
  if (input_dtype != LLKV_DT_INT64 && input_dtype != LLKV_DT_FLOAT64 && input_dtype != LLKV_DT_UTF8 && input_dtype != LLKV_DT_BOOLEAN &&
      input_dtype != LLKV_DT_DATE32 && input_dtype != LLKV_DT_DECIMAL128)
    return fail(LLKV_INVALID_ARGUMENT, "COUNT(DISTINCT) is not supported for column type %s", dtype_name(input_dtype));
  out->distinct_dtype = input_dtype;
  if (agg_kind != LLKV_AGG_COUNT && agg_kind != LLKV_AGG_SUM && agg_kind != LLKV_AGG_TOTAL && agg_kind != LLKV_AGG_AVG)
    return fail(LLKV_UNSUPPORTED, "DISTINCT form of aggregate kind %d", agg_kind);
  out->distinct = 1;
  out->distinct_kind = agg_kind;
  out->distinct_f64 = input_dtype == LLKV_DT_FLOAT64;
  return LLKV_OK;
}

typedef struct seen_ent { uint64_t v; uint64_t idx; } seen_ent;
static int seen_cmp(const void *a, const void *b) {
  const seen_ent *x = a, *y = b;
  if (x->v != y->v) return x->v < y->v ? -1 : 1;
  return x->idx < y->idx ? -1 : x->idx > y->idx;
}
static int idx_cmp(const void *a, const void *b) {
  const seen_ent *x = a, *y = b;
  return x->idx < y->idx ? -1 : x->idx > y->idx;
}

/* finalize of the DISTINCT accumulators: :1502-1510 (count), :1531-1543,1555-1565 (sum), :1619-1637 (total),
 * :1684-1720 (avg).  Returns the error update() would have raised. */
static int32_t acc_finalize_distinct(const acc *a, llkv_value *out) {
  memset(out, 0, sizeof *out);
  if (a->distinct_dtype != LLKV_DT_INT64 && a->distinct_dtype != LLKV_DT_FLOAT64) {
    const uint64_t m = a->n_keys;
    if (a->distinct_rc) return a->distinct_rc; /* the error update() raised */
    if (a->distinct_kind == LLKV_AGG_COUNT) { out->dtype = LLKV_DT_INT64; out->i64 = (int64_t)m; return LLKV_OK; } /* :1502-1510 */
    if (a->distinct_dtype == LLKV_DT_DECIMAL128) { /* :1583-1612, 1656-1672, 1762-1800 */
      out->dtype = LLKV_DT_DECIMAL128; out->precision = a->precision; out->scale = a->scale;
      if (m == 0) { out->is_null = a->distinct_kind != LLKV_AGG_TOTAL; return LLKV_OK; }
      __int128 v = a->d;
      if (a->distinct_kind == LLKV_AGG_AVG) { /* half away from zero */
        const __int128 cnt = (__int128)m;
        __int128 avg = v / cnt, rem = v % cnt;
        if ((rem < 0 ? -rem : rem) * 2 >= cnt) avg += v >= 0 ? 1 : -1;
        v = avg;
      }
      out->i64 = (int64_t)(uint64_t)(unsigned __int128)v;
      out->i64_hi = (int64_t)(v >> 64);
      return LLKV_OK;
    }
    out->dtype = LLKV_DT_FLOAT64; /* Sum / Total / AvgDistinctFloat64 :1544-1565, 1638-1655, 1721-1760 */
    if (a->distinct_kind == LLKV_AGG_TOTAL) { out->f64 = a->f; return LLKV_OK; }
    if (m == 0) { out->is_null = 1; return LLKV_OK; }
    out->f64 = a->distinct_kind == LLKV_AGG_AVG ? a->f / (double)m : a->f;
    return LLKV_OK;
  }
  seen_ent *e = xmalloc((a->n_seen ? a->n_seen : 1) * sizeof *e);
  for (uint64_t i = 0; i < a->n_seen; ++i) { e[i].v = a->seen[i]; e[i].idx = i; }
  qsort(e, a->n_seen, sizeof *e, seen_cmp);
  uint64_t m = 0;
  for (uint64_t i = 0; i < a->n_seen; ++i) if (i == 0 || e[i].v != e[i - 1].v) e[m++] = e[i]; /* first appearance of each key */
  qsort(e, m, sizeof *e, idx_cmp);
  int32_t rc = LLKV_OK;
  double fs = 0.0;
  int64_t is = 0;
  for (uint64_t i = 0; i < m && rc == LLKV_OK; ++i) {
    if (a->distinct_f64) { double v; memcpy(&v, &e[i].v, 8); fs += v; }
    else if (a->distinct_kind == LLKV_AGG_TOTAL) fs += (double)(int64_t)e[i].v;
    else if (a->distinct_kind != LLKV_AGG_COUNT && __builtin_add_overflow(is, (int64_t)e[i].v, &is))
      rc = fail(LLKV_INVALID_ARGUMENT, a->distinct_kind == LLKV_AGG_AVG ? "AVG(DISTINCT) aggregate sum exceeds i64 range" : "integer overflow");
  }
  free(e);
  if (rc) return rc;
  switch (a->distinct_kind) {
  case LLKV_AGG_COUNT: out->dtype = LLKV_DT_INT64; out->i64 = (int64_t)m; break;
  case LLKV_AGG_TOTAL: out->dtype = LLKV_DT_FLOAT64; out->f64 = fs; break;
  case LLKV_AGG_SUM:
    out->dtype = a->distinct_f64 ? LLKV_DT_FLOAT64 : LLKV_DT_INT64;
    out->is_null = m == 0;
    if (m) { if (a->distinct_f64) out->f64 = fs; else out->i64 = is; }
    break;
  default: /* AVG */
    out->dtype = LLKV_DT_FLOAT64;
    out->is_null = m == 0;
    if (m) out->f64 = (a->distinct_f64 ? fs : (double)is) / (double)m;
    break;
  }
  return LLKV_OK;
}

/* update(&RecordBatch) :759-1477 — strictly sequential, arrival order. */
static int32_t acc_update(acc *a, const arr *col, uint64_t num_rows) {
  if (a->distinct && a->distinct_dtype != LLKV_DT_INT64 && a->distinct_dtype != LLKV_DT_FLOAT64) {
    /* CountDistinctColumn :787-799; Sum / Total / AvgDistinctFloat64 over a non-float column :889-924 (a NEW key adds its
     * numeric image: Str → array_value_to_numeric, Bool → 1 / 0, Date → the day number); Sum / Total / AvgDistinctDecimal128
     * :943-967,1089-1112,1260-1284 (i128 checked_add of a new raw value) */
    if (col->dtype == LLKV_DT_NULL || a->distinct_rc) return LLKV_OK;
    for (uint64_t i = 0; i < col->n; ++i) {
      if (!col->valid[i]) continue;
      uint64_t at;
      if (col->dtype == LLKV_DT_UTF8) {
        if (!acc_insert_key(a, col->strings[i], (uint32_t)strlen(col->strings[i]), &at)) continue;
        a->f += rust_parse_f64_or_zero(col->strings[i]);
      } else if (col->dtype == LLKV_DT_BOOLEAN) {
        if (!acc_insert_key(a, (const char *)col->values + i, 1, &at)) continue;
        a->f += ((const uint8_t *)col->values)[i] ? 1.0 : 0.0;
      } else if (col->dtype == LLKV_DT_DATE32) {
        if (!acc_insert_key(a, (const char *)col->values + i * 4, 4, &at)) continue;
        a->f += (double)((const int32_t *)col->values)[i];
      } else if (col->dtype == LLKV_DT_DECIMAL128) {
        if (!acc_insert_key(a, (const char *)col->values + i * 16, 16, &at)) continue;
        __int128 v;
        memcpy(&v, (const char *)col->values + i * 16, 16);
        if (a->distinct_kind != LLKV_AGG_COUNT && __builtin_add_overflow(a->d, v, &a->d))
          a->distinct_rc = fail(LLKV_INVALID_ARGUMENT, a->distinct_kind == LLKV_AGG_TOTAL ? "Decimal128 total overflow" : "Decimal128 sum overflow");
      } else {
        return fail(LLKV_INVALID_ARGUMENT, "COUNT(DISTINCT) is not supported for column type %s", dtype_name(col->dtype));
      }
      if (a->distinct_rc) return a->distinct_rc;
    }
    return LLKV_OK;
  }
  if (a->distinct) {
    if (col->dtype == LLKV_DT_NULL) return LLKV_OK;
    for (uint64_t i = 0; i < col->n; ++i) {
      if (!col->valid[i]) continue;
      if (a->n_seen == a->cap_seen) { a->cap_seen = a->cap_seen ? a->cap_seen * 2 : 1024; a->seen = xrealloc(a->seen, a->cap_seen * 8); }
      memcpy(&a->seen[a->n_seen++], (const char *)col->values + i * 8, 8);
    }
    return LLKV_OK;
  }
  switch (a->kind) {
  case ACC_COUNT_STAR:
    if (__builtin_add_overflow(a->i, (int64_t)num_rows, &a->i)) return fail(LLKV_INVALID_ARGUMENT, "COUNT result exceeds i64 range");
    return LLKV_OK;
  case ACC_COUNT_COLUMN: {
    if (col->dtype == LLKV_DT_NULL) return LLKV_OK;
    int64_t nn = 0;
    for (uint64_t i = 0; i < col->n; ++i) nn += col->valid[i];
    if (__builtin_add_overflow(a->i, nn, &a->i)) return fail(LLKV_INVALID_ARGUMENT, "COUNT result exceeds i64 range");
    return LLKV_OK;
  }
  case ACC_COUNT_NULLS:
    for (uint64_t i = 0; i < col->n; ++i) a->i += col->valid[i];
    a->count += (int64_t)col->n;
    return LLKV_OK;
  case ACC_SUM_I64: case ACC_AVG_I64: case ACC_TOTAL_I64: case ACC_MIN_I64: case ACC_MAX_I64:
    if (col->dtype == LLKV_DT_NULL) return LLKV_OK;
    if (col->dtype != LLKV_DT_INT64) {
      const char *fn = a->kind == ACC_SUM_I64 ? "SUM" : a->kind == ACC_AVG_I64 ? "AVG" : a->kind == ACC_TOTAL_I64 ? "TOTAL" : a->kind == ACC_MIN_I64 ? "MIN" : "MAX";
      return fail(LLKV_INVALID_ARGUMENT, "%s aggregate expected an INT column in execution", fn);
    }
    for (uint64_t i = 0; i < col->n; ++i) {
      if (!col->valid[i]) continue;
      int64_t v = ((int64_t *)col->values)[i];
      switch (a->kind) {
      case ACC_SUM_I64: /* :801-830 checked_add */
        a->has = 1;
        if (__builtin_add_overflow(a->i, v, &a->i)) return fail(LLKV_INVALID_ARGUMENT, "integer overflow");
        break;
      case ACC_AVG_I64: /* :1114-1144 */
        if (__builtin_add_overflow(a->i, v, &a->i)) return fail(LLKV_INVALID_ARGUMENT, "AVG aggregate sum exceeds i64 range");
        a->count += 1;
        break;
      case ACC_TOTAL_I64: a->f += (double)v; break; /* :968-988 */
      case ACC_MIN_I64: a->i = a->has ? (v < a->i ? v : a->i) : v; a->has = 1; break;
      case ACC_MAX_I64: a->i = a->has ? (v > a->i ? v : a->i) : v; a->has = 1; break;
      }
    }
    return LLKV_OK;
  case ACC_SUM_DEC: case ACC_TOTAL_DEC: case ACC_AVG_DEC: case ACC_MIN_DEC: case ACC_MAX_DEC:
    if (col->dtype != LLKV_DT_DECIMAL128) return fail(LLKV_INVALID_ARGUMENT, "Expected Decimal128 array");
    a->precision = col->precision; a->scale = col->scale;
    for (uint64_t i = 0; i < col->n; ++i) {
      if (!col->valid[i]) continue;
      __int128 v;
      memcpy(&v, (const char *)col->values + i * 16, 16);
      switch (a->kind) {
      case ACC_SUM_DEC: case ACC_AVG_DEC:
        if (__builtin_add_overflow(a->d, v, &a->d)) return fail(LLKV_INVALID_ARGUMENT, "Decimal128 sum overflow");
        if (a->kind == ACC_AVG_DEC) a->count += 1;
        break;
      case ACC_TOTAL_DEC:
        if (__builtin_add_overflow(a->d, v, &a->d)) return fail(LLKV_INVALID_ARGUMENT, "Decimal128 total overflow");
        break;
      case ACC_MIN_DEC: a->d = a->has ? (v < a->d ? v : a->d) : v; a->has = 1; break;
      case ACC_MAX_DEC: a->d = a->has ? (v > a->d ? v : a->d) : v; a->has = 1; break;
      }
    }
    return LLKV_OK;
  default: /* Float64 accumulators with numeric coercion */
    if (col->dtype == LLKV_DT_NULL) return LLKV_OK;
    for (uint64_t i = 0; i < col->n; ++i) {
      if (!col->valid[i]) continue;
      double v = 0;
      int32_t rc = value_to_numeric(col, i, &v);
      if (rc) return rc;
      switch (a->kind) {
      case ACC_SUM_F64: a->f += v; a->has = 1; break;      /* :870-888 */
      case ACC_TOTAL_F64: a->f += v; break;                /* :1015-1034 */
      case ACC_AVG_F64: a->f += v; a->count += 1; break;   /* :1177-1199 */
      case ACC_MIN_F64: /* :1309-1331 partial_cmp strictly Less */
        if (!a->has) { a->f = v; a->has = 1; } else if (v < a->f) a->f = v;
        break;
      case ACC_MAX_F64: /* :1377-1399 */
        if (!a->has) { a->f = v; a->has = 1; } else if (v > a->f) a->f = v;
        break;
      }
    }
    return LLKV_OK;
  }
}

/* finalize :1488-1939 */
static void acc_finalize(const acc *a, llkv_value *out) {
  memset(out, 0, sizeof *out);
  switch (a->kind) {
  case ACC_COUNT_STAR: case ACC_COUNT_COLUMN: out->dtype = LLKV_DT_INT64; out->i64 = a->i; break;
  case ACC_COUNT_NULLS: out->dtype = LLKV_DT_INT64; out->i64 = a->count - a->i; break;
  case ACC_SUM_I64: out->dtype = LLKV_DT_INT64; out->is_null = !a->has; out->i64 = a->has ? a->i : 0; break;
  case ACC_SUM_F64: out->dtype = LLKV_DT_FLOAT64; out->is_null = !a->has; out->f64 = a->has ? a->f : 0; break;
  case ACC_TOTAL_I64: case ACC_TOTAL_F64: out->dtype = LLKV_DT_FLOAT64; out->f64 = a->f; break;
  case ACC_AVG_I64: out->dtype = LLKV_DT_FLOAT64; out->is_null = a->count <= 0; if (a->count > 0) out->f64 = (double)a->i / (double)a->count; break;
  case ACC_AVG_F64: out->dtype = LLKV_DT_FLOAT64; out->is_null = a->count <= 0; if (a->count > 0) out->f64 = a->f / (double)a->count; break;
  case ACC_MIN_I64: case ACC_MAX_I64: out->dtype = LLKV_DT_INT64; out->is_null = !a->has; out->i64 = a->has ? a->i : 0; break;
  case ACC_MIN_F64: case ACC_MAX_F64: out->dtype = LLKV_DT_FLOAT64; out->is_null = !a->has; out->f64 = a->has ? a->f : 0; break;
  case ACC_SUM_DEC: case ACC_TOTAL_DEC: case ACC_AVG_DEC: case ACC_MIN_DEC: case ACC_MAX_DEC: {
    __int128 v = a->d;
    out->dtype = LLKV_DT_DECIMAL128; out->precision = a->precision; out->scale = a->scale;
    if (a->kind == ACC_AVG_DEC) { /* :1720-1760 sum / count, rounded half away from zero */
      if (a->count > 0) {
        const __int128 n = a->count, rem = a->d % n;
        v = a->d / n;
        if ((rem < 0 ? -rem : rem) * 2 >= n) v += (a->d > 0) ? 1 : -1; /* sum.signum() vs count.signum() (> 0) */
      } else out->is_null = 1;
    } else if (a->kind == ACC_MIN_DEC || a->kind == ACC_MAX_DEC) out->is_null = !a->has;
    /* SUM / TOTAL: `vec![sum]` — 0, never NULL, even without rows (:1567-1582,1640-1655) */
    if (out->is_null) v = 0;
    out->i64 = (int64_t)(uint64_t)v;
    out->i64_hi = (int64_t)(v >> 64);
    break;
  }
  }
}

/* -------------------------------------------------- ungrouped aggregates */
typedef struct agg_ctx {
  acc *accs;
  const int32_t *proj_of_agg; /* -1 for COUNT(*) */
  uint32_t n_aggs;
} agg_ctx;

static void agg_window(const arr *cols, uint32_t n_cols, const uint64_t *row_ids, uint64_t n, void *user, int32_t *rc) {
  (void)row_ids; (void)n_cols;
  agg_ctx *c = user;
  /* for state in states { state.update(&batch) } llkv-executor/src/lib.rs:5633-5643 */
  for (uint32_t i = 0; i < c->n_aggs && *rc == LLKV_OK; ++i) {
    arr dummy;
    memset(&dummy, 0, sizeof dummy);
    const arr *col = c->proj_of_agg[i] >= 0 ? &cols[c->proj_of_agg[i]] : &dummy;
    *rc = acc_update(&c->accs[i], col, n);
  }
}

/* execute_aggregates llkv-executor/src/lib.rs:5357-5682 (bare columns) and
 * compute_aggregate_values :6087-6665 (aggregate arguments that are expressions
 * become computed scan projections, ensure_computed_projection :470-501).
 * Scan options: include_nulls = true (:5569-5576). */
int32_t orc_aggregate(const orc_table *t, const llkv_filter *filters, uint32_t n_filters,
                      const llkv_eval_op *ops, uint32_t n_ops, const llkv_aggregate_spec *aggs,
                      uint32_t n_aggs, llkv_value *out_values) {
  if (n_aggs == 0) return fail(LLKV_INVALID_ARGUMENT, "aggregate query requires at least one aggregate expression");
  acc *accs = xcalloc(n_aggs, sizeof(acc));
  int32_t *proj_of = xcalloc(n_aggs, sizeof(int32_t));
  proj_plan *projs = xcalloc(n_aggs + 1, sizeof(proj_plan));
  uint32_t n_projs = 0;
  int32_t rc = LLKV_OK;
  for (uint32_t i = 0; i < n_aggs && rc == LLKV_OK; ++i) {
    if (aggs[i].kind == LLKV_AGG_COUNT_STAR) { proj_of[i] = -1; rc = acc_new(aggs[i].kind, LLKV_DT_NULL, &accs[i]); continue; }
    int has_div;
    int32_t dt = infer_expr_type(t, aggs[i].expr, aggs[i].expr_len, &has_div, &rc);
    if (rc) break;
    if (!is_simple_column(aggs[i].expr, aggs[i].expr_len) && !has_div && (dt == LLKV_DT_INT32 || dt == LLKV_DT_UINT32)) {
      rc = fail(LLKV_UNSUPPORTED, "aggregate over a 32-bit-only integer expression (the reference has no Int32 accumulator)"); /* (its array would be Int32: not restated) */
      break;
    }
    if (!is_simple_column(aggs[i].expr, aggs[i].expr_len) && dt != LLKV_DT_FLOAT64) dt = LLKV_DT_INT64;
    const int is_distinct = aggs[i].distinct && aggs[i].kind != LLKV_AGG_MIN && aggs[i].kind != LLKV_AGG_MAX; /* MIN / MAX have no DISTINCT form */
    rc = is_distinct ? acc_new_distinct(aggs[i].kind, dt, &accs[i]) : acc_new(aggs[i].kind, dt, &accs[i]);
    if (rc) break;
    if (dt == LLKV_DT_DECIMAL128) { /* the spec's DataType::Decimal128(precision, scale) */
      const orc_column *dc = find_col(t, aggs[i].expr[0].field_id);
      accs[i].precision = dc->precision; accs[i].scale = dc->scale;
    }
    proj_of[i] = (int32_t)n_projs;
    projs[n_projs].computed = !is_simple_column(aggs[i].expr, aggs[i].expr_len);
    projs[n_projs].field_id = aggs[i].expr[0].field_id;
    projs[n_projs].expr = aggs[i].expr;
    projs[n_projs].expr_len = aggs[i].expr_len;
    ++n_projs;
  }
  if (rc == LLKV_OK) {
    if (n_projs == 0) { /* COUNT(*) only: the scan still needs a column to drive row ids */
      uint64_t *ids = NULL, n_ids = 0;
      rc = orc_filter_row_ids(t, filters, n_filters, ops, n_ops, &ids, &n_ids);
      if (rc == LLKV_OK) for (uint32_t i = 0; i < n_aggs && rc == LLKV_OK; ++i) { arr d; memset(&d, 0, sizeof d); rc = acc_update(&accs[i], &d, n_ids); }
      free(ids);
    } else {
      agg_ctx c = {accs, proj_of, n_aggs};
      rc = scan_core(t, projs, n_projs, filters, n_filters, ops, n_ops, /*include_nulls=*/1, agg_window, &c);
      if (rc == LLKV_NOT_FOUND) rc = LLKV_OK; /* NotFound = empty table, :5646-5650 */
    }
  }
  for (uint32_t i = 0; i < n_aggs && rc == LLKV_OK; ++i) {
    if (accs[i].distinct) rc = acc_finalize_distinct(&accs[i], &out_values[i]);
    else acc_finalize(&accs[i], &out_values[i]);
  }
  for (uint32_t i = 0; i < n_aggs; ++i) acc_free_distinct(&accs[i]);
  free(accs); free(proj_of); free(projs);
  return rc;
}

/* ------------------------------------------------------------- GROUP BY */
/* PlanValue (llkv-plan/src/plans.rs:1038-1061) restricted to this path. */
enum { PV_NULL, PV_INT, PV_FLOAT, PV_STR, PV_DEC };
typedef struct pval { int tag; int64_t i; double f; const char *s; i128 d; int32_t scale; /* PV_DEC: DecimalValue{value, scale} */ } pval;

/* ---- DecimalValue and its exact arithmetic: llkv-types/src/decimal.rs:58-110,218-262 and
 * llkv-compute/src/scalar/decimal.rs:1-234.  The reference computes in arrow_buffer::i256 and narrows with to_i128;
 * restated here over a sign + 256-bit magnitude (four 64-bit limbs, least significant first). */
#define DEC_MAX_PRECISION 38 /* DECIMAL128_MAX_PRECISION */
typedef struct u256 { uint64_t w[4]; } u256;

static u256 u256_from_u128(u128 v) { u256 r = {{(uint64_t)v, (uint64_t)(v >> 64), 0, 0}}; return r; }
static int u256_is_zero(const u256 *a) { return !(a->w[0] | a->w[1] | a->w[2] | a->w[3]); }
static int u256_cmp(const u256 *a, const u256 *b) {
  for (int k = 3; k >= 0; --k) if (a->w[k] != b->w[k]) return a->w[k] < b->w[k] ? -1 : 1;
  return 0;
}
/* a * 10, 0 when the product leaves the i256 magnitude range (≥ 2^255): checked_mul → DecimalError::Overflow */
static int u256_mul10(u256 *a) {
  u128 carry = 0;
  for (int k = 0; k < 4; ++k) { u128 t = (u128)a->w[k] * 10u + carry; a->w[k] = (uint64_t)t; carry = t >> 64; }
  return carry == 0 && !(a->w[3] >> 63);
}
static void u256_shl1(u256 *a) { for (int k = 3; k > 0; --k) a->w[k] = (a->w[k] << 1) | (a->w[k - 1] >> 63); a->w[0] <<= 1; }
static void u256_sub(u256 *a, const u256 *b) {
  unsigned borrow = 0;
  for (int k = 0; k < 4; ++k) { u128 t = (u128)a->w[k] - b->w[k] - borrow; a->w[k] = (uint64_t)t; borrow = (unsigned)((t >> 64) & 1); }
}
/* magnitudes: q = n / d, r = n % d (d != 0): shift-subtract, one bit at a time */
static void u256_divmod(const u256 *n, const u256 *d, u256 *q, u256 *r) {
  u256 rem = {{0, 0, 0, 0}}, quo = {{0, 0, 0, 0}};
  for (int bit = 255; bit >= 0; --bit) {
    u256_shl1(&rem);
    rem.w[0] |= (n->w[bit >> 6] >> (bit & 63)) & 1u;
    u256_shl1(&quo);
    if (u256_cmp(&rem, d) >= 0) { u256_sub(&rem, d); quo.w[0] |= 1u; }
  }
  *q = quo; *r = rem;
}
static u128 i128_mag(i128 v) { return v < 0 ? (u128)0 - (u128)v : (u128)v; }
/* digit_count_i256 llkv-types/src/decimal.rs:218-231: 1 for zero */
static int dec_digits(i128 v) {
  u128 m = i128_mag(v);
  int n = 0;
  do { m /= 10; ++n; } while (m);
  return n;
}
static int dec_scale_ok(int s) { return s >= -DEC_MAX_PRECISION && s <= DEC_MAX_PRECISION; } /* scale_within_bounds :258-261 */
/* DecimalValue::new :66-76: scale bounds, then at most 38 digits */
static int32_t dec_new(i128 value, int scale, const char *what, pval *out) {
  if (!dec_scale_ok(scale)) return fail(LLKV_INVALID_ARGUMENT, "%s: decimal scale %d outside supported range", what, scale);
  if (dec_digits(value) > DEC_MAX_PRECISION) return fail(LLKV_INVALID_ARGUMENT, "%s: decimal value with scale %d exceeds maximum precision", what, scale);
  memset(out, 0, sizeof *out);
  out->tag = PV_DEC; out->d = value; out->scale = scale;
  return LLKV_OK;
}
/* rescale :29-64 towards a LARGER scale (add / sub align to the maximum): value · 10^diff in i256 (checked), to_i128,
 * DecimalValue::new.  A product beyond i128 is Overflow at the latest in to_i128, so i128 steps decide the same. */
static int32_t dec_rescale_up(pval v, int target, const char *what, pval *out) {
  if (!dec_scale_ok(target)) return fail(LLKV_INVALID_ARGUMENT, "%s: decimal scale %d outside supported range", what, target);
  i128 x = v.d;
  for (int k = v.scale; k < target; ++k)
    if (__builtin_mul_overflow(x, (i128)10, &x)) return fail(LLKV_INVALID_ARGUMENT, "%s: decimal arithmetic overflow", what);
  return dec_new(x, target, what, out);
}
/* add / sub :128-151, mul :153-166, div :168-234 (target scale = the left operand's, llkv-executor/src/lib.rs:7303-7316) */
static int32_t dec_binary(pval l, pval r, int32_t op, const char *what, pval *out) {
  if (op == LLKV_BIN_ADD || op == LLKV_BIN_SUB) {
    const int target = l.scale > r.scale ? l.scale : r.scale;
    pval a, b;
    int32_t rc;
    if ((rc = dec_rescale_up(l, target, what, &a)) || (rc = dec_rescale_up(r, target, what, &b))) return rc;
    i128 z;
    if (op == LLKV_BIN_ADD ? __builtin_add_overflow(a.d, b.d, &z) : __builtin_sub_overflow(a.d, b.d, &z))
      return fail(LLKV_INVALID_ARGUMENT, "%s: decimal arithmetic overflow", what); /* the i256 sum does not fit i128 */
    return dec_new(z, target, what, out);
  }
  if (op == LLKV_BIN_MUL) {
    const int scale = l.scale + r.scale;
    if (!dec_scale_ok(scale)) return fail(LLKV_INVALID_ARGUMENT, "%s: decimal scale %d outside supported range", what, scale);
    i128 z;
    if (__builtin_mul_overflow(l.d, r.d, &z)) return fail(LLKV_INVALID_ARGUMENT, "%s: decimal arithmetic overflow", what);
    return dec_new(z, scale, what, out);
  }
  /* div: numerator · 10^(target + rhs.scale − lhs.scale) = 10^rhs.scale here, exact division by a negative power,
   * truncating quotient, then "half away from zero" as the reference writes it: half = denominator / 2 truncated, so an
   * odd denominator rounds up from (|d| − 1) / 2, and the direction follows the signs of the TRUNCATED quotient and the
   * denominator (a quotient of 0 counts as positive) */
  const int target = l.scale;
  const int adj = target + r.scale - l.scale;
  u256 num = u256_from_u128(i128_mag(l.d));
  const int num_neg = l.d < 0, den_neg = r.d < 0;
  if (adj > 0) {
    if (adj > 2 * DEC_MAX_PRECISION) return fail(LLKV_INVALID_ARGUMENT, "%s: decimal scale %d outside supported range", what, adj);
    for (int k = 0; k < adj; ++k) if (!u256_mul10(&num)) return fail(LLKV_INVALID_ARGUMENT, "%s: decimal arithmetic overflow", what);
  } else if (adj < 0) {
    u256 f = u256_from_u128(1), q, rem;
    for (int k = 0; k < -adj; ++k) u256_mul10(&f);
    u256_divmod(&num, &f, &q, &rem);
    if (!u256_is_zero(&rem)) return fail(LLKV_INVALID_ARGUMENT, "%s: cannot rescale decimal from scale %d to %d without losing precision", what, l.scale, l.scale + adj);
    num = q;
  }
  const u256 den = u256_from_u128(i128_mag(r.d));
  u256 q, rem;
  u256_divmod(&num, &den, &q, &rem);
  int q_neg = (num_neg != den_neg) && !u256_is_zero(&q);
  if (!u256_is_zero(&rem)) {
    const u256 half = u256_from_u128(i128_mag(r.d) / 2);
    if (u256_cmp(&rem, &half) >= 0) {
      const int up = (!q_neg) == (!den_neg); /* (quotient >= 0) == (denominator >= 0) → + 1, else − 1 */
      /* signed q ± 1 on the magnitude */
      if (up == !q_neg) { /* moving away from zero (or from 0 upwards) */
        unsigned carry = 1;
        for (int k = 0; k < 4 && carry; ++k) { q.w[k] += 1; carry = q.w[k] == 0; }
      } else if (u256_is_zero(&q)) { /* 0 − 1 */
        q.w[0] = 1; q_neg = 1;
      } else { /* towards zero */
        const u256 one = u256_from_u128(1);
        u256_sub(&q, &one);
        if (u256_is_zero(&q)) q_neg = 0;
      }
    }
  }
  /* to_i128 */
  if (q.w[2] | q.w[3]) return fail(LLKV_INVALID_ARGUMENT, "%s: decimal arithmetic overflow", what);
  const u128 mag = ((u128)q.w[1] << 64) | q.w[0];
  if (mag > ((u128)1 << 127) || (!q_neg && mag == ((u128)1 << 127))) return fail(LLKV_INVALID_ARGUMENT, "%s: decimal arithmetic overflow", what);
  return dec_new(q_neg ? (i128)((u128)0 - mag) : (i128)mag, target, what, out);
}

static pval pv_from_arr(const arr *a, uint64_t i) {
  pval v = {PV_NULL, 0, 0, NULL, 0, 0};
  if (!a->valid[i]) return v;
  switch (a->dtype) {
  case LLKV_DT_INT64: v.tag = PV_INT; v.i = ((int64_t *)a->values)[i]; break;
  case LLKV_DT_INT32: case LLKV_DT_DATE32: v.tag = PV_INT; v.i = ((int32_t *)a->values)[i]; break;
  case LLKV_DT_UINT32: v.tag = PV_INT; v.i = ((uint32_t *)a->values)[i]; break;
  case LLKV_DT_UINT64: v.tag = PV_INT; v.i = (int64_t)((uint64_t *)a->values)[i]; break;
  case LLKV_DT_FLOAT64: v.tag = PV_FLOAT; v.f = ((double *)a->values)[i]; break;
  case LLKV_DT_FLOAT32: v.tag = PV_FLOAT; v.f = ((float *)a->values)[i]; break;
  case LLKV_DT_UTF8: v.tag = PV_STR; v.s = a->strings[i]; break;
  case LLKV_DT_DECIMAL128: /* plan_value_from_array llkv-plan/src/plans.rs:1160-1174: DecimalValue::new(raw, scale) — checked by pv_eval */
    v.tag = PV_DEC; memcpy(&v.d, (const char *)a->values + i * 16, 16); v.scale = a->scale; break;
  default: break;
  }
  return v;
}

/* Rust `f64 as i64`: saturating, NaN → 0. */
static int64_t f64_as_i64(double x) {
  if (x != x) return 0;
  if (x >= 9223372036854775808.0) return INT64_MAX;
  if (x <= -9223372036854775808.0) return INT64_MIN;
  return (int64_t)x;
}

/* evaluate_expr_with_plan_value_aggregates_and_row, Binary arm:
 * llkv-executor/src/lib.rs:7193-7389.  Int∘Int for + - * is computed in f64 and cast
 * back to i64 (:7338-7389); Int/Int truncates, i64::MIN / -1 → Float (:7213-7227);
 * x/0, x%0 → NULL; NULL propagates. */
static int32_t pv_binary(pval l, pval r, int32_t op, pval *out) {
  pval z = {PV_NULL, 0, 0, NULL, 0, 0};
  if (l.tag == PV_NULL || r.tag == PV_NULL) { *out = z; return LLKV_OK; }
  if (op == LLKV_BIN_DIV && l.tag == PV_INT && r.tag == PV_INT) {
    if (r.i == 0) { *out = z; return LLKV_OK; }
    if (l.i == INT64_MIN && r.i == -1) { z.tag = PV_FLOAT; z.f = (double)l.i / (double)r.i; *out = z; return LLKV_OK; }
    z.tag = PV_INT; z.i = l.i / r.i; *out = z; return LLKV_OK;
  }
  /* a Decimal operand: exact decimal arithmetic, the other side converted (:7229-7330) — Integer through
   * DecimalValue::from_i64 (scale 0), Float is an error, Modulo is an error, a zero divisor gives NULL */
  if (l.tag == PV_DEC || r.tag == PV_DEC) {
    pval side[2] = {l, r};
    for (int k = 0; k < 2; ++k) {
      if (side[k].tag == PV_INT) { side[k].tag = PV_DEC; side[k].d = side[k].i; side[k].scale = 0; }
      else if (side[k].tag == PV_FLOAT) return fail(LLKV_INVALID_ARGUMENT, "Cannot perform exact decimal arithmetic with Float operands");
      else if (side[k].tag != PV_DEC) return fail(LLKV_INVALID_ARGUMENT, "Non-numeric value in binary operation");
    }
    switch (op) {
    case LLKV_BIN_ADD: return dec_binary(side[0], side[1], op, "Decimal addition overflow", out);
    case LLKV_BIN_SUB: return dec_binary(side[0], side[1], op, "Decimal subtraction overflow", out);
    case LLKV_BIN_MUL: return dec_binary(side[0], side[1], op, "Decimal multiplication overflow", out);
    case LLKV_BIN_DIV:
      if (side[1].d == 0) { *out = z; return LLKV_OK; }
      return dec_binary(side[0], side[1], op, "Decimal division error", out);
    case LLKV_BIN_MOD: return fail(LLKV_INVALID_ARGUMENT, "Modulo not supported for Decimal types");
    default: return fail(LLKV_INTERNAL, "bad binary op");
    }
  }
  if (l.tag == PV_STR || r.tag == PV_STR) return fail(LLKV_INVALID_ARGUMENT, "Non-numeric value in binary operation");
  int lf = l.tag == PV_FLOAT, rf = r.tag == PV_FLOAT;
  double a = lf ? l.f : (double)l.i, b = rf ? r.f : (double)r.i, res = 0;
  switch (op) {
  case LLKV_BIN_ADD: res = a + b; break;
  case LLKV_BIN_SUB: res = a - b; break;
  case LLKV_BIN_MUL: res = a * b; break;
  case LLKV_BIN_DIV: if (b == 0.0) { *out = z; return LLKV_OK; } res = a / b; break;
  case LLKV_BIN_MOD: if (b == 0.0) { *out = z; return LLKV_OK; } res = fmod(a, b); break;
  default: return fail(LLKV_INTERNAL, "bad binary op");
  }
  if (op == LLKV_BIN_DIV || lf || rf) { z.tag = PV_FLOAT; z.f = res; }
  else { z.tag = PV_INT; z.i = f64_as_i64(res); }
  *out = z;
  return LLKV_OK;
}

static int32_t pv_eval(const llkv_expr_token *e, uint32_t n, const gathered *g, uint32_t n_g, uint64_t row, pval *out) {
  pval st[64];
  uint32_t sp = 0;
  for (uint32_t i = 0; i < n; ++i) {
    switch (e[i].kind) {
    case LLKV_TOK_COLUMN: {
      const arr *a = find_gathered(g, n_g, e[i].field_id);
      if (!a) return fail(LLKV_INVALID_ARGUMENT, "column not found for aggregate");
      pval cv = pv_from_arr(a, row);
      if (cv.tag == PV_DEC) { /* DecimalValue::new over the cell: more than 38 digits fail the conversion */
        int32_t rc = dec_new(cv.d, cv.scale, "failed to convert Decimal128 value", &cv);
        if (rc) return rc;
      }
      st[sp++] = cv;
      break;
    }
    case LLKV_TOK_LITERAL: {
      pval v = {PV_NULL, 0, 0, NULL, 0, 0};
      const llkv_literal *l = &e[i].literal;
      if (l->tag == LLKV_LIT_INT128) { v.tag = PV_INT; v.i = (int64_t)lit_i128(l); } /* `*v as i64` llkv-executor/src/lib.rs:7019 */
      else if (l->tag == LLKV_LIT_FLOAT64) { v.tag = PV_FLOAT; v.f = l->f64; }
      else if (l->tag == LLKV_LIT_DECIMAL128) { v.tag = PV_DEC; v.d = lit_i128(l); v.scale = l->scale; } /* Literal::Decimal128(DecimalValue) :7021 */
      else if (l->tag != LLKV_LIT_NULL) return fail(LLKV_UNSUPPORTED, "literal kind in aggregate expression");
      st[sp++] = v;
      break;
    }
    case LLKV_TOK_BINARY: {
      pval r = st[--sp], l = st[--sp], z;
      int32_t rc = pv_binary(l, r, e[i].binop, &z);
      if (rc) return rc;
      st[sp++] = z;
      break;
    }
    }
  }
  *out = st[0];
  return LLKV_OK;
}

/* GroupKeyValue llkv-executor/src/lib.rs:99-106; group_key_value :9362-9456:
 * every integer width and Date32 collapse to Int, Utf8 → owned String, NULL is its
 * own key; Float keys are rejected. */
typedef struct gkey { int tag; int64_t i; char *s; } gkey; /* tag: PV_NULL / PV_INT / PV_STR */

struct orc_groups {
  uint32_t n_groups, n_keys, n_aggs;
  gkey *keys;        /* [n_groups][n_keys] */
  llkv_value *vals;  /* [n_groups][n_aggs] */
};

uint32_t orc_groups_len(const orc_groups *g) { return g->n_groups; }
int32_t orc_groups_key(const orc_groups *g, uint32_t group, uint32_t key, llkv_value *out) {
  if (group >= g->n_groups || key >= g->n_keys) return fail(LLKV_INVALID_ARGUMENT, "index out of range");
  const gkey *k = &g->keys[(size_t)group * g->n_keys + key];
  memset(out, 0, sizeof *out);
  out->is_null = k->tag == PV_NULL;
  out->dtype = k->tag == PV_STR ? LLKV_DT_UTF8 : LLKV_DT_INT64;
  out->i64 = k->i;
  out->str = k->s;
  return LLKV_OK;
}
int32_t orc_groups_value(const orc_groups *g, uint32_t group, uint32_t agg, llkv_value *out) {
  if (group >= g->n_groups || agg >= g->n_aggs) return fail(LLKV_INVALID_ARGUMENT, "index out of range");
  *out = g->vals[(size_t)group * g->n_aggs + agg];
  return LLKV_OK;
}
void orc_groups_free(orc_groups *g) {
  if (!g) return;
  for (size_t i = 0; i < (size_t)g->n_groups * g->n_keys; ++i) free(g->keys[i].s);
  free(g->keys); free(g->vals); free(g);
}

static int gkey_eq(const gkey *a, const gkey *b, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) {
    if (a[i].tag != b[i].tag) return 0;
    if (a[i].tag == PV_INT && a[i].i != b[i].i) return 0;
    if (a[i].tag == PV_STR && strcmp(a[i].s, b[i].s)) return 0;
  }
  return 1;
}
static uint64_t gkey_hash(const gkey *a, uint32_t n) {
  uint64_t h = 0xcbf29ce484222325ULL;
  for (uint32_t i = 0; i < n; ++i) {
    h = (h ^ (uint64_t)a[i].tag) * 0x100000001b3ULL;
    if (a[i].tag == PV_INT) h = (h ^ (uint64_t)a[i].i) * 0x100000001b3ULL;
    else if (a[i].tag == PV_STR) for (const char *p = a[i].s; *p; ++p) h = (h ^ (uint8_t)*p) * 0x100000001b3ULL;
  }
  return h;
}
/* lexsort ascending, NULLs first (arrow SortOptions default of the reference's
 * ORDER BY ASC, llkv-executor/src/lib.rs:13762-13868) */
static int gkey_cmp(const gkey *a, const gkey *b, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) {
    if (a[i].tag == PV_NULL || b[i].tag == PV_NULL) { if (a[i].tag != b[i].tag) return a[i].tag == PV_NULL ? -1 : 1; continue; }
    if (a[i].tag == PV_INT) { if (a[i].i != b[i].i) return a[i].i < b[i].i ? -1 : 1; }
    else { int c = strcmp(a[i].s, b[i].s); if (c) return c; }
  }
  return 0;
}

typedef struct gb_rows {
  gathered *cols; /* every needed field, all filtered rows materialised (:4512-4524) */
  uint32_t n_cols;
  uint64_t n, cap;
} gb_rows;

static void gb_window(const arr *cols, uint32_t n_cols, const uint64_t *row_ids, uint64_t n, void *user, int32_t *rc) {
  (void)row_ids; (void)rc;
  gb_rows *m = user;
  if (m->n + n > m->cap) {
    m->cap = (m->n + n) * 2;
    for (uint32_t j = 0; j < n_cols; ++j) {
      arr *a = &m->cols[j].a;
      size_t w = dtype_width(a->dtype);
      if (w) a->values = xrealloc(a->values, m->cap * w);
      a->valid = xrealloc(a->valid, m->cap);
      if (a->dtype == LLKV_DT_UTF8) a->strings = xrealloc(a->strings, m->cap * sizeof(char *));
    }
  }
  for (uint32_t j = 0; j < n_cols; ++j) {
    arr *a = &m->cols[j].a;
    size_t w = dtype_width(a->dtype);
    if (w) memcpy((char *)a->values + m->n * w, cols[j].values, n * w);
    memcpy(a->valid + m->n, cols[j].valid, n);
    if (a->dtype == LLKV_DT_UTF8) for (uint64_t i = 0; i < n; ++i) a->strings[m->n + i] = strdup(cols[j].strings[i]);
    a->n = m->n + n;
  }
  m->n += n;
}

int32_t orc_groupby(const orc_table *t, const llkv_filter *filters, uint32_t n_filters,
                    const llkv_eval_op *ops, uint32_t n_ops, const uint32_t *key_fields, uint32_t n_keys,
                    const llkv_aggregate_spec *aggs, uint32_t n_aggs, int32_t order_by_keys, orc_groups **out) {
  if (n_keys == 0) return fail(LLKV_INVALID_ARGUMENT, "GROUP BY requires at least one key");
  /* fields to materialise: keys + aggregate inputs (the reference materialises every
   * table column, :4454; the unused ones do not influence the result) */
  uint32_t fields[128], n_fields = 0;
#define ADD_FIELD(fid) do { uint32_t _j = 0; while (_j < n_fields && fields[_j] != (fid)) ++_j; if (_j == n_fields) fields[n_fields++] = (fid); } while (0)
  for (uint32_t k = 0; k < n_keys; ++k) ADD_FIELD(key_fields[k]);
  for (uint32_t a = 0; a < n_aggs; ++a) for (uint32_t k = 0; k < aggs[a].expr_len; ++k) if (aggs[a].expr[k].kind == LLKV_TOK_COLUMN) ADD_FIELD(aggs[a].expr[k].field_id);
#undef ADD_FIELD
  proj_plan *projs = xcalloc(n_fields, sizeof *projs);
  gb_rows m = {xcalloc(n_fields, sizeof(gathered)), n_fields, 0, 0};
  for (uint32_t j = 0; j < n_fields; ++j) {
    const orc_column *c = find_col(t, fields[j]);
    if (!c) { free(projs); free(m.cols); return fail(LLKV_INVALID_ARGUMENT, "column %u not found in GROUP BY input", fields[j]); }
    projs[j].field_id = fields[j];
    m.cols[j].field_id = fields[j];
    m.cols[j].a.dtype = c->dtype; m.cols[j].a.precision = c->precision; m.cols[j].a.scale = c->scale;
  }
  int32_t rc = scan_core(t, projs, n_fields, filters, n_filters, ops, n_ops, /*include_nulls=*/1, gb_window, &m);
  free(projs);

  /* pass 1: per-row key → group index in first-appearance order (:5065-5089) */
  uint32_t n_groups = 0, cap_groups = 16;
  gkey *gkeys = xcalloc((size_t)cap_groups * n_keys, sizeof(gkey));
  uint32_t *group_of_row = xmalloc((m.n ? m.n : 1) * sizeof(uint32_t));
  uint64_t tab_cap = 1024;
  int64_t *tab = xmalloc(tab_cap * sizeof(int64_t));
  memset(tab, 0xff, tab_cap * sizeof(int64_t));
  gkey *cur = xcalloc(n_keys, sizeof(gkey));
  for (uint64_t r = 0; r < m.n && rc == LLKV_OK; ++r) {
    for (uint32_t k = 0; k < n_keys; ++k) {
      const arr *a = find_gathered(m.cols, n_fields, key_fields[k]);
      pval v = pv_from_arr(a, r);
      if (v.tag == PV_FLOAT || a->dtype == LLKV_DT_DECIMAL128) { rc = fail(LLKV_INVALID_ARGUMENT, "GROUP BY does not support column type %s", dtype_name(a->dtype)); break; }
      cur[k].tag = v.tag; cur[k].i = v.i; cur[k].s = (char *)v.s;
    }
    if (rc) break;
    uint64_t h = gkey_hash(cur, n_keys) & (tab_cap - 1);
    for (;;) {
      if (tab[h] < 0) {
        if (n_groups == cap_groups) { cap_groups *= 2; gkeys = xrealloc(gkeys, (size_t)cap_groups * n_keys * sizeof(gkey)); }
        for (uint32_t k = 0; k < n_keys; ++k) { gkey *d = &gkeys[(size_t)n_groups * n_keys + k]; *d = cur[k]; if (d->tag == PV_STR) d->s = strdup(cur[k].s); }
        tab[h] = n_groups;
        group_of_row[r] = n_groups++;
        if ((uint64_t)n_groups * 2 > tab_cap) { /* rehash */
          tab_cap *= 4; tab = xrealloc(tab, tab_cap * sizeof(int64_t)); memset(tab, 0xff, tab_cap * sizeof(int64_t));
          for (uint32_t g = 0; g < n_groups; ++g) { uint64_t hh = gkey_hash(&gkeys[(size_t)g * n_keys], n_keys) & (tab_cap - 1); while (tab[hh] >= 0) hh = (hh + 1) & (tab_cap - 1); tab[hh] = g; }
        }
        break;
      }
      if (gkey_eq(&gkeys[(size_t)tab[h] * n_keys], cur, n_keys)) { group_of_row[r] = (uint32_t)tab[h]; break; }
      h = (h + 1) & (tab_cap - 1);
    }
  }
  free(cur); free(tab);

  /* pass 2: per group, rows in scan order; per aggregate a fresh AggregateState fed
   * either the bare column or the per-row PlanValue results (:5101-5252) */
  llkv_value *vals = xcalloc((size_t)(n_groups ? n_groups : 1) * (n_aggs ? n_aggs : 1), sizeof(llkv_value));
  if (rc == LLKV_OK) {
    /* bucket rows by group, preserving order */
    uint64_t *start = xcalloc((size_t)n_groups + 1, sizeof(uint64_t));
    for (uint64_t r = 0; r < m.n; ++r) start[group_of_row[r] + 1]++;
    for (uint32_t g = 0; g < n_groups; ++g) start[g + 1] += start[g];
    uint64_t *fill = xmalloc(((size_t)n_groups + 1) * sizeof(uint64_t));
    memcpy(fill, start, ((size_t)n_groups + 1) * sizeof(uint64_t));
    uint64_t *order = xmalloc((m.n ? m.n : 1) * sizeof(uint64_t));
    for (uint64_t r = 0; r < m.n; ++r) order[fill[group_of_row[r]]++] = r;
    for (uint32_t g = 0; g < n_groups && rc == LLKV_OK; ++g) {
      uint64_t gn = start[g + 1] - start[g];
      const uint64_t *rows = order + start[g];
      for (uint32_t a = 0; a < n_aggs && rc == LLKV_OK; ++a) {
        acc st;
        arr col;
        memset(&col, 0, sizeof col);
        /* every group runs its own accumulator over its rows in scan order (:5222-5247): DISTINCT forms included */
        const int is_distinct = aggs[a].distinct && aggs[a].kind != LLKV_AGG_MIN && aggs[a].kind != LLKV_AGG_MAX && aggs[a].kind != LLKV_AGG_COUNT_STAR;
        if (aggs[a].kind == LLKV_AGG_COUNT_STAR) {
          rc = acc_new(aggs[a].kind, LLKV_DT_NULL, &st);
          if (rc == LLKV_OK) rc = acc_update(&st, &col, gn);
        } else if (is_simple_column(aggs[a].expr, aggs[a].expr_len)) {
          const arr *src = find_gathered(m.cols, n_fields, aggs[a].expr[0].field_id);
          rc = is_distinct ? acc_new_distinct(aggs[a].kind, src->dtype, &st) : acc_new(aggs[a].kind, src->dtype, &st);
          if (rc) break;
          st.precision = src->precision; st.scale = src->scale;
          /* arrow `take` of the group's rows (:5131-5146) */
          col.dtype = src->dtype; col.precision = src->precision; col.scale = src->scale; col.n = gn; col.valid = xmalloc(gn ? gn : 1);
          size_t w = dtype_width(src->dtype);
          col.values = xmalloc((gn ? gn : 1) * (w ? w : 1));
          if (src->dtype == LLKV_DT_UTF8) col.strings = xcalloc(gn ? gn : 1, sizeof(char *));
          for (uint64_t i = 0; i < gn; ++i) {
            col.valid[i] = src->valid[rows[i]];
            if (w) memcpy((char *)col.values + i * w, (char *)src->values + rows[i] * w, w);
            if (col.strings) col.strings[i] = strdup(src->strings[rows[i]]);
          }
          rc = acc_update(&st, &col, gn);
          arr_free(&col);
        } else {
          /* row-by-row PlanValue interpreter (:5186-5199), temp column typed from the
           * first non-NULL value (plan_values_to_arrow_array :298-406) */
          pval *pv = xmalloc((gn ? gn : 1) * sizeof(pval));
          int first = PV_NULL;
          uint64_t first_at = 0;
          for (uint64_t i = 0; i < gn && rc == LLKV_OK; ++i) {
            rc = pv_eval(aggs[a].expr, aggs[a].expr_len, m.cols, n_fields, rows[i], &pv[i]);
            if (first == PV_NULL && rc == LLKV_OK) { first = pv[i].tag; first_at = i; }
          }
          if (rc == LLKV_OK) {
            col.n = gn; col.valid = xmalloc(gn ? gn : 1); col.values = xmalloc((gn ? gn : 1) * 16);
            col.dtype = first == PV_FLOAT ? LLKV_DT_FLOAT64 : first == PV_INT ? LLKV_DT_INT64 : first == PV_DEC ? LLKV_DT_DECIMAL128 : LLKV_DT_NULL;
            if (first == PV_DEC) {
              /* Decimal128Array::builder(..).with_precision_and_scale(d.precision(), d.scale()) of the FIRST non-NULL value
               * (:314-324): its digit count is the column's precision, and arrow-rs 57.1 validate_decimal_precision_and_scale
               * refuses a positive scale above the precision — a first value below 10^(scale−1) in magnitude fails the query */
              col.precision = dec_digits(pv[first_at].d); col.scale = pv[first_at].scale;
              if (col.scale > 0 && col.scale > col.precision)
                rc = fail(LLKV_INVALID_ARGUMENT, "invalid Decimal128 precision/scale: scale %d is greater than precision %d", col.scale, col.precision);
            }
            for (uint64_t i = 0; i < gn && rc == LLKV_OK; ++i) {
              col.valid[i] = pv[i].tag != PV_NULL;
              if (col.dtype == LLKV_DT_DECIMAL128) { /* append_value(raw) — the later values are not checked against the precision */
                if (pv[i].tag != PV_DEC && pv[i].tag != PV_NULL) rc = fail(LLKV_INVALID_ARGUMENT, "expected DECIMAL plan value");
                else { const i128 raw = pv[i].tag == PV_DEC ? pv[i].d : 0; memcpy((char *)col.values + i * 16, &raw, 16); }
              }
              else if (pv[i].tag == PV_DEC) rc = fail(LLKV_INVALID_ARGUMENT, "expected %s plan value, found Decimal", col.dtype == LLKV_DT_FLOAT64 ? "FLOAT" : "INTEGER");
              else if (col.dtype == LLKV_DT_FLOAT64) ((double *)col.values)[i] = pv[i].tag == PV_FLOAT ? pv[i].f : (double)pv[i].i;
              else if (pv[i].tag == PV_FLOAT) rc = fail(LLKV_INVALID_ARGUMENT, "expected INTEGER plan value, found Float");
              else ((int64_t *)col.values)[i] = pv[i].i;
            }
            if (col.dtype == LLKV_DT_NULL) { col.dtype = LLKV_DT_INT64; } /* new_null_array(Int64) */
            if (rc == LLKV_OK) rc = is_distinct ? acc_new_distinct(aggs[a].kind, col.dtype, &st) : acc_new(aggs[a].kind, col.dtype, &st);
            if (rc == LLKV_OK) { st.precision = col.precision; st.scale = col.scale; }
            if (rc == LLKV_OK) rc = acc_update(&st, &col, gn);
            arr_free(&col);
          }
          free(pv);
        }
        if (rc == LLKV_OK) {
          llkv_value *fv = &vals[(size_t)g * n_aggs + a];
          if (st.distinct) { rc = acc_finalize_distinct(&st, fv); acc_free_distinct(&st); }
          else acc_finalize(&st, fv);
          /* the finalized array goes back through plan_value_from_array (:5241): a Decimal128 cell of more than 38 digits
           * fails DecimalValue::new (llkv-plan/src/plans.rs:1160-1174) */
          if (rc == LLKV_OK && fv->dtype == LLKV_DT_DECIMAL128 && !fv->is_null) {
            const i128 raw = (i128)(((u128)(uint64_t)fv->i64_hi << 64) | (u128)(uint64_t)fv->i64);
            if (dec_digits(raw) > DEC_MAX_PRECISION) rc = fail(LLKV_INVALID_ARGUMENT, "failed to convert Decimal128 value: exceeds maximum precision");
          }
        }
      }
    }
    free(start); free(fill); free(order);
  }
  free(group_of_row);
  for (uint32_t j = 0; j < n_fields; ++j) arr_free(&m.cols[j].a);
  free(m.cols);
  if (rc != LLKV_OK) { for (size_t i = 0; i < (size_t)n_groups * n_keys; ++i) free(gkeys[i].s); free(gkeys); free(vals); return rc; }

  orc_groups *res = xcalloc(1, sizeof *res);
  res->n_groups = n_groups; res->n_keys = n_keys; res->n_aggs = n_aggs;
  res->keys = gkeys; res->vals = vals;
  if (order_by_keys && n_groups > 1) { /* insertion sort: stable, tiny result */
    for (uint32_t i = 1; i < n_groups; ++i)
      for (uint32_t j = i; j > 0 && gkey_cmp(&gkeys[(size_t)(j - 1) * n_keys], &gkeys[(size_t)j * n_keys], n_keys) > 0; --j) {
        for (uint32_t k = 0; k < n_keys; ++k) { gkey tmp = gkeys[(size_t)(j - 1) * n_keys + k]; gkeys[(size_t)(j - 1) * n_keys + k] = gkeys[(size_t)j * n_keys + k]; gkeys[(size_t)j * n_keys + k] = tmp; }
        for (uint32_t a = 0; a < n_aggs; ++a) { llkv_value tmp = vals[(size_t)(j - 1) * n_aggs + a]; vals[(size_t)(j - 1) * n_aggs + a] = vals[(size_t)j * n_aggs + a]; vals[(size_t)j * n_aggs + a] = tmp; }
      }
  }
  *out = res;
  return LLKV_OK;
}

/* --------------------------------------------------------------- hash join */
/* Integer fast path llkv-join/src/hash_join.rs:955-1417: build on the right table
 * (:209-215) — FxHashMap<key, Vec<RowRef>> in scan order (:1080-1137); probe the
 * left in scan order (:1141-1214); NULL keys never match unless null_equals_null,
 * which uses the sentinel i64::MIN (:1116-1123,1429,1441); batches of ≥ batch_size
 * pairs are flushed after finishing a probe row (:1181-1193). */
#define ORC_MAX_JOIN_KEYS 8
typedef struct jt_entry { uint64_t head, tail; int used; } jt_entry;

/* One part of a join key.  kind 0: no key (the row matches nothing); 1: a value of type `dtype` with the bit
 * pattern `bits` (fast path: the i64 image, sentinels included); 2: a string.
 * Generic path (hash_join.rs:62-148,377-505): NULL → KeyValue::Null (equal to nothing) or, under
 * null_equals_null, Utf8("<NULL>"); floats by bit pattern; values of different types are never equal; types
 * extract_key_value does not list (Date32, Boolean, Decimal128) fail the extraction → no key. */
typedef struct jk_part { int kind; int32_t dtype; uint64_t bits; const char *str; uint32_t len; } jk_part;

/* executor rules (fast == 2): normalize_join_column llkv-executor/src/lib.rs:12405-12427 then arrow-row bytes
 * (:12458-12581): Boolean and every integer type are cast to Int64 (arrow's safe cast: a UInt64 above i64::MAX
 * becomes NULL), Float32 to Float64; a NULL key part skips the row; keys are equal when their encodings are — same
 * class (Int64, Float64 by bits, Utf8, Date32, Decimal128 raw value) and same value. */
static jk_part join_key_part_executor(const orc_column *c, uint64_t row) {
  jk_part p = {0, 0, 0, NULL, 0};
  if (!col_valid(c, row)) return p;
  p.kind = 1;
  switch (c->dtype) {
  case LLKV_DT_BOOLEAN: p.dtype = LLKV_DT_INT64; p.bits = ((const uint8_t *)c->values)[row] ? 1 : 0; return p;
  case LLKV_DT_INT32: p.dtype = LLKV_DT_INT64; p.bits = (uint64_t)(int64_t)((const int32_t *)c->values)[row]; return p;
  case LLKV_DT_UINT32: p.dtype = LLKV_DT_INT64; p.bits = ((const uint32_t *)c->values)[row]; return p;
  case LLKV_DT_INT64: p.dtype = LLKV_DT_INT64; p.bits = ((const uint64_t *)c->values)[row]; return p;
  case LLKV_DT_UINT64: p.dtype = LLKV_DT_INT64; p.bits = ((const uint64_t *)c->values)[row]; if (p.bits >> 63) p.kind = 0; return p;
  case LLKV_DT_FLOAT32: { double d = (double)((const float *)c->values)[row]; p.dtype = LLKV_DT_FLOAT64; memcpy(&p.bits, &d, 8); return p; }
  case LLKV_DT_FLOAT64: p.dtype = LLKV_DT_FLOAT64; p.bits = ((const uint64_t *)c->values)[row]; return p;
  case LLKV_DT_DATE32: p.dtype = LLKV_DT_DATE32; p.bits = (uint64_t)(int64_t)((const int32_t *)c->values)[row]; return p;
  case LLKV_DT_DECIMAL128: p.dtype = LLKV_DT_DECIMAL128; p.bits = ((const uint64_t *)c->values)[2 * row]; p.len = (uint32_t)(((const uint64_t *)c->values)[2 * row + 1] & 0xffffffffu); return p;
  case LLKV_DT_UTF8: p.kind = 2; p.dtype = LLKV_DT_UTF8; p.str = (const char *)c->data + c->offsets[row]; p.len = (uint32_t)(c->offsets[row + 1] - c->offsets[row]); return p;
  default: p.kind = 0; return p;
  }
}

static jk_part join_key_part(const orc_column *c, uint64_t row, int null_eq, int fast) {
  if (fast == 2) return join_key_part_executor(c, row);
  jk_part p = {0, c->dtype, 0, NULL, 0};
  if (!col_valid(c, row)) {
    if (!null_eq) return p;
    if (fast) { /* per-type sentinels of the fast paths, hash_join.rs:1429-1465 */
      p.kind = 1;
      p.bits = c->dtype == LLKV_DT_INT64 ? (uint64_t)INT64_MIN : c->dtype == LLKV_DT_UINT64 ? UINT64_MAX : c->dtype == LLKV_DT_UINT32 ? (uint64_t)UINT32_MAX : (uint64_t)(int64_t)INT32_MIN;
      return p;
    }
    p.kind = 2; p.dtype = LLKV_DT_UTF8; p.str = "<NULL>"; p.len = 6;
    return p;
  }
  switch (c->dtype) {
  case LLKV_DT_INT64: case LLKV_DT_UINT64: case LLKV_DT_FLOAT64: p.kind = 1; p.bits = ((const uint64_t *)c->values)[row]; return p;
  case LLKV_DT_INT32: p.kind = 1; p.bits = (uint64_t)(int64_t)((const int32_t *)c->values)[row]; return p;
  case LLKV_DT_UINT32: case LLKV_DT_FLOAT32: p.kind = 1; p.bits = ((const uint32_t *)c->values)[row]; return p;
  case LLKV_DT_UTF8: p.kind = 2; p.str = (const char *)c->data + c->offsets[row]; p.len = (uint32_t)(c->offsets[row + 1] - c->offsets[row]); return p;
  default: return p; /* "Unsupported join key type" → the row is skipped (`if let Ok(key)`) */
  }
}

static int join_row_hash(const orc_column *const *cols, const llkv_join_key *keys, uint32_t n_keys, int fast, uint64_t row, uint64_t *out) {
  uint64_t h = 1469598103934665603ULL;
  for (uint32_t i = 0; i < n_keys; ++i) {
    jk_part p = join_key_part(cols[i], row, keys[i].null_equals_null, fast);
    if (p.kind == 0) return 0;
    if (p.kind == 1) h = (h ^ p.bits) * 1099511628211ULL;
    else for (uint32_t j = 0; j < p.len; ++j) h = (h ^ (uint8_t)p.str[j]) * 1099511628211ULL;
    h ^= h >> 29;
  }
  *out = h;
  return 1;
}

/* both rows have keys (join_row_hash succeeded) */
static int join_rows_equal(const orc_column *const *ca, uint64_t ra, const orc_column *const *cb, uint64_t rb, const llkv_join_key *keys, uint32_t n_keys, int fast) {
  for (uint32_t i = 0; i < n_keys; ++i) {
    jk_part a = join_key_part(ca[i], ra, keys[i].null_equals_null, fast), b = join_key_part(cb[i], rb, keys[i].null_equals_null, fast);
    if (a.kind != b.kind || a.dtype != b.dtype) return 0;
    if (a.kind == 1 ? (a.bits != b.bits || a.len != b.len) : (a.len != b.len || memcmp(a.str, b.str, a.len) != 0)) return 0;
  }
  return 1;
}

/* The rows a side's scan delivers (scan_stream(all user columns, ScanStreamOptions::default()), hash_join.rs:203-240,
 * :346-375) and the batches they arrive in: the surviving rows of every 65 536-row-id window (execute.rs:297-372). */
typedef struct join_rows {
  uint64_t *rows;        /* row ids, ascending */
  uint64_t n;
  uint64_t *batch_start; /* [n_batches + 1] indices into rows */
  uint64_t n_batches;
} join_rows;
static void join_rows_free(join_rows *r) { free(r->rows); free(r->batch_start); memset(r, 0, sizeof *r); }

/* `fields` == NULL: every row (the index-pair delivery); else GatherNullPolicy::DropNulls over the projected fields
 * (store/projection.rs:1326-1330): a row NULL in all of them is dropped; a window left without rows is no batch. */
static int32_t join_scan_rows(const orc_table *t, const uint32_t *fields, uint32_t n_fields, int one_batch, join_rows *out) {
  memset(out, 0, sizeof *out);
  out->rows = xmalloc((t->rows ? t->rows : 1) * sizeof(uint64_t));
  out->batch_start = xmalloc((t->rows / ROW_STREAM_CHUNK_SIZE + 2) * sizeof(uint64_t));
  for (uint64_t w0 = 0; w0 < t->rows; w0 += ROW_STREAM_CHUNK_SIZE) {
    const uint64_t wn = t->rows - w0 < ROW_STREAM_CHUNK_SIZE ? t->rows - w0 : ROW_STREAM_CHUNK_SIZE, first = out->n;
    for (uint64_t r = w0; r < w0 + wn; ++r) {
      int keep = fields == NULL;
      for (uint32_t j = 0; j < n_fields && !keep; ++j) keep = col_valid(find_col(t, fields[j]), r);
      if (keep) out->rows[out->n++] = r;
    }
    if (out->n > first && (!one_batch || out->n_batches == 0)) out->batch_start[out->n_batches++] = first;
  }
  out->batch_start[out->n_batches] = out->n;
  /* a table whose every row is dropped comes out of the scan as ONE synthetic batch of total_rows NULL rows
   * (llkv-scan/src/execute.rs:355-372 `if !emitted_rows`, llkv-compute/src/projection.rs:36-66): the rows are back, every cell
   * NULL as it was, in one batch instead of one per window */
  if (fields && t->rows && out->n == 0) {
    for (uint64_t r = 0; r < t->rows; ++r) out->rows[r] = r;
    out->n = t->rows;
    out->n_batches = 1;
    out->batch_start[0] = 0;
    out->batch_start[1] = out->n;
  }
  return LLKV_OK;
}

typedef void (*join_pair_sink)(const uint64_t *left_rows, const uint64_t *right_rows, uint64_t n_pairs, void *user, int32_t *rc);

/* hash_join_stream hash_join.rs:151-335 / the integer fast paths :955-1417 / build_join_match_indices
 * llkv-executor/src/lib.rs:12458-12581 over the rows the two scans delivered; `fast`: 1 integer fast path, 0 generic
 * typed-key path, 2 executor rules. */
static int32_t hash_join_core(const orc_table *left, const orc_table *right, const llkv_join_key *keys, uint32_t n_keys, int jt, uint64_t batch_size,
                              int fast, const join_rows *L, const join_rows *R, join_pair_sink sink, void *user) {
  const orc_column *lcs[ORC_MAX_JOIN_KEYS], *rcs[ORC_MAX_JOIN_KEYS];
  for (uint32_t i = 0; i < n_keys; ++i) {
    lcs[i] = find_col(left, keys[i].left_field);
    rcs[i] = find_col(right, keys[i].right_field);
    if (!lcs[i] || !rcs[i]) return fail(LLKV_NOT_FOUND, "join key field not found");
  }
  /* build: open addressing on the key (first row with that key), chained row lists in insertion order */
  uint64_t cap = 16;
  while (cap < R->n * 2 + 16) cap <<= 1;
  jt_entry *tab = xcalloc(cap, sizeof(jt_entry));
  uint64_t *next = xmalloc((right->rows ? right->rows : 1) * sizeof(uint64_t));
  for (uint64_t i = 0; i < R->n; ++i) {
    const uint64_t r = R->rows[i];
    next[r] = UINT64_MAX;
    uint64_t hk;
    if (!join_row_hash(rcs, keys, n_keys, fast, r, &hk)) continue;
    uint64_t h = (hk * 0x9E3779B97F4A7C15ULL) & (cap - 1);
    while (tab[h].used && !join_rows_equal(rcs, tab[h].head, rcs, r, keys, n_keys, fast)) h = (h + 1) & (cap - 1);
    if (!tab[h].used) { tab[h].used = 1; tab[h].head = tab[h].tail = r; }
    else { next[tab[h].tail] = r; tab[h].tail = r; }
  }
  /* probe: one scan batch at a time (:1010-1070); the generic path cuts every scan batch into slices of batch_size
   * rows first (:228-246); inside a batch / slice the pairs are flushed after the probe row that brings them to
   * >= batch_size, and at its end (:1181-1213, :509-565); the executor materialises one batch */
  uint64_t pcap = 1024, np = 0;
  uint64_t *pl = xmalloc(pcap * sizeof(uint64_t)), *pr = xmalloc(pcap * sizeof(uint64_t));
  int32_t rc = LLKV_OK;
#define JOIN_PUSH(L_, R_) do { if (np == pcap) { pcap *= 2; pl = xrealloc(pl, pcap * sizeof(uint64_t)); pr = xrealloc(pr, pcap * sizeof(uint64_t)); } \
                               pl[np] = (L_); pr[np] = (R_); ++np; } while (0)
  for (uint64_t b = 0; b < L->n_batches && rc == LLKV_OK; ++b) {
    const uint64_t b0 = L->batch_start[b], m = L->batch_start[b + 1] - b0;
    for (uint64_t idx = 0; idx < m && rc == LLKV_OK; ++idx) {
      const uint64_t l = L->rows[b0 + idx];
      int matched = 0;
      uint64_t h = 0, hk;
      if (join_row_hash(lcs, keys, n_keys, fast, l, &hk)) {
        h = (hk * 0x9E3779B97F4A7C15ULL) & (cap - 1);
        while (tab[h].used && !join_rows_equal(rcs, tab[h].head, lcs, l, keys, n_keys, fast)) h = (h + 1) & (cap - 1);
        matched = tab[h].used;
      }
      switch (jt) {
      case LLKV_JOIN_INNER:
        if (matched) for (uint64_t r = tab[h].head; r != UINT64_MAX; r = next[r]) JOIN_PUSH(l, r);
        break;
      case LLKV_JOIN_LEFT: /* unmatched left rows padded with NULLs */
        if (matched) for (uint64_t r = tab[h].head; r != UINT64_MAX; r = next[r]) JOIN_PUSH(l, r);
        else JOIN_PUSH(l, UINT64_MAX);
        break;
      case LLKV_JOIN_SEMI: if (matched) JOIN_PUSH(l, 0); break;
      case LLKV_JOIN_ANTI: if (!matched) JOIN_PUSH(l, 0); break;
      }
      const int boundary = idx + 1 == m || (fast == 0 && (idx + 1) % batch_size == 0);
      if (np && (np >= batch_size || boundary)) { sink(pl, (jt == LLKV_JOIN_SEMI || jt == LLKV_JOIN_ANTI) ? NULL : pr, np, user, &rc); np = 0; }
    }
  }
#undef JOIN_PUSH
  free(pl); free(pr); free(next); free(tab);
  return rc;
}

/* validate_join_options llkv-join/src/lib.rs:284-310; hash_join.rs:328-332; llkv-executor/src/lib.rs:12387-12391 */
static int32_t join_check_options(const llkv_join_options *options, uint32_t n_keys, int *executor, uint64_t *batch_size, int *jt) {
  *executor = options && options->key_rules == LLKV_JOIN_KEYS_EXECUTOR;
  *batch_size = *executor ? UINT64_MAX : options ? options->batch_size : 8192;
  *jt = options ? options->join_type : LLKV_JOIN_INNER;
  if (*executor && *jt != LLKV_JOIN_INNER && *jt != LLKV_JOIN_LEFT)
    return fail(LLKV_INTERNAL, "join type not supported in hash_join_table_batches; use llkv-join");
  if (*executor && n_keys == 0) return fail(LLKV_INVALID_ARGUMENT, "executor join rules need at least one key pair");
  if (*batch_size == 0) return fail(LLKV_INVALID_ARGUMENT, "join batch_size must be greater than zero");
  if (*jt == LLKV_JOIN_RIGHT || *jt == LLKV_JOIN_FULL) return fail(LLKV_INVALID_ARGUMENT, "Right and Full joins are not yet implemented");
  return LLKV_OK;
}
/* integer fast path: one key, identical integer key types (hash_join.rs:171-200); everything else takes the generic
 * typed-key path; 2 = the executor's key rules */
static int32_t join_path(const orc_table *left, const orc_table *right, const llkv_join_key *keys, uint32_t n_keys, int executor, int *fast) {
  if (n_keys > ORC_MAX_JOIN_KEYS) return fail(LLKV_UNSUPPORTED, "more than %d join key pairs (n_keys=%u)", ORC_MAX_JOIN_KEYS, n_keys); /* (the reference has no limit, hash_join.rs:200-335: a fixed array here) */
  const orc_column *l0 = find_col(left, keys[0].left_field), *r0 = find_col(right, keys[0].right_field);
  for (uint32_t i = 0; i < n_keys; ++i)
    if (!find_col(left, keys[i].left_field) || !find_col(right, keys[i].right_field)) return fail(LLKV_NOT_FOUND, "join key field not found");
  *fast = n_keys == 1 && l0->dtype == r0->dtype &&
          (l0->dtype == LLKV_DT_INT32 || l0->dtype == LLKV_DT_INT64 || l0->dtype == LLKV_DT_UINT32 || l0->dtype == LLKV_DT_UINT64);
  if (executor) *fast = 2;
  return LLKV_OK;
}

typedef struct pair_fwd { orc_on_join_batch cb; void *user; } pair_fwd;
static void pair_forward(const uint64_t *l, const uint64_t *r, uint64_t n, void *user, int32_t *rc) { (void)rc; pair_fwd *f = user; f->cb(l, r, n, f->user); }

int32_t orc_hash_join(const orc_table *left, const orc_table *right, const llkv_join_key *keys,
                      uint32_t n_keys, const llkv_join_options *options, orc_on_join_batch on_batch, void *user) {
  int executor, jt, fast = 0;
  uint64_t batch_size;
  int32_t rc = join_check_options(options, n_keys, &executor, &batch_size, &jt);
  if (rc) return rc;
  if (n_keys == 0) { /* cross_product_stream hash_join.rs:1500-1599, cross_join_pair cartesian.rs:22-80 */
    if (jt == LLKV_JOIN_SEMI || jt == LLKV_JOIN_ANTI) return fail(LLKV_INTERNAL, "cross join schema mismatch");
    if (right->rows == 0 && jt == LLKV_JOIN_INNER) return LLKV_OK;
    for (uint64_t l0 = 0; l0 < left->rows; l0 += ROW_STREAM_CHUNK_SIZE) {
      uint64_t ln = left->rows - l0 < ROW_STREAM_CHUNK_SIZE ? left->rows - l0 : ROW_STREAM_CHUNK_SIZE;
      if (right->rows == 0) {
        uint64_t *l = xmalloc(ln * 8), *r = xmalloc(ln * 8);
        for (uint64_t i = 0; i < ln; ++i) { l[i] = l0 + i; r[i] = UINT64_MAX; }
        on_batch(l, r, ln, user);
        free(l); free(r);
        continue;
      }
      for (uint64_t r0 = 0; r0 < right->rows; r0 += ROW_STREAM_CHUNK_SIZE) {
        uint64_t rn = right->rows - r0 < ROW_STREAM_CHUNK_SIZE ? right->rows - r0 : ROW_STREAM_CHUNK_SIZE, np = ln * rn, k = 0;
        uint64_t *l = xmalloc(np * 8), *r = xmalloc(np * 8);
        for (uint64_t i = 0; i < ln; ++i) for (uint64_t j = 0; j < rn; ++j) { l[k] = l0 + i; r[k++] = r0 + j; }
        on_batch(l, r, np, user);
        free(l); free(r);
      }
    }
    return LLKV_OK;
  }
  if ((rc = join_path(left, right, keys, n_keys, executor, &fast))) return rc;
  join_rows L, R;
  if ((rc = join_scan_rows(left, NULL, 0, executor, &L))) return rc;
  if ((rc = join_scan_rows(right, NULL, 0, 0, &R))) { join_rows_free(&L); return rc; }
  pair_fwd f = {on_batch, user};
  rc = hash_join_core(left, right, keys, n_keys, jt, batch_size, fast, &L, &R, pair_forward, &f);
  join_rows_free(&L); join_rows_free(&R);
  return rc;
}

/* --- the joined RecordBatches ------------------------------------------------------------------------------------
 * emit_joined_batch / emit_left_joined_batch / emit_semi_batch hash_join.rs:715-772: the probe batch's columns gathered
 * at the probe rows, then the build batches' at the build rows (None → NULL, gather_optional_indices_from_batches
 * llkv-column-map/src/gather.rs:167-260); output schema build_output_schema :877-943. */
typedef struct batch_sink {
  const orc_table *left, *right;
  const llkv_join_output *out;
  int left_only;
  const char *const *names;
  orc_on_join_record_batch cb;
  void *user;
} batch_sink;

static arr gather_optional(const orc_column *c, const uint64_t *ids, uint64_t n) {
  uint64_t *safe = xmalloc((n ? n : 1) * sizeof(uint64_t));
  for (uint64_t i = 0; i < n; ++i) safe[i] = ids[i] == UINT64_MAX ? 0 : ids[i];
  arr a;
  int any_row = 0;
  for (uint64_t i = 0; i < n; ++i) any_row |= ids[i] != UINT64_MAX;
  if (any_row) a = gather_column(c, safe, n);
  else { /* nothing to read (the build side may hold no row at all): NULL cells */
    memset(&a, 0, sizeof a);
    a.dtype = c->dtype; a.precision = c->precision; a.scale = c->scale; a.n = n;
    a.valid = xcalloc(n ? n : 1, 1);
    if (c->dtype == LLKV_DT_UTF8) { a.strings = xcalloc(n ? n : 1, sizeof(char *)); for (uint64_t i = 0; i < n; ++i) a.strings[i] = xcalloc(1, 1); }
    else a.values = xcalloc(n ? n : 1, dtype_width(c->dtype));
  }
  for (uint64_t i = 0; i < n; ++i)
    if (ids[i] == UINT64_MAX) {
      a.valid[i] = 0;
      if (a.strings) a.strings[i][0] = 0;
      else memset((char *)a.values + i * dtype_width(c->dtype), 0, dtype_width(c->dtype));
    }
  free(safe);
  return a;
}

static void batch_from_pairs(const uint64_t *l, const uint64_t *r, uint64_t n, void *user, int32_t *rc) {
  (void)rc;
  batch_sink *b = user;
  const uint32_t nl = b->out->n_left, nr = b->left_only ? 0 : b->out->n_right, nc = nl + nr;
  arr *cols = xcalloc(nc ? nc : 1, sizeof(arr));
  orc_batch_column *bc = xcalloc(nc ? nc : 1, sizeof *bc);
  for (uint32_t i = 0; i < nl; ++i) cols[i] = gather_column(find_col(b->left, b->out->left_columns[i].field_id), l, n);
  for (uint32_t i = 0; i < nr; ++i) cols[nl + i] = gather_optional(find_col(b->right, b->out->right_columns[i].field_id), r, n);
  for (uint32_t i = 0; i < nc; ++i) {
    bc[i].dtype = cols[i].dtype; bc[i].values = cols[i].values; bc[i].valid = cols[i].valid;
    bc[i].strings = (const char *const *)cols[i].strings; bc[i].precision = cols[i].precision; bc[i].scale = cols[i].scale;
  }
  orc_batch ob = {n, nc, bc, NULL};
  b->cb(&ob, b->names, b->user);
  for (uint32_t i = 0; i < nc; ++i) arr_free(&cols[i]);
  free(cols); free(bc);
}

int32_t orc_hash_join_batches(const orc_table *left, const orc_table *right, const llkv_join_key *keys, uint32_t n_keys,
                              const llkv_join_options *options, const llkv_join_output *output, orc_on_join_record_batch on_batch, void *user) {
  int executor, jt, fast = 0;
  uint64_t batch_size;
  int32_t rc = join_check_options(options, n_keys, &executor, &batch_size, &jt);
  if (rc) return rc;
  const int left_only = jt == LLKV_JOIN_SEMI || jt == LLKV_JOIN_ANTI;
  const uint32_t nl = output->n_left, nr = output->n_right;
  for (uint32_t i = 0; i < nl; ++i) if (!find_col(left, output->left_columns[i].field_id)) return fail(LLKV_NOT_FOUND, "join output field %u not found", output->left_columns[i].field_id);
  for (uint32_t i = 0; i < nr; ++i) if (!find_col(right, output->right_columns[i].field_id)) return fail(LLKV_NOT_FOUND, "join output field %u not found", output->right_columns[i].field_id);
  /* build_output_schema :877-943: left fields, then right fields, a name already taken gets "_1"; SEMI / ANTI: left only;
   * the executor keeps the names as given (llkv-executor/src/lib.rs:12237-12244) */
  char **names = xcalloc(nl + nr + 1, sizeof(char *));
  uint32_t nn = 0;
  for (uint32_t i = 0; i < nl; ++i) { const char *s = output->left_columns[i].name ? output->left_columns[i].name : ""; names[nn] = xmalloc(strlen(s) + 1); strcpy(names[nn++], s); }
  for (uint32_t i = 0; i < nr && !left_only; ++i) {
    const char *s = output->right_columns[i].name ? output->right_columns[i].name : "";
    int taken = 0;
    for (uint32_t k = 0; k < nn && !executor; ++k) taken |= strcmp(names[k], s) == 0;
    names[nn] = xmalloc(strlen(s) + 3);
    strcpy(names[nn], s);
    if (taken) strcat(names[nn], "_1");
    ++nn;
  }
  uint32_t *lf = xmalloc((nl ? nl : 1) * sizeof(uint32_t)), *rf = xmalloc((nr ? nr : 1) * sizeof(uint32_t));
  for (uint32_t i = 0; i < nl; ++i) lf[i] = output->left_columns[i].field_id;
  for (uint32_t i = 0; i < nr; ++i) rf[i] = output->right_columns[i].field_id;
  batch_sink sink = {left, right, output, left_only, (const char *const *)names, on_batch, user};
  join_rows L, R;
  memset(&L, 0, sizeof L); memset(&R, 0, sizeof R);
  /* the executor's tables come with their NULL rows (collect_table_data scans with include_nulls) */
  if (nr && (rc = join_scan_rows(right, executor ? NULL : rf, nr, 0, &R))) goto done;
  if (nr == 0) { R.rows = xmalloc(8); R.batch_start = xcalloc(1, 8); } /* no right projections: nothing is scanned (:211-215) */
  if (n_keys == 0) { /* cross_product_stream :1500-1599 */
    const int right_empty = R.n == 0;
    if (right_empty && jt == LLKV_JOIN_INNER) goto done;
    if (nl == 0) goto done;
    if ((rc = join_scan_rows(left, lf, nl, 0, &L))) goto done;
    for (uint64_t lb = 0; lb < L.n_batches && rc == LLKV_OK; ++lb) {
      const uint64_t l0 = L.batch_start[lb], ln = L.batch_start[lb + 1] - l0;
      if (right_empty) {
        if (jt != LLKV_JOIN_LEFT) continue;
        uint64_t *pr = xmalloc(ln * 8);
        for (uint64_t i = 0; i < ln; ++i) pr[i] = UINT64_MAX; /* synthesize_left_join_nulls :1468-1497 */
        batch_from_pairs(L.rows + l0, pr, ln, &sink, &rc);
        free(pr);
        continue;
      }
      for (uint64_t rb = 0; rb < R.n_batches && rc == LLKV_OK; ++rb) {
        if (left_only) { rc = fail(LLKV_INTERNAL, "cross join schema mismatch: semi / anti joins deliver left columns only"); break; } /* cartesian.rs:36-44 */
        const uint64_t r0 = R.batch_start[rb], rn = R.batch_start[rb + 1] - r0, np = ln * rn;
        uint64_t *pl = xmalloc(np * 8), *pr = xmalloc(np * 8), k = 0;
        for (uint64_t i = 0; i < ln; ++i) for (uint64_t j = 0; j < rn; ++j) { pl[k] = L.rows[l0 + i]; pr[k++] = R.rows[r0 + j]; }
        batch_from_pairs(pl, pr, np, &sink, &rc);
        free(pl); free(pr);
      }
    }
    goto done;
  }
  if ((rc = join_path(left, right, keys, n_keys, executor, &fast))) goto done;
  if (nl == 0) goto done; /* no left projections: nothing is probed (:226) */
  if ((rc = join_scan_rows(left, executor ? NULL : lf, nl, executor, &L))) goto done;
  if (fast == 0 && L.n != left->rows) { rc = fail(LLKV_UNSUPPORTED, "generic join path over a probe side with rows that are NULL in every user column"); goto done; } /* (kept in step with the GPU path's limit) */
  /* a build side without a batch: a LEFT join's gather_optional_indices_from_batches returns no arrays and
   * RecordBatch::try_new fails on the column count — the fast path logs and drops that error per probe batch
   * (:1058-1060), the generic path returns it (:313-317) */
  if (R.n_batches == 0 && jt == LLKV_JOIN_LEFT && !executor) {
    if (fast == 1 || L.n == 0) goto done;
    rc = fail(LLKV_INTERNAL, "Invalid argument error: number of columns(%u) must match number of fields(%u) in schema", nl, nl + nr);
    goto done;
  }
  rc = hash_join_core(left, right, keys, n_keys, jt, batch_size, fast, &L, &R, batch_from_pairs, &sink);
done:
  join_rows_free(&L); join_rows_free(&R);
  for (uint32_t i = 0; i < nn; ++i) free(names[i]);
  free(names); free(lf); free(rf);
  return rc;
}
