// plan.cpp — see plan.hpp.
#include "plan.hpp"

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace llkv {

typedef __int128 i128;
typedef unsigned __int128 u128;

static constexpr int kMaxColsHost = 16;
static constexpr int kMaxLitsHost = 48; // = kMaxLits (scan_params.h)
static constexpr int kMaxKeysHost = 4;
bool plan_exact_f64_sums();
static constexpr uint32_t kMaxDenseGroups = 64; // bounded further by the LDS image (lanes · 2 KiB ≤ 160 KiB)
// shared-image route: one image [lane][group] of 8-byte slots per workgroup; 150 KiB of the CU's 160 KiB of LDS
static constexpr uint32_t kMaxImageGroups = 1u << 16;
static constexpr uint32_t kMaxPartGroups = 1u << 24; // partitioned route: dense group ids (the result image is ng × k × 8 bytes of HBM)
static constexpr size_t kMaxImageBytes = 158u * 1024; // of the 160 KB of a CU (the kernel keeps one more word)
static constexpr int kMaxImagePasses = 4; // scans of the table a shared-image plan may take (the groups cut into slices)

const char *dtype_name(int32_t dt) {
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

const char *dtype_tag(int32_t dt) {
  switch (dt) {
  case LLKV_DT_INT64: case LLKV_DT_DECIMAL128: return "I64"; // decimals are staged narrowed to 64 bits
  case LLKV_DT_FLOAT64: return "F64";
  case LLKV_DT_INT32: case LLKV_DT_DATE32: return "I32";
  case LLKV_DT_UINT64: return "U64";
  case LLKV_DT_UINT32: return "U32";
  case LLKV_DT_FLOAT32: return "F32";
  case LLKV_DT_UTF8: case LLKV_DT_BOOLEAN: return "U8";
  default: return "?";
  }
}

uint32_t dtype_width(int32_t dt) {
  switch (dt) {
  case LLKV_DT_INT64: case LLKV_DT_FLOAT64: case LLKV_DT_UINT64: case LLKV_DT_DECIMAL128: return 8;
  case LLKV_DT_INT32: case LLKV_DT_DATE32: case LLKV_DT_UINT32: case LLKV_DT_FLOAT32: return 4;
  case LLKV_DT_UTF8: case LLKV_DT_BOOLEAN: return 1;
  default: return 0;
  }
}

uint32_t dtype_out_width(int32_t dt) { return dt == LLKV_DT_DECIMAL128 ? 16 : dtype_width(dt); }

static i128 lit_i128(const llkv_literal &l) { return (i128)(((u128)(uint64_t)l.hi << 64) | (u128)l.lo); }

static const char *lit_kind(const llkv_literal &l) {
  switch (l.tag) {
  case LLKV_LIT_FLOAT64: return "float";
  case LLKV_LIT_BOOLEAN: return "boolean";
  case LLKV_LIT_STRING: return "string";
  case LLKV_LIT_DATE32: return "date";
  case LLKV_LIT_DECIMAL128: return "decimal";
  case LLKV_LIT_NULL: return "null";
  default: return "integer";
  }
}

// compiler-rt __powidf2 — what Rust's f64::powi lowers to (llkv-types/src/decimal.rs:102-108)
static double powi_f64(double a, int b) {
  const bool recip = b < 0;
  double r = 1;
  for (;;) {
    if (b & 1) r *= a;
    b /= 2;
    if (b == 0) break;
    a *= a;
  }
  return recip ? 1 / r : r;
}

static int set_err(std::string *err, int code, const std::string &msg) {
  if (err) *err = msg;
  return code;
}

// llkv-types/src/literal.rs:364-520
int cast_literal_for_column(const llkv_literal &lit, int32_t dtype, NativeLit *out, std::string *err) {
  auto int_cast = [&](i128 lo, i128 hi, const char *target) -> int {
    i128 v;
    if (lit.tag == LLKV_LIT_INT128) v = lit_i128(lit);
    else if (lit.tag == LLKV_LIT_DECIMAL128 && lit.scale == 0) v = lit_i128(lit);
    else return set_err(err, LLKV_PREDICATE_BUILD, std::string("literal cast error: expected integer, got ") + lit_kind(lit));
    if (v < lo || v > hi) return set_err(err, LLKV_PREDICATE_BUILD, std::string("literal cast error: value out of range for ") + target);
    out->i = (int64_t)v;
    return LLKV_OK;
  };
  auto f64_cast = [&](double *dst) -> int {
    switch (lit.tag) {
    case LLKV_LIT_FLOAT64: *dst = lit.f64; return LLKV_OK;
    case LLKV_LIT_INT128: *dst = (double)lit_i128(lit); return LLKV_OK;
    case LLKV_LIT_DECIMAL128: {
      i128 raw = lit_i128(lit);
      *dst = raw == 0 ? 0.0 : (double)raw / powi_f64(10.0, lit.scale);
      return LLKV_OK;
    }
    default: return set_err(err, LLKV_PREDICATE_BUILD, std::string("literal cast error: expected float, got ") + lit_kind(lit));
    }
  };
  *out = NativeLit{};
  switch (dtype) {
  case LLKV_DT_INT64: return int_cast((i128)INT64_MIN, (i128)INT64_MAX, "i64");
  case LLKV_DT_INT32: case LLKV_DT_DATE32: return int_cast(INT32_MIN, INT32_MAX, "i32"); // Date32 filters as i32
  case LLKV_DT_UINT32: return int_cast(0, UINT32_MAX, "u32");
  case LLKV_DT_UINT64: {
    out->is_unsigned = true;
    i128 v;
    if (lit.tag == LLKV_LIT_INT128 || (lit.tag == LLKV_LIT_DECIMAL128 && lit.scale == 0)) v = lit_i128(lit);
    else return set_err(err, LLKV_PREDICATE_BUILD, std::string("literal cast error: expected integer, got ") + lit_kind(lit));
    if (v < 0 || v > (i128)UINT64_MAX) return set_err(err, LLKV_PREDICATE_BUILD, "literal cast error: value out of range for u64");
    out->i = (int64_t)(uint64_t)v;
    return LLKV_OK;
  }
  case LLKV_DT_FLOAT64: out->is_float = true; return f64_cast(&out->f);
  case LLKV_DT_FLOAT32: {
    out->is_float = true;
    double v;
    int rc = f64_cast(&v);
    if (rc) return rc;
    float c = (float)v;
    if (!std::isfinite(c)) return set_err(err, LLKV_PREDICATE_BUILD, "literal cast error: float out of range for f32");
    out->f = (double)c;
    return LLKV_OK;
  }
  default:
    return set_err(err, LLKV_INTERNAL, std::string("Filtering on type ") + dtype_name(dtype) + " is not supported");
  }
}

namespace {

} // namespace
double parse_numeric_or_zero(const std::string &text) {
  // str::trim(): Unicode White_Space — the ASCII ones and the multi-byte ones a UTF-8 string can start / end with
  auto ws_at = [&](size_t i, bool backwards) -> size_t { // length of the white-space character at i (ending at i), 0 if none
    static const char *const multi[] = {"\xC2\x85", "\xC2\xA0", "\xE1\x9A\x80", "\xE2\x80\x80", "\xE2\x80\x81", "\xE2\x80\x82", "\xE2\x80\x83",
                                        "\xE2\x80\x84", "\xE2\x80\x85", "\xE2\x80\x86", "\xE2\x80\x87", "\xE2\x80\x88", "\xE2\x80\x89", "\xE2\x80\x8A",
                                        "\xE2\x80\xA8", "\xE2\x80\xA9", "\xE2\x80\xAF", "\xE2\x81\x9F", "\xE3\x80\x80"};
    const unsigned char c = (unsigned char)text[i];
    if (c == ' ' || (c >= 9 && c <= 13)) return 1;
    for (const char *m : multi) {
      const size_t n = std::strlen(m);
      if (!backwards) { if (text.compare(i, n, m) == 0) return n; }
      else if (i + 1 >= n && text.compare(i + 1 - n, n, m) == 0) return n;
    }
    return 0;
  };
  size_t b = 0, e = text.size();
  for (size_t n; b < e && (n = ws_at(b, false)); b += n) {}
  for (size_t n; e > b && (n = ws_at(e - 1, true)); e -= n) {}
  const std::string t = text.substr(b, e - b);
  // <f64 as FromStr>: Sign? ( 'inf' | 'infinity' | 'nan' | Number ),  Number = ( Digit+ | Digit+ '.' Digit* | Digit* '.' Digit+ ) Exp?,
  // Exp = ('e' | 'E') Sign? Digit+
  size_t i = 0;
  bool neg = false;
  if (i < t.size() && (t[i] == '+' || t[i] == '-')) neg = t[i++] == '-';
  auto ieq = [&](const char *w) {
    const size_t n = std::strlen(w);
    if (t.size() - i != n) return false;
    for (size_t k = 0; k < n; ++k) if (std::tolower((unsigned char)t[i + k]) != w[k]) return false;
    return true;
  };
  if (ieq("inf") || ieq("infinity")) return neg ? -INFINITY : INFINITY;
  if (ieq("nan")) return neg ? -std::nan("") : std::nan("");
  size_t k = i, int_digits = 0, frac_digits = 0;
  while (k < t.size() && std::isdigit((unsigned char)t[k])) { ++k; ++int_digits; }
  if (k < t.size() && t[k] == '.') {
    ++k;
    while (k < t.size() && std::isdigit((unsigned char)t[k])) { ++k; ++frac_digits; }
  }
  if (int_digits + frac_digits == 0) return 0.0;
  if (k < t.size() && (t[k] == 'e' || t[k] == 'E')) {
    ++k;
    if (k < t.size() && (t[k] == '+' || t[k] == '-')) ++k;
    size_t exp_digits = 0;
    while (k < t.size() && std::isdigit((unsigned char)t[k])) { ++k; ++exp_digits; }
    if (exp_digits == 0) return 0.0;
  }
  if (k != t.size()) return 0.0;
  return std::strtod(t.c_str(), nullptr); // a decimal literal of that grammar: both sides round it correctly
}
static std::string cols_string(const LoweredPlan &p, uint64_t *bytes);
namespace {

struct PlanValueInfo { bool is_decimal = false; int scale = 0; i128 lo = 0, hi = 0; bool bounded = false; };

struct Lowering {
  const ColumnResolver &resolve;
  LoweredPlan &p;
  std::string *err;
  bool grouped;
  // f64 nodes that must return a computed NaN exactly as the reference's host does (sign included): set where the
  // NaN's bits are observable — totalOrder compares, IN lists, projected / emitted values — and left off for aggregate
  // arguments, where no accumulator looks at them (fused_scan.hip.h: f64_result_as_sse2; it costs Q1 5 %)
  bool exact_nan = false;
  int last_fast_32 = 0; // expr_fast: the expression's root type is Int32 (1) / UInt32 (2): the nodes are wrapped in Fit32
  std::string nan_flag(bool is_float) const { return exact_nan && is_float ? ",1" : ""; }
  // shared-image plans: f64 sums as exact two-level pairs (SumF64X), which need a bound on |argument|
  bool exact_f64 = false;
  // planning option (plan_set_exact_f64_sums): every f64 SUM / AVG / TOTAL is the correctly rounded EXACT sum of its rows'
  // values, on every route — the last grid sits at the ulp of the smallest non-zero |argument| the statistics allow, so
  // no row drops a bit
  bool strict_exact = false;
  bool image_plan = false;
  bool whole_table_image = false; // partitioned route: one LDS image may receive every row of the table
  uint64_t table_rows = 0; // rows of the table the plan scans (the N of the exact sums)
  bool allow_dict_num = true; // the kernels of this plan see ScanParams::dict_num (not the sort route's reduce kernel)
  bool allow_sorted_distinct = false; // reduce plans: DISTINCT aggregates over ONE Int64 / Float64 column (it sorts last)
  // shared-image plans: exchange lane j of a lane group = xf(kernel lane src) (LoweredPlan::image_src / image_xf); empty = as is
  std::vector<std::vector<std::pair<uint8_t, uint8_t>>> group_expand = {};
  std::vector<std::pair<uint8_t, uint8_t>> next_expand = {};
  bool bounds_all_finite = true; // (expr_bounds) no column of the expression holds NaN / ±∞
  // shared-image plans: the largest |contribution| a row makes to any lane of a lane group (counts: 1; SumI64Fast: max |v| of
  // its column) when every lane of the group is a plain integer add; −1: the group needs 8-byte cells.  Set before add_group.
  double next_narrow = -1.0;
  std::vector<double> group_narrow = {};

  // What the column statistics say about an aggregate argument: an interval [lo, hi] (integer min / max, largest
  // finite |v| of float columns) and `nz`, a lower bound on |value| wherever the value is not zero (smallest non-zero
  // finite |v| of float columns, 1 for integers; 0 = unknown).  Literals and + − × only: a division has no useful
  // bound, and a sum of terms of either sign may cancel to anything (nz unknown unless the interval excludes zero).
  bool expr_bounds(const llkv_expr_token *e, uint32_t n, double *absmax, double *nzmin) {
    struct I { double lo, hi, nz; };
    std::vector<I> st;
    bounds_all_finite = true;
    for (uint32_t i = 0; i < n; ++i) {
      if (e[i].kind == LLKV_TOK_COLUMN) {
        const ColumnInfo *ci = resolve(e[i].field_id);
        if (!ci) return false;
        table_rows = std::max(table_rows, ci->rows);
        if ((ci->dtype == LLKV_DT_FLOAT64 || ci->dtype == LLKV_DT_FLOAT32) && ci->has_fstats) {
          st.push_back({-ci->f_absmax, ci->f_absmax, ci->f_absmin_nz});
          bounds_all_finite &= ci->f_all_finite;
        }
        else if (ci->dtype == LLKV_DT_BOOLEAN) st.push_back({0.0, 1.0, 1.0});
        else if (ci->dtype == LLKV_DT_UTF8) { // numeric image of the dictionary
          I b{0.0, 0.0, 0.0};
          for (const std::string &w : ci->dictionary) {
            const double v = parse_numeric_or_zero(w);
            if (!std::isfinite(v)) return false;
            b.lo = std::min(b.lo, v); b.hi = std::max(b.hi, v);
            if (v != 0.0) b.nz = b.nz == 0.0 ? std::fabs(v) : std::min(b.nz, std::fabs(v));
          }
          if (b.nz == 0.0) b.nz = 1.0; // every string counts as 0
          st.push_back(b);
        }
        else if (is_int_class(ci->dtype) && ci->has_stats) st.push_back({(double)ci->min_i, (double)ci->max_i, 1.0});
        else return false;
      } else if (e[i].kind == LLKV_TOK_LITERAL) {
        const llkv_literal &lit = e[i].literal;
        double v;
        if (lit.tag == LLKV_LIT_FLOAT64) v = lit.f64;
        else if (lit.tag == LLKV_LIT_INT128) v = (double)lit_i128(lit);
        else return false;
        if (!std::isfinite(v)) return false;
        st.push_back({v, v, std::fabs(v)});
      } else {
        if (st.size() < 2) return false;
        const I r = st.back(); st.pop_back();
        const I l = st.back(); st.pop_back();
        I o;
        switch (e[i].binop) {
        case LLKV_BIN_ADD: o = {l.lo + r.lo, l.hi + r.hi, 0.0}; break;
        case LLKV_BIN_SUB: o = {l.lo - r.hi, l.hi - r.lo, 0.0}; break;
        case LLKV_BIN_MUL: {
          const double c[4] = {l.lo * r.lo, l.lo * r.hi, l.hi * r.lo, l.hi * r.hi};
          o = {std::min(std::min(c[0], c[1]), std::min(c[2], c[3])), std::max(std::max(c[0], c[1]), std::max(c[2], c[3])), l.nz * r.nz * 0.999999};
          break;
        }
        default: return false;
        }
        if (!std::isfinite(o.lo) || !std::isfinite(o.hi)) return false;
        if (e[i].binop != LLKV_BIN_MUL) { // the rounded endpoints of a sum: move them outwards a little before trusting their sign
          const double slack = 1e-12 * std::max(std::fabs(o.lo), std::fabs(o.hi));
          if (o.lo - slack > 0.0) o.nz = o.lo - slack;
          else if (o.hi + slack < 0.0) o.nz = -(o.hi + slack);
        }
        st.push_back(o);
      }
    }
    if (st.size() != 1) return false;
    *absmax = std::max(std::fabs(st[0].lo), std::fabs(st[0].hi));
    *nzmin = st[0].nz;
    return true;
  }

  // SumF64X constants (fused_scan.hip.h) for an argument with |v| ≤ absmax, |v| ≥ nzmin where v ≠ 0, over a table of
  // `rows` rows: B = 2^b ≥ absmax, L = ⌈log2(rows + 1)⌉ ≥ 2, grids u1 = 2^(b+L−52), u(j+1) = u(j)·2^(L−53),
  // C(j) = 1.5·2^52·u(j).  As many levels (2 or 3) as it takes for the last grid to resolve the smallest non-zero
  // value to 2^-30 of itself: what a row drops is then ≤ 2^-31 of its own magnitude, so a group's sum is within
  // 5e-10 of Σ|v| — inside the contract whatever the group holds.  0 levels: no such choice (the caller's route).
  int exact_sum_constants(double absmax, double nzmin, uint64_t rows, double c[4]) {
    if (!(absmax >= 0.0) || !std::isfinite(absmax)) return 0;
    if (absmax == 0.0) { absmax = 1.0; nzmin = 1.0; } // the argument is always zero
    if (!(nzmin > 0.0)) return 0;
    absmax *= 1.0000001; // the kernel evaluates the argument in f64: every operation rounds, the interval endpoints did too
    int ex;
    const double m = std::frexp(absmax, &ex); // absmax = m·2^ex, m in [0.5, 1)
    const int b = m == 0.5 ? ex - 1 : ex;
    int L = 2;
    while (L < 63 && ((uint64_t)1 << L) < rows + 1) ++L;
    if (b < -900 || b + L > 1000 || L > 45) return 0;
    int nz_ex;
    (void)std::frexp(nzmin, &nz_ex); // nzmin ≥ 2^(nz_ex − 1)
    const int keep = strict_exact ? 52 : 30; // bits of the smallest non-zero value the last grid resolves (52: all of them)
    for (int levels = 2; levels <= (strict_exact ? 4 : 3); ++levels) {
      const int e_last = b + L - 52 + (levels - 1) * (L - 53); // log2 of the last grid
      if (e_last < -1000) return 0;
      if (e_last <= nz_ex - 1 - keep) {
        for (int j = 0; j < levels; ++j) c[j] = std::ldexp(1.5, b + L + j * (L - 53));
        return levels;
      }
    }
    return 0;
  }

  // SumF64Q (fused_scan.hip.h): the grid 2^e at 2^-30 of the smallest non-zero |v| (the last grid of SumF64X: the
  // same per-row rounding), if |v| / 2^e summed over the rows ONE workgroup image can see stays below 2^62 — a shared-
  // image scan runs at least 256 workgroups over tiles of ≤ 8 192 rows (engine.cpp: pick_image_grid honours
  // LoweredPlan::image_min_grid), so an image sees at most rows / 128 + 16 384 rows of the table.
  bool fixed_point_grid(double absmax, double nzmin, uint64_t rows, int *e_out) {
    if (!(absmax >= 0.0) || !std::isfinite(absmax)) return false;
    if (absmax == 0.0) { absmax = 1.0; nzmin = 1.0; }
    if (!(nzmin > 0.0)) return false;
    absmax *= 1.0000001;
    int ex, nz_ex;
    const double m = std::frexp(absmax, &ex);
    const int b = m == 0.5 ? ex - 1 : ex;
    (void)std::frexp(nzmin, &nz_ex);
    const int e = nz_ex - 1 - (strict_exact ? 52 : 30);
    const uint64_t image_rows = whole_table_image ? rows : rows / 128 + 16384; // (a partition's image may see every row)
    int lr = 1;
    while (lr < 63 && ((uint64_t)1 << lr) < image_rows) ++lr;
    if (b - e + lr > 61 || e < -900 || e > 900) return false;
    *e_out = e;
    return true;
  }
  // SumF64Q2 (register / per-thread-column plans under the exact-sum option): the row's value as an integer count of steps
  // 2^e, e = the ulp of the smallest non-zero |v| — exact, since every value is a multiple of it — split by the kernel
  // into its low 32 bits and the rest: two ADD_I64 lanes that cannot overflow below 2^31 rows.  Needs |v| / 2^e < 2^62.
  bool exact_fixed_point(double absmax, double nzmin, uint64_t rows, int *e_out) {
    if (!(absmax >= 0.0) || !std::isfinite(absmax) || rows >= (1ull << 31)) return false;
    if (absmax == 0.0) { absmax = 1.0; nzmin = 1.0; }
    if (!(nzmin > 0.0)) return false;
    absmax *= 1.0000001;
    int ex, nz_ex;
    const double m = std::frexp(absmax, &ex);
    const int b = m == 0.5 ? ex - 1 : ex;
    (void)std::frexp(nzmin, &nz_ex);
    const int e = nz_ex - 1 - 52;
    if (b - e > 62 || e < -900 || e > 900) return false;
    *e_out = e;
    return true;
  }

  int fail(int code, const std::string &m) { return set_err(err, code, m); }

  int slot_of(uint32_t field, const ColumnInfo **ci_out, int *slot) {
    const ColumnInfo *ci = resolve(field);
    if (!ci) return fail(LLKV_NOT_FOUND, "field " + std::to_string(field) + " not found");
    *ci_out = ci;
    // every reader of a column's values comes through here: a wide Decimal128 column has no 8 B/row value image
    if (ci->wide128)
      return fail(LLKV_UNSUPPORTED, "Decimal128 values beyond 64 bits in field " + std::to_string(field) + ": only SUM / TOTAL / AVG / COUNT over the bare column are on the GPU path");
    for (size_t i = 0; i < p.slot_fields.size(); ++i)
      if (p.slot_fields[i] == field && !p.slot_is_valid[i]) { *slot = (int)i; return LLKV_OK; }
    if ((int)p.slot_fields.size() >= kMaxColsHost) return fail(LLKV_UNSUPPORTED, "plan touches more than 16 column buffers");
    p.slot_fields.push_back(field);
    p.slot_dtypes.push_back(ci->dtype);
    p.slot_is_valid.push_back(0);
    *slot = (int)p.slot_fields.size() - 1;
    return LLKV_OK;
  }
  // The two buffers of a wide Decimal128 column: low halves (u64), high halves (i64).
  int wide_slots_of(uint32_t field, int *lo, int *hi, bool want_hi = true) {
    *lo = *hi = -1;
    for (size_t i = 0; i < p.slot_fields.size(); ++i) {
      if (p.slot_fields[i] != field) continue;
      if (p.slot_is_valid[i] == 0) *lo = (int)i;
      if (p.slot_is_valid[i] == 2) *hi = (int)i;
    }
    for (int part = 0; part < (want_hi ? 2 : 1); ++part) {
      int &slot = part ? *hi : *lo;
      if (slot >= 0) continue;
      if ((int)p.slot_fields.size() >= kMaxColsHost) return fail(LLKV_UNSUPPORTED, "plan touches more than 16 column buffers");
      p.slot_fields.push_back(field);
      p.slot_dtypes.push_back(part ? LLKV_DT_INT64 : LLKV_DT_UINT64);
      p.slot_is_valid.push_back(part ? 2 : 0);
      slot = (int)p.slot_fields.size() - 1;
    }
    return LLKV_OK;
  }
  // Validity of a field as a predicate node: "" when the column has no NULL cell, else `Valid<slot>` over the
  // field's 1 B/row validity mask (a NULL cell is a row id absent from the column, llkv-table/src/table.rs:1202-1223).
  int valid_of_field(uint32_t field, std::string *v) {
    v->clear();
    const ColumnInfo *ci = resolve(field);
    if (!ci) return fail(LLKV_NOT_FOUND, "field " + std::to_string(field) + " not found");
    if (!ci->nullable) return LLKV_OK;
    int slot = -1;
    for (size_t i = 0; i < p.slot_fields.size(); ++i)
      if (p.slot_fields[i] == field && p.slot_is_valid[i] == 1) slot = (int)i;
    if (slot < 0) {
      if ((int)p.slot_fields.size() >= kMaxColsHost) return fail(LLKV_UNSUPPORTED, "plan touches more than 16 column buffers");
      p.slot_fields.push_back(field);
      p.slot_dtypes.push_back(LLKV_DT_UTF8); // 1-byte cells
      p.slot_is_valid.push_back(1);
      slot = (int)p.slot_fields.size() - 1;
    }
    *v = "Valid<" + std::to_string(slot) + ">";
    return LLKV_OK;
  }
  static std::string all_of(const std::vector<std::string> &vs) { // conjunction of validity nodes; "" = always valid
    if (vs.empty()) return "";
    if (vs.size() == 1) return vs[0];
    std::string s = "And<";
    for (size_t i = 0; i < vs.size(); ++i) s += (i ? "," : "") + vs[i];
    return s + ">";
  }
  // NULL propagates through arithmetic: an expression is valid where all of its columns are.
  int valid_of_expr(const llkv_expr_token *e, uint32_t n, std::vector<std::string> *vs) {
    for (uint32_t i = 0; i < n; ++i) {
      if (e[i].kind != LLKV_TOK_COLUMN) continue;
      std::string v;
      int rc = valid_of_field(e[i].field_id, &v);
      if (rc) return rc;
      if (!v.empty() && std::find(vs->begin(), vs->end(), v) == vs->end()) vs->push_back(v);
    }
    return LLKV_OK;
  }
  std::string col_node(int slot, int32_t dtype) { return "Col<" + std::to_string(slot) + "," + dtype_tag(dtype) + ">"; }
  // Column as an operand of an expression: carries its validity when it has NULL cells (NULL propagates, and
  // a node raises arithmetic errors only where its operands are valid).
  int expr_col_node(uint32_t field, const ColumnInfo **ci, std::string *node) {
    int slot, rc;
    if ((rc = slot_of(field, ci, &slot))) return rc;
    std::string v;
    if ((rc = valid_of_field(field, &v))) return rc;
    if (v.empty()) *node = col_node(slot, (*ci)->dtype);
    else *node = "ColN<" + std::to_string(slot) + "," + dtype_tag((*ci)->dtype) + "," + v.substr(6, v.size() - 7) + ">"; // "Valid<k>" → k
    return LLKV_OK;
  }
  // Does the value of an expression depend on more than "all columns present"?  (NULL cells or divisions.)
  static bool has_division(const llkv_expr_token *e, uint32_t n) {
    for (uint32_t i = 0; i < n; ++i) if (e[i].kind == LLKV_TOK_BINARY && (e[i].binop == LLKV_BIN_DIV || e[i].binop == LLKV_BIN_MOD)) return true;
    return false;
  }
  // Validity of a lowered expression node as a predicate: "" when it can never be NULL.
  int valid_of_node(const llkv_expr_token *e, uint32_t n, const std::string &node, bool grouped_semantics, std::string *v) {
    std::vector<std::string> vs;
    int rc = valid_of_expr(e, n, &vs);
    if (rc) return rc;
    bool div = false;
    for (uint32_t i = 0; i < n; ++i)
      if (e[i].kind == LLKV_TOK_BINARY && (e[i].binop == LLKV_BIN_DIV || (grouped_semantics && e[i].binop == LLKV_BIN_MOD))) div = true; // x/0 (and PlanValue x%0) → NULL
    if (n == 1 && e[0].kind == LLKV_TOK_COLUMN) { *v = all_of(vs); return LLKV_OK; }
    *v = (vs.empty() && !div) ? std::string() : "VE<" + node + ">";
    return LLKV_OK;
  }

  // A literal takes a slot of the plan's bank; a value the bank already holds shares its slot (so the same expression under
  // two aggregates lowers to the same node, and the lane groups — deduplicated by node — are shared too)
  int lit_i(int64_t v, std::string *node, const char *kind = "LitI") {
    size_t at = 0;
    while (at < p.lit_i.size() && p.lit_i[at] != v) ++at;
    if (at == p.lit_i.size()) {
      if ((int)p.lit_i.size() >= kMaxLitsHost) return fail(LLKV_UNSUPPORTED, "too many integer literals");
      p.lit_i.push_back(v);
    }
    *node = std::string(kind) + "<" + std::to_string(at) + ">";
    return LLKV_OK;
  }
  int lit_f(double v, std::string *node) {
    size_t at = 0;
    while (at < p.lit_f.size() && std::memcmp(&p.lit_f[at], &v, 8) != 0) ++at; // by bit pattern: −0.0 and NaNs keep their own slots
    if (at == p.lit_f.size()) {
      if ((int)p.lit_f.size() >= kMaxLitsHost) return fail(LLKV_UNSUPPORTED, "too many float literals");
      p.lit_f.push_back(v);
    }
    *node = "LitF<" + std::to_string(at) + ">";
    return LLKV_OK;
  }

  int native_node(const llkv_literal &l, int32_t dtype, std::string *node) {
    NativeLit n;
    int rc = cast_literal_for_column(l, dtype, &n, err);
    if (rc) return rc;
    if (n.is_float) return lit_f(n.f, node);
    return lit_i(n.i, node, n.is_unsigned ? "LitU" : "LitI");
  }

  // One leaf → (matching rows, domain).  The domain is where the leaf is determined — the rows where the
  // field is present (DomainOp::PushFieldAll, llkv-compute/src/program.rs:466, llkv-scan/src/predicate.rs:665-777);
  // "True" = every row.  A NULL cell never matches (table.rs:1241-1244).
  int leaf(const llkv_filter &f, std::string *rows, std::string *dom) {
    if (f.op == LLKV_OP_COMPARE) return compare_leaf(f, rows, dom);
    if (f.op == LLKV_OP_IN_LIST) return in_list_leaf(f, rows, dom);
    if (f.op == LLKV_OP_IS_NULL_EXPR) return is_null_expr_leaf(f, rows, dom);
    std::string v, x;
    int rc = valid_of_field(f.field_id, &v);
    if (rc) return rc;
    *dom = v.empty() ? "True" : v;
    if (f.op == LLKV_OP_IS_NOT_NULL) { *rows = *dom; return LLKV_OK; } // row-universe algebra, table.rs:1128-1143
    if (f.op == LLKV_OP_IS_NULL) { *rows = v.empty() ? "False" : "Not<" + v + ">"; return LLKV_OK; }
    if ((rc = leaf_values(f, &x))) return rc;
    const bool all_rows = f.op == LLKV_OP_RANGE && f.lower_kind == LLKV_BOUND_UNBOUNDED && f.upper_kind == LLKV_BOUND_UNBOUNDED; // NULLs included, table.rs:1146-1153
    if (v.empty() || all_rows || x == "False") *rows = x;
    else *rows = x == "True" ? v : "And<" + v + "," + x + ">";
    return LLKV_OK;
  }

  // The value test of a leaf filter, NULL cells aside (llkv-table/src/table.rs:1117-1171).
  int leaf_values(const llkv_filter &f, std::string *out) {
    const ColumnInfo *ci = resolve(f.field_id);
    if (!ci) return fail(LLKV_NOT_FOUND, "field " + std::to_string(f.field_id) + " not found");
    if (f.op == LLKV_OP_MVCC_VISIBLE) { // MvccRowIdFilter as a leaf
      const ColumnInfo *cd = resolve((uint32_t)f.value.lo);
      if (!cd) return fail(LLKV_NOT_FOUND, "deleted_by field " + std::to_string((uint32_t)f.value.lo) + " not found");
      if (ci->dtype != LLKV_DT_UINT64 || cd->dtype != LLKV_DT_UINT64) return fail(LLKV_INVALID_ARGUMENT, "MVCC columns must be UInt64");
      // every non-committed id is one literal slot and two compares per row (created_by, deleted_by) against a scalar register;
      // beyond 32 of them a sorted array and a bisection per row would be the form (not built)
      if (f.in_len > 32) return fail(LLKV_UNSUPPORTED, "more than 32 non-committed transactions in the snapshot");
      int sc, sd, rc2;
      const ColumnInfo *tmp;
      if ((rc2 = slot_of(f.field_id, &tmp, &sc)) || (rc2 = slot_of((uint32_t)f.value.lo, &tmp, &sd))) return rc2;
      std::string txn, snap, un;
      if ((rc2 = lit_i((int64_t)f.lower.lo, &txn, "LitU")) || (rc2 = lit_i((int64_t)f.upper.lo, &snap, "LitU"))) return rc2;
      for (uint32_t i = 0; i < f.in_len; ++i) {
        std::string u;
        if ((rc2 = lit_i((int64_t)f.in_list[i].lo, &u, "LitU"))) return rc2;
        un += "," + u;
      }
      *out = "Mvcc<" + col_node(sc, LLKV_DT_UINT64) + "," + col_node(sd, LLKV_DT_UINT64) + "," + txn + "," + snap + un + ">";
      return LLKV_OK;
    }
    if (f.op == LLKV_OP_RANGE && f.lower_kind == LLKV_BOUND_UNBOUNDED && f.upper_kind == LLKV_BOUND_UNBOUNDED) { *out = "True"; return LLKV_OK; }
    int slot;
    int rc;
    if (ci->dtype == LLKV_DT_UTF8) { // dictionary codes: equality only
      auto code_of = [&](const llkv_literal &l, int *code) -> int {
        if (l.tag != LLKV_LIT_STRING || !l.str) return fail(LLKV_PREDICATE_BUILD, std::string("literal cast error: expected string, got ") + lit_kind(l));
        *code = -1;
        for (size_t i = 0; i < ci->dictionary.size(); ++i) if (ci->dictionary[i] == l.str) *code = (int)i;
        return LLKV_OK;
      };
      if (f.op == LLKV_OP_EQUALS) {
        int code;
        if ((rc = code_of(f.value, &code))) return rc;
        if (code < 0) { *out = "False"; return LLKV_OK; }
        if ((rc = slot_of(f.field_id, &ci, &slot))) return rc;
        std::string lit;
        if ((rc = lit_i(code, &lit))) return rc;
        *out = "Eq<" + col_node(slot, ci->dtype) + "," + lit + ">";
        return LLKV_OK;
      }
      if (f.op == LLKV_OP_IN) {
        std::string lits;
        for (uint32_t i = 0; i < f.in_len; ++i) {
          int code;
          if ((rc = code_of(f.in_list[i], &code))) return rc;
          if (code < 0) continue;
          std::string lit;
          if ((rc = lit_i(code, &lit))) return rc;
          lits += "," + lit;
        }
        if (lits.empty()) { *out = "False"; return LLKV_OK; }
        if ((rc = slot_of(f.field_id, &ci, &slot))) return rc;
        *out = "In<" + col_node(slot, ci->dtype) + lits + ">";
        return LLKV_OK;
      }
      // ordering predicates: evaluate once per dictionary string (str::cmp = byte order), ship the set of codes
      auto lit_str = [&](const llkv_literal &l, std::string *out_s) -> int {
        if (l.tag != LLKV_LIT_STRING || !l.str) return fail(LLKV_PREDICATE_BUILD, std::string("literal cast error: expected string, got ") + lit_kind(l));
        *out_s = l.str;
        return LLKV_OK;
      };
      std::string lo_s, hi_s, pat;
      int lo_k = LLKV_BOUND_UNBOUNDED, hi_k = LLKV_BOUND_UNBOUNDED;
      const bool pattern_op = f.op == LLKV_OP_STARTS_WITH || f.op == LLKV_OP_ENDS_WITH || f.op == LLKV_OP_CONTAINS;
      auto ascii = [](const std::string &x) { for (unsigned char ch : x) if (ch >= 0x80) return false; return true; };
      auto lower = [](std::string x) { for (char &ch : x) if (ch >= 'A' && ch <= 'Z') ch = (char)(ch + 32); return x; };
      if (pattern_op) { // Operator::{StartsWith, EndsWith, Contains} (typed_predicate.rs:186-210)
        if ((rc = lit_str(f.value, &pat))) return rc;
        if (!f.case_sensitive) {
          if (!ascii(pat)) return fail(LLKV_UNSUPPORTED, "case-insensitive pattern with non-ASCII characters (Unicode to_lowercase)");
          pat = lower(pat);
        }
      }
      switch (pattern_op ? LLKV_OP_RANGE + 1000 : f.op) {
      case LLKV_OP_RANGE + 1000: break;
      case LLKV_OP_GT: lo_k = LLKV_BOUND_EXCLUDED; if ((rc = lit_str(f.value, &lo_s))) return rc; break;
      case LLKV_OP_GE: lo_k = LLKV_BOUND_INCLUDED; if ((rc = lit_str(f.value, &lo_s))) return rc; break;
      case LLKV_OP_LT: hi_k = LLKV_BOUND_EXCLUDED; if ((rc = lit_str(f.value, &hi_s))) return rc; break;
      case LLKV_OP_LE: hi_k = LLKV_BOUND_INCLUDED; if ((rc = lit_str(f.value, &hi_s))) return rc; break;
      case LLKV_OP_RANGE:
        lo_k = f.lower_kind; hi_k = f.upper_kind;
        if (lo_k != LLKV_BOUND_UNBOUNDED && (rc = lit_str(f.lower, &lo_s))) return rc;
        if (hi_k != LLKV_BOUND_UNBOUNDED && (rc = lit_str(f.upper, &hi_s))) return rc;
        break;
      default: return fail(LLKV_PREDICATE_BUILD, "unsupported operator for typed predicate: operator lacks string literal support");
      }
      uint64_t mask[4] = {0, 0, 0, 0};
      bool any = false;
      for (size_t i = 0; i < ci->dictionary.size() && i < 256; ++i) {
        const std::string &v = ci->dictionary[i];
        bool ok = true;
        if (pattern_op) {
          if (!f.case_sensitive && !ascii(v)) return fail(LLKV_UNSUPPORTED, "case-insensitive pattern over non-ASCII strings (Unicode to_lowercase)");
          const std::string x = f.case_sensitive ? v : lower(v);
          if (f.op == LLKV_OP_STARTS_WITH) ok = x.size() >= pat.size() && x.compare(0, pat.size(), pat) == 0;
          else if (f.op == LLKV_OP_ENDS_WITH) ok = x.size() >= pat.size() && x.compare(x.size() - pat.size(), pat.size(), pat) == 0;
          else ok = x.find(pat) != std::string::npos;
        }
        if (lo_k == LLKV_BOUND_INCLUDED) ok = ok && v.compare(lo_s) >= 0;
        if (lo_k == LLKV_BOUND_EXCLUDED) ok = ok && v.compare(lo_s) > 0;
        if (hi_k == LLKV_BOUND_INCLUDED) ok = ok && v.compare(hi_s) <= 0;
        if (hi_k == LLKV_BOUND_EXCLUDED) ok = ok && v.compare(hi_s) < 0;
        if (ok) { mask[i >> 6] |= 1ull << (i & 63); any = true; }
      }
      if (!any) { *out = "False"; return LLKV_OK; }
      if ((rc = slot_of(f.field_id, &ci, &slot))) return rc;
      std::string m[4];
      for (int w = 0; w < 4; ++w) if ((rc = lit_i((int64_t)mask[w], &m[w], "LitU"))) return rc;
      *out = "InMask<" + col_node(slot, ci->dtype) + "," + m[0] + "," + m[1] + "," + m[2] + "," + m[3] + ">";
      return LLKV_OK;
    }
    if (ci->dtype == LLKV_DT_DECIMAL128) // llkv-table/src/table.rs:1160-1167
      return fail(LLKV_INTERNAL, "Filtering on type Decimal128(" + std::to_string(ci->precision) + ", " + std::to_string(ci->scale) + ") is not supported");
    if (dtype_width(ci->dtype) == 0 || ci->dtype == LLKV_DT_BOOLEAN)
      return fail(LLKV_INTERNAL, std::string("Filtering on type ") + dtype_name(ci->dtype) + " is not supported");
    // cast every literal first: cast errors surface even when the column needs no slot yet
    std::string a, lo, hi;
    switch (f.op) {
    case LLKV_OP_EQUALS: case LLKV_OP_GT: case LLKV_OP_GE: case LLKV_OP_LT: case LLKV_OP_LE:
      if ((rc = native_node(f.value, ci->dtype, &a))) return rc;
      break;
    case LLKV_OP_RANGE:
      if (f.lower_kind != LLKV_BOUND_UNBOUNDED && (rc = native_node(f.lower, ci->dtype, &lo))) return rc;
      if (f.upper_kind != LLKV_BOUND_UNBOUNDED && (rc = native_node(f.upper, ci->dtype, &hi))) return rc;
      break;
    case LLKV_OP_IN:
      for (uint32_t i = 0; i < f.in_len; ++i) {
        std::string n;
        if ((rc = native_node(f.in_list[i], ci->dtype, &n))) return rc;
        a += "," + n;
      }
      break;
    default:
      return fail(LLKV_PREDICATE_BUILD, "unsupported operator for typed predicate: operator lacks typed literal support");
    }
    if ((rc = slot_of(f.field_id, &ci, &slot))) return rc;
    const std::string col = col_node(slot, ci->dtype);
    switch (f.op) {
    case LLKV_OP_EQUALS: *out = "Eq<" + col + "," + a + ">"; break;
    case LLKV_OP_GT: *out = "Range<" + col + ",2," + a + ",0,Nil>"; break;
    case LLKV_OP_GE: *out = "Range<" + col + ",1," + a + ",0,Nil>"; break;
    case LLKV_OP_LT: *out = "Range<" + col + ",0,Nil,2," + a + ">"; break;
    case LLKV_OP_LE: *out = "Range<" + col + ",0,Nil,1," + a + ">"; break;
    case LLKV_OP_RANGE:
      *out = "Range<" + col + "," + std::to_string(f.lower_kind) + "," + (lo.empty() ? "Nil" : lo) + "," +
             std::to_string(f.upper_kind) + "," + (hi.empty() ? "Nil" : hi) + ">";
      break;
    case LLKV_OP_IN:
      *out = a.empty() ? std::string("False") : "In<" + col + a + ">";
      break;
    }
    return LLKV_OK;
  }

  // Predicate program (llkv-compute/src/program.rs:48-78) → nested node.  Every stack entry carries the rows
  // it matches and its domain, as the reference's DomainProgram does (program.rs:447-520): And → intersect,
  // Or → union, Not → domain(child) − rows(child) with the domain unchanged (predicate.rs:167-186).
  struct PredEntry { std::string rows, dom; };
  static std::string nary(const char *op, const std::vector<std::string> &xs) {
    std::string s = std::string(op) + "<";
    for (size_t i = 0; i < xs.size(); ++i) s += (i ? "," : "") + xs[i];
    return s + ">";
  }
  int predicate(const llkv_filter *filters, uint32_t n_filters, const llkv_eval_op *ops, uint32_t n_ops, std::string *out) {
    int rc;
    if (n_ops == 0) {
      if (n_filters == 0) { *out = "True"; return LLKV_OK; }
      std::vector<std::string> ls;
      for (uint32_t i = 0; i < n_filters; ++i) {
        std::string l, d;
        if ((rc = leaf(filters[i], &l, &d))) return rc;
        ls.push_back(l);
      }
      *out = nary("And", ls);
      return LLKV_OK;
    }
    std::vector<PredEntry> st;
    for (uint32_t k = 0; k < n_ops; ++k) {
      switch (ops[k].op) {
      case LLKV_EVAL_PUSH_PREDICATE: {
        if (ops[k].arg >= n_filters) return fail(LLKV_INTERNAL, "predicate index out of range");
        PredEntry e;
        if ((rc = leaf(filters[ops[k].arg], &e.rows, &e.dom))) return rc;
        st.push_back(e);
        break;
      }
      case LLKV_EVAL_PUSH_LITERAL: st.push_back({ops[k].arg ? "True" : "False", "True"}); break;
      case LLKV_EVAL_AND: case LLKV_EVAL_OR: {
        if (ops[k].arg == 0 || ops[k].arg > st.size()) return fail(LLKV_INTERNAL, "predicate stack underflow");
        const bool is_and = ops[k].op == LLKV_EVAL_AND;
        const size_t b = st.size() - ops[k].arg;
        std::vector<std::string> rows, doms;
        bool dom_all = false;
        for (size_t i = b; i < st.size(); ++i) {
          rows.push_back(st[i].rows);
          if (st[i].dom == "True") dom_all = true; // neutral for the intersection, absorbing for the union
          else if (std::find(doms.begin(), doms.end(), st[i].dom) == doms.end()) doms.push_back(st[i].dom);
        }
        PredEntry e;
        e.rows = nary(is_and ? "And" : "Or", rows);
        if (is_and) e.dom = doms.empty() ? "True" : doms.size() == 1 ? doms[0] : nary("And", doms);
        else e.dom = (dom_all || doms.empty()) ? "True" : doms.size() == 1 ? doms[0] : nary("Or", doms);
        st.resize(b);
        st.push_back(e);
        break;
      }
      case LLKV_EVAL_NOT:
        if (st.empty()) return fail(LLKV_INTERNAL, "predicate stack underflow");
        if (st.back().dom == "False") { if (st.back().rows == "True") st.back().rows = "False"; } // determined nowhere: NOT of it holds nowhere — and neither does it (matched ⊆ determined), so it stays as it is: what it evaluates (ErrOnly) still counts
        else if (st.back().rows == "False") st.back().rows = st.back().dom; // NOT FALSE = where it is determined
        else if (st.back().rows == "True" && st.back().dom == "True") st.back().rows = "False";
        else st.back().rows = st.back().dom == "True" ? "Not<" + st.back().rows + ">" : "And<" + st.back().dom + ",Not<" + st.back().rows + ">>";
        break;
      default: return fail(LLKV_INTERNAL, "unknown predicate opcode");
      }
    }
    if (st.size() != 1) return fail(LLKV_INTERNAL, "predicate program left " + std::to_string(st.size()) + " entries");
    *out = st[0].rows;
    return LLKV_OK;
  }

  // ---- Expr::Compare (collect_row_ids_for_compare, llkv-scan/src/predicate.rs:333-396) ----
  // Arrow type class of one side: a bare column keeps its own type, an integer literal is Int64
  // (literal_type, llkv-compute/src/eval.rs:167-185), a computed side has the fast path's result type.
  enum class Side { S32, U32, S64, U64, F };
  int expr_side(const llkv_expr_token *e, uint32_t n, std::string *node, Side *cls) {
    struct Exact { bool &f, was; explicit Exact(bool &x) : f(x), was(x) { f = true; } ~Exact() { f = was; } } exact(exact_nan); // compared by totalOrder
    for (uint32_t i = 0; i < n; ++i)
      if (e[i].kind == LLKV_TOK_LITERAL && e[i].literal.tag == LLKV_LIT_NULL) return fail(LLKV_UNSUPPORTED, "NULL literal in a comparison");
    if (n == 1 && e[0].kind == LLKV_TOK_COLUMN) {
      const ColumnInfo *ci;
      int slot, rc;
      if ((rc = slot_of(e[0].field_id, &ci, &slot))) return rc;
      const std::string c = col_node(slot, ci->dtype);
      switch (ci->dtype) {
      case LLKV_DT_FLOAT64: *node = c; *cls = Side::F; break;
      case LLKV_DT_FLOAT32: *node = "ToF64<" + c + ">"; *cls = Side::F; break; // totalOrder survives the exact widening
      case LLKV_DT_INT64: *node = c; *cls = Side::S64; break;
      case LLKV_DT_UINT64: *node = c; *cls = Side::U64; break;
      case LLKV_DT_INT32: *node = "ToI64<" + c + ">"; *cls = Side::S32; break;
      case LLKV_DT_UINT32: *node = "ToI64<" + c + ">"; *cls = Side::U32; break;
      default: return fail(LLKV_UNSUPPORTED, std::string("comparison over ") + dtype_name(ci->dtype));
      }
      return LLKV_OK;
    }
    if (n == 1 && e[0].kind == LLKV_TOK_LITERAL) {
      const llkv_literal &lit = e[0].literal;
      if (lit.tag == LLKV_LIT_FLOAT64) { *cls = Side::F; return lit_f(lit.f64, node); }
      if (lit.tag != LLKV_LIT_INT128) return fail(LLKV_UNSUPPORTED, "non-numeric literal in a comparison");
      const __int128 v = lit_i128(lit);
      if (v < (__int128)INT64_MIN || v > (__int128)INT64_MAX) return fail(LLKV_UNSUPPORTED, "integer literal beyond Int64 in a comparison");
      *cls = Side::S64;
      return lit_i((int64_t)v, node);
    }
    bool is_f64 = false;
    const int rc = expr_fast(e, n, node, &is_f64);
    *cls = is_f64 ? Side::F : last_fast_32 == 1 ? Side::S32 : last_fast_32 == 2 ? Side::U32 : Side::S64;
    return rc;
  }

  // get_common_type (llkv-compute/src/kernels.rs:179-242) over the side classes
  static Side common_side(Side a, Side b) {
    if (a == b) return a;
    if (a == Side::F || b == Side::F) return Side::F;
    const auto uns = [](Side s) { return s == Side::U32 || s == Side::U64; };
    const auto wide = [](Side s) { return s == Side::S64 || s == Side::U64; };
    if (uns(a) && uns(b)) return Side::U64;
    if (!uns(a) && !uns(b)) return Side::S64;
    return (wide(a) || wide(b)) ? Side::F : Side::S64; // signed ⋈ unsigned, 64 bits wide → Float64
  }
  // arrow `cast` of a side node to the common type (the integer classes share an i64 / u64 image already)
  static std::string cast_side(const std::string &node, Side from, Side to) { return (to == Side::F && from != Side::F) ? "ToF64<" + node + ">" : node; }

  // A side without a column, as evaluate_value leaves it: one numeric or NULL literal (what `simplify` folds literal arithmetic to)
  int constant_side(const llkv_expr_token *e, uint32_t n, llkv_literal *out) {
    std::vector<llkv_expr_token> fl;
    int rc = fold_constants(e, n, &fl);
    if (rc) return rc;
    if (fl.size() != 1 || fl[0].kind != LLKV_TOK_LITERAL || (fl[0].literal.tag != LLKV_LIT_INT128 && fl[0].literal.tag != LLKV_LIT_FLOAT64 && fl[0].literal.tag != LLKV_LIT_NULL))
      return fail(LLKV_UNSUPPORTED, "constant predicate over a side that does not fold to a numeric literal");
    *out = fl[0].literal;
    return LLKV_OK;
  }
  static bool constants_equal(const llkv_literal &a, const llkv_literal &b) { // arrow `eq` in the common type (floats by totalOrder)
    if (a.tag == LLKV_LIT_FLOAT64 || b.tag == LLKV_LIT_FLOAT64) {
      const double x = a.tag == LLKV_LIT_FLOAT64 ? a.f64 : (double)(int64_t)lit_i128(a), y = b.tag == LLKV_LIT_FLOAT64 ? b.f64 : (double)(int64_t)lit_i128(b);
      return std::memcmp(&x, &y, 8) == 0;
    }
    return (int64_t)lit_i128(a) == (int64_t)lit_i128(b);
  }

  // Expr::InList (evaluate_in_list_over_rows, llkv-scan/src/predicate.rs:443-560): the target is coerced item by
  // item — its type can only widen along the list — and compared with `eq`; over rows where all fields are present
  // nothing is NULL, so or_kleene / not are plain OR / NOT.
  int in_list_leaf(const llkv_filter &f, std::string *out, std::string *dom) {
    if (!f.cmp_left || !f.cmp_left_len) return fail(LLKV_INVALID_ARGUMENT, "IN list needs a target expression");
    if (f.list_len && (!f.list_exprs || !f.list_expr_lens)) return fail(LLKV_INVALID_ARGUMENT, "IN list arrays are NULL");
    bool any_col = false, div = has_division(f.cmp_left, f.cmp_left_len);
    for (uint32_t i = 0; i < f.cmp_left_len; ++i) any_col |= f.cmp_left[i].kind == LLKV_TOK_COLUMN;
    for (uint32_t k = 0; k < f.list_len; ++k) {
      div |= has_division(f.list_exprs[k], f.list_expr_lens[k]);
      for (uint32_t i = 0; i < f.list_expr_lens[k]; ++i) any_col |= f.list_exprs[k][i].kind == LLKV_TOK_COLUMN;
    }
    if (!any_col) {
      // evaluate_constant_in_list (:909-963): a NULL target matches and determines nothing; a matching item decides (→ !negated);
      // else a NULL item leaves it NULL; else `negated`.  TRUE keeps every row, FALSE none, both determined everywhere.
      llkv_literal tgt, item;
      int rc;
      if ((rc = constant_side(f.cmp_left, f.cmp_left_len, &tgt))) return rc;
      if (tgt.tag == LLKV_LIT_NULL) { *out = "False"; *dom = "False"; return LLKV_OK; }
      bool matched = false, saw_null = false;
      for (uint32_t k = 0; k < f.list_len && !matched; ++k) {
        if ((rc = constant_side(f.list_exprs[k], f.list_expr_lens[k], &item))) return rc;
        if (item.tag == LLKV_LIT_NULL) { saw_null = true; continue; }
        matched = constants_equal(tgt, item);
      }
      if (!matched && saw_null) { *out = "False"; *dom = "False"; return LLKV_OK; }
      *out = (matched ? !f.negated : (bool)f.negated) ? "True" : "False";
      *dom = "True";
      return LLKV_OK;
    }
    if (div) return fail(LLKV_UNSUPPORTED, "division inside an IN list (three-valued OR over NULL items)");
    std::string tn;
    Side tc;
    int rc;
    if ((rc = expr_side(f.cmp_left, f.cmp_left_len, &tn, &tc))) return rc;
    std::vector<std::string> vs, eqs;
    if ((rc = valid_of_expr(f.cmp_left, f.cmp_left_len, &vs))) return rc;
    for (uint32_t k = 0; k < f.list_len; ++k) {
      std::string in;
      Side ic;
      if ((rc = expr_side(f.list_exprs[k], f.list_expr_lens[k], &in, &ic)) || (rc = valid_of_expr(f.list_exprs[k], f.list_expr_lens[k], &vs))) return rc;
      const Side t = common_side(tc, ic);
      tn = cast_side(tn, tc, t); // the coerced target replaces the target (`target_array = new_target`)
      tc = t;
      eqs.push_back("Cmp<1," + cast_side(in, ic, t) + "," + tn + ">");
    }
    const std::string v = all_of(vs);
    *dom = v.empty() ? "True" : v;
    std::string hit = eqs.empty() ? "False" : eqs.size() == 1 ? eqs[0] : nary("Or", eqs);
    if (f.negated) hit = hit == "False" ? "True" : "Not<" + hit + ">";
    if (v.empty()) *out = hit;
    else *out = hit == "True" ? v : hit == "False" ? "False" : "And<" + v + "," + hit + ">";
    return LLKV_OK;
  }

  // Expr::IsNull over a scalar expression (collect_row_ids_for_is_null, predicate.rs:249-331).
  int is_null_expr_leaf(const llkv_filter &f, std::string *out, std::string *dom) {
    if (!f.cmp_left || !f.cmp_left_len) return fail(LLKV_INVALID_ARGUMENT, "IS NULL needs an expression");
    if (f.cmp_left_len == 1 && f.cmp_left[0].kind == LLKV_TOK_COLUMN) { // bare column: the leaf filter
      llkv_filter leaf_f{};
      leaf_f.field_id = f.cmp_left[0].field_id;
      leaf_f.op = f.negated ? LLKV_OP_IS_NOT_NULL : LLKV_OP_IS_NULL;
      return leaf(leaf_f, out, dom);
    }
    std::vector<std::string> vs;
    bool some_never_null = false, any_col = false;
    int rc;
    for (uint32_t i = 0; i < f.cmp_left_len; ++i) {
      if (f.cmp_left[i].kind != LLKV_TOK_COLUMN) continue;
      any_col = true;
      std::string v;
      if ((rc = valid_of_field(f.cmp_left[i].field_id, &v))) return rc;
      if (v.empty()) some_never_null = true;
      else if (std::find(vs.begin(), vs.end(), v) == vs.end()) vs.push_back(v);
    }
    if (!any_col) { // (:276-284) the value is NULL or it is not; every row of the table or none, determined everywhere (:746-749)
      llkv_literal v;
      if ((rc = constant_side(f.cmp_left, f.cmp_left_len, &v))) return rc;
      *out = ((v.tag == LLKV_LIT_NULL) != (bool)f.negated) ? "True" : "False";
      *dom = "True";
      return LLKV_OK;
    }
    std::string node;
    bool is_f64 = false;
    if ((rc = expr_fast(f.cmp_left, f.cmp_left_len, &node, &is_f64))) return rc;
    std::string ve;
    if ((rc = valid_of_node(f.cmp_left, f.cmp_left_len, node, false, &ve))) return rc;
    // the scanned rows are those where AT LEAST ONE referenced field is present (:286-291)
    const std::string uni = some_never_null ? "" : (vs.size() == 1 ? vs[0] : nary("Or", vs));
    std::string hit = ve.empty() ? (f.negated ? "True" : "False") : (f.negated ? ve : "Not<" + ve + ">");
    if (uni.empty() || hit == "False") *out = hit;
    else *out = hit == "True" ? uni : "And<" + uni + "," + hit + ">";
    *dom = vs.empty() ? "True" : all_of(vs); // PushIsNullDomain: rows where every field is present (:744-767)
    return LLKV_OK;
  }

  int compare_leaf(const llkv_filter &f, std::string *out, std::string *dom) {
    if (f.cmp_op < LLKV_CMP_EQ || f.cmp_op > LLKV_CMP_GT_EQ) return fail(LLKV_INVALID_ARGUMENT, "unknown compare operator");
    if (!f.cmp_left || !f.cmp_right || !f.cmp_left_len || !f.cmp_right_len) return fail(LLKV_INVALID_ARGUMENT, "compare needs two expressions");
    const llkv_expr_token *l = f.cmp_left, *r = f.cmp_right;
    // simple_compare_filter (predicate.rs:970-1010): column ⋈ non-NULL literal (not <>) is a leaf filter
    // with the leaf's typed-literal semantics; literal ⋈ column flips the operator.
    const bool l_col = f.cmp_left_len == 1 && l[0].kind == LLKV_TOK_COLUMN, r_col = f.cmp_right_len == 1 && r[0].kind == LLKV_TOK_COLUMN;
    const bool l_lit = f.cmp_left_len == 1 && l[0].kind == LLKV_TOK_LITERAL, r_lit = f.cmp_right_len == 1 && r[0].kind == LLKV_TOK_LITERAL;
    if (f.cmp_op != LLKV_CMP_NOT_EQ && ((l_col && r_lit) || (l_lit && r_col))) {
      const llkv_literal &lit = l_col ? r[0].literal : l[0].literal;
      if (lit.tag != LLKV_LIT_NULL) {
        static const int32_t direct[7] = {0, LLKV_OP_EQUALS, 0, LLKV_OP_LT, LLKV_OP_LE, LLKV_OP_GT, LLKV_OP_GE};
        static const int32_t flipped[7] = {0, LLKV_OP_EQUALS, 0, LLKV_OP_GT, LLKV_OP_GE, LLKV_OP_LT, LLKV_OP_LE};
        llkv_filter leaf_f{};
        leaf_f.field_id = l_col ? l[0].field_id : r[0].field_id;
        leaf_f.op = l_col ? direct[f.cmp_op] : flipped[f.cmp_op];
        leaf_f.value = lit;
        return leaf(leaf_f, out, dom);
      }
    }
    bool any_col = false;
    for (uint32_t i = 0; i < f.cmp_left_len; ++i) any_col |= l[i].kind == LLKV_TOK_COLUMN;
    for (uint32_t i = 0; i < f.cmp_right_len; ++i) any_col |= r[i].kind == LLKV_TOK_COLUMN;
    int rc;
    if (!any_col) {
      // no field at all (predicate.rs:354-360, :791-796): evaluate_constant_compare — both sides are evaluated once, compared in their
      // common type; TRUE selects every row of the table, FALSE none (both are "determined" everywhere), NULL selects and determines
      // nothing.  Restated for sides that fold to one numeric or NULL literal (what `simplify` leaves of literal arithmetic).
      std::vector<llkv_expr_token> fl, fr;
      if ((rc = fold_constants(l, f.cmp_left_len, &fl)) || (rc = fold_constants(r, f.cmp_right_len, &fr))) return rc;
      auto one_lit = [](const std::vector<llkv_expr_token> &v) {
        return v.size() == 1 && v[0].kind == LLKV_TOK_LITERAL && (v[0].literal.tag == LLKV_LIT_INT128 || v[0].literal.tag == LLKV_LIT_FLOAT64 || v[0].literal.tag == LLKV_LIT_NULL);
      };
      if (!one_lit(fl) || !one_lit(fr)) return fail(LLKV_UNSUPPORTED, "constant comparison over sides that do not fold to a numeric literal");
      const llkv_literal &a = fl[0].literal, &b = fr[0].literal;
      if (a.tag == LLKV_LIT_NULL || b.tag == LLKV_LIT_NULL) { *out = "False"; *dom = "False"; return LLKV_OK; }
      bool m;
      auto rel = [&](auto x, auto y) { return f.cmp_op == LLKV_CMP_EQ ? x == y : f.cmp_op == LLKV_CMP_NOT_EQ ? x != y : f.cmp_op == LLKV_CMP_LT ? x < y : f.cmp_op == LLKV_CMP_LT_EQ ? x <= y : f.cmp_op == LLKV_CMP_GT ? x > y : x >= y; };
      if (a.tag == LLKV_LIT_FLOAT64 || b.tag == LLKV_LIT_FLOAT64) { // Float64 by totalOrder
        auto key = [](double v) { int64_t k; std::memcpy(&k, &v, 8); return k ^ (int64_t)((uint64_t)(k >> 63) >> 1); };
        const double x = a.tag == LLKV_LIT_FLOAT64 ? a.f64 : (double)(int64_t)lit_i128(a), y = b.tag == LLKV_LIT_FLOAT64 ? b.f64 : (double)(int64_t)lit_i128(b);
        m = rel(key(x), key(y));
      } else m = rel((int64_t)lit_i128(a), (int64_t)lit_i128(b)); // (literal_to_array: `*v as i64`)
      *out = m ? "True" : "False";
      *dom = "True";
      return LLKV_OK;
    }
    {
      // one side is the NULL literal itself: the compare is NULL on every row — nothing matches, nothing is determined — but the other
      // side is evaluated over the rows where its fields are present, and its checked arithmetic can fail the scan
      const bool l_null = f.cmp_left_len == 1 && l[0].kind == LLKV_TOK_LITERAL && l[0].literal.tag == LLKV_LIT_NULL;
      const bool r_null = f.cmp_right_len == 1 && r[0].kind == LLKV_TOK_LITERAL && r[0].literal.tag == LLKV_LIT_NULL;
      if (l_null || r_null) {
        const llkv_expr_token *e = l_null ? r : l;
        const uint32_t en = l_null ? f.cmp_right_len : f.cmp_left_len;
        *dom = "False";
        if (en == 1) { *out = "False"; return LLKV_OK; } // (a bare column: nothing to evaluate)
        std::string node;
        Side cls;
        if ((rc = expr_side(e, en, &node, &cls))) return rc;
        std::vector<std::string> vs;
        if ((rc = valid_of_expr(e, en, &vs))) return rc;
        if (has_division(e, en)) vs.push_back("VE<" + node + ">");
        const std::string v = all_of(vs);
        *out = "ErrOnly<" + node + (v.empty() ? "" : "," + v) + ">";
        return LLKV_OK;
      }
    }
    std::string ln, rn;
    Side lc, rc_;
    if ((rc = expr_side(l, f.cmp_left_len, &ln, &lc)) || (rc = expr_side(r, f.cmp_right_len, &rn, &rc_))) return rc;
    // get_common_type (llkv-compute/src/kernels.rs:179-242) of the two sides
    const auto is_unsigned = [](Side s) { return s == Side::U32 || s == Side::U64; };
    const auto is_64 = [](Side s) { return s == Side::S64 || s == Side::U64; };
    bool as_float = lc == Side::F || rc_ == Side::F;
    if (!as_float && is_unsigned(lc) != is_unsigned(rc_) && (is_64(lc) || is_64(rc_))) as_float = true; // signed ⋈ unsigned, 64 bits wide → Float64
    if (as_float) {
      if (lc != Side::F) ln = "ToF64<" + ln + ">";
      if (rc_ != Side::F) rn = "ToF64<" + rn + ">";
    }
    // otherwise both signed (Int32/Int64 → i64 compare), both unsigned (→ u64 compare; a widened UInt32
    // is a non-negative i64, which C++ converts to u64 unchanged) or 32-bit mixed (→ Int64)
    // rows where every referenced field is present = where both sides are determined (predicate.rs:366-388,635-638)
    std::vector<std::string> vs;
    if ((rc = valid_of_expr(l, f.cmp_left_len, &vs)) || (rc = valid_of_expr(r, f.cmp_right_len, &vs))) return rc;
    if (has_division(l, f.cmp_left_len)) vs.push_back("VE<" + ln + ">");  // a NULL side (x / 0) leaves the row undetermined
    if (has_division(r, f.cmp_right_len)) vs.push_back("VE<" + rn + ">");
    const std::string v = all_of(vs);
    *dom = v.empty() ? "True" : v;
    *out = "Cmp<" + std::to_string(f.cmp_op) + "," + ln + "," + rn + (v.empty() ? "" : "," + v) + ">";
    return LLKV_OK;
  }

  static bool is_int_class(int32_t dt) { return dt == LLKV_DT_INT64 || dt == LLKV_DT_INT32 || dt == LLKV_DT_UINT32; }
  static bool is_float_class(int32_t dt) { return dt == LLKV_DT_FLOAT64 || dt == LLKV_DT_FLOAT32; }

  // Computed projection, fast numeric path: final type first, then every column cast to
  // it and every literal broadcast in it (llkv-compute/src/fast_numeric.rs:69-121).
  // ScalarEvaluator::simplify (llkv-compute/src/eval.rs:761-791), which the scan applies to every computed projection
  // before anything looks at it (llkv-scan/src/execute.rs:91): literal ⊕ literal folds bottom-up through
  // fold_binary_literals (:1010-1031) = compute_binary over two one-element arrays (Int128 literal → Int64 by `as i64`,
  // Float64 → Float64; kernels.rs:99-177) — checked integer arithmetic, a zero divisor nullified first (so x / 0 is the
  // NULL literal; x / −0.0 is not, the float compare is totalOrder), IEEE floats, a NULL side → NULL.  A fold that
  // errors (integer overflow, x % 0) leaves the node as it was: such a plan is handed back.
  int fold_constants(const llkv_expr_token *e, uint32_t n, std::vector<llkv_expr_token> *out) {
    out->clear();
    auto is_num = [](const llkv_expr_token &t) {
      return t.kind == LLKV_TOK_LITERAL && (t.literal.tag == LLKV_LIT_INT128 || t.literal.tag == LLKV_LIT_FLOAT64 || t.literal.tag == LLKV_LIT_NULL);
    };
    for (uint32_t i = 0; i < n; ++i) {
      const size_t m = out->size();
      if (!(e[i].kind == LLKV_TOK_BINARY && m >= 2 && is_num((*out)[m - 1]) && is_num((*out)[m - 2]))) { out->push_back(e[i]); continue; }
      const llkv_literal &a = (*out)[m - 2].literal, &b = (*out)[m - 1].literal;
      llkv_expr_token r = (*out)[m - 2];
      r.literal = llkv_literal{};
      const int op = e[i].binop;
      if (op < LLKV_BIN_ADD || op > LLKV_BIN_MOD) return fail(LLKV_UNSUPPORTED, "constant sub-expression under this operator");
      if (a.tag == LLKV_LIT_NULL || b.tag == LLKV_LIT_NULL) r.literal.tag = LLKV_LIT_NULL;
      else if (a.tag == LLKV_LIT_FLOAT64 || b.tag == LLKV_LIT_FLOAT64) {
        const double x = a.tag == LLKV_LIT_FLOAT64 ? a.f64 : (double)(int64_t)lit_i128(a), y = b.tag == LLKV_LIT_FLOAT64 ? b.f64 : (double)(int64_t)lit_i128(b);
        r.literal.tag = LLKV_LIT_FLOAT64;
        uint64_t ybits;
        memcpy(&ybits, &y, 8);
        switch (op) {
        case LLKV_BIN_ADD: r.literal.f64 = x + y; break;
        case LLKV_BIN_SUB: r.literal.f64 = x - y; break;
        case LLKV_BIN_MUL: r.literal.f64 = x * y; break;
        case LLKV_BIN_DIV: if (ybits == 0) r.literal.tag = LLKV_LIT_NULL; else r.literal.f64 = x / y; break;
        default: r.literal.f64 = std::fmod(x, y); break;
        }
      } else {
        const int64_t x = (int64_t)lit_i128(a), y = (int64_t)lit_i128(b);
        int64_t z = 0;
        bool bad = false, null = false;
        switch (op) {
        case LLKV_BIN_ADD: bad = __builtin_add_overflow(x, y, &z); break;
        case LLKV_BIN_SUB: bad = __builtin_sub_overflow(x, y, &z); break;
        case LLKV_BIN_MUL: bad = __builtin_mul_overflow(x, y, &z); break;
        case LLKV_BIN_DIV: if (y == 0) null = true; else if (x == INT64_MIN && y == -1) bad = true; else z = x / y; break;
        default: if (y == 0) bad = true; else z = y == -1 ? 0 : x % y; break;
        }
        if (bad) return fail(LLKV_UNSUPPORTED, "constant sub-expression the reference leaves unfolded (its fold errors)");
        if (null) r.literal.tag = LLKV_LIT_NULL;
        else { r.literal.tag = LLKV_LIT_INT128; r.literal.lo = (uint64_t)z; r.literal.hi = z < 0 ? -1 : 0; }
      }
      out->pop_back();
      out->back() = r;
    }
    return LLKV_OK;
  }

  int expr_fast(const llkv_expr_token *e_in, uint32_t n_in, std::string *node, bool *is_f64) {
    last_fast_32 = 0;
    std::vector<llkv_expr_token> folded;
    int frc = fold_constants(e_in, n_in, &folded);
    if (frc) return frc;
    const llkv_expr_token *e = folded.data();
    const uint32_t n = (uint32_t)folded.size();
    bool any_float = false, any_u64 = false, any_other = false;
    for (uint32_t i = 0; i < n; ++i) {
      if (e[i].kind == LLKV_TOK_COLUMN) {
        const ColumnInfo *ci = resolve(e[i].field_id);
        if (!ci) return fail(LLKV_NOT_FOUND, "field " + std::to_string(e[i].field_id) + " not found");
        if (ci->dtype == LLKV_DT_UINT64) { any_u64 = true; continue; }
        any_other = true;
        if (is_float_class(ci->dtype)) any_float = true;
        else if (!is_int_class(ci->dtype)) return fail(LLKV_UNSUPPORTED, std::string("computed projection over ") + dtype_name(ci->dtype));
      } else if (e[i].kind == LLKV_TOK_LITERAL) {
        any_other = true;
        if (e[i].literal.tag == LLKV_LIT_FLOAT64) any_float = true;
        else if (e[i].literal.tag != LLKV_LIT_INT128) return fail(LLKV_UNSUPPORTED, "non-numeric literal in computed projection");
      } else if (e[i].kind == LLKV_TOK_BINARY) {
        if (e[i].binop == LLKV_BIN_DIV) return expr_generic(e, n, node, is_f64); // Divide leaves the fast path (fast_numeric.rs:273-275)
      }
    }
    // Int32 ⊕ Int32 (UInt32 ⊕ UInt32) stays 32 bits wide in the reference (checked i32 arithmetic, Int32 result): what decides is the
    // ROOT type — columns are cast to it before the first kernel — and that is 32 bits wide only when every leaf is such a column
    int fit32 = 0;
    if (n > 1 && !any_float && !any_u64) {
      bool all_i32 = true, all_u32 = true;
      for (uint32_t i = 0; i < n; ++i) {
        if (e[i].kind == LLKV_TOK_LITERAL) all_i32 = all_u32 = false;
        else if (e[i].kind == LLKV_TOK_COLUMN) { const int32_t dt = resolve(e[i].field_id)->dtype; all_i32 &= dt == LLKV_DT_INT32; all_u32 &= dt == LLKV_DT_UINT32; }
      }
      if (all_i32 || all_u32) fit32 = all_i32 ? 1 : 2; // every node's result must fit 32 bits (Fit32, fused_scan.hip.h)
    }
    last_fast_32 = fit32;
    for (uint32_t i = 2; i < n; ++i) // (only non-numeric literal pairs are left unfolded)
      if (e[i].kind == LLKV_TOK_BINARY && e[i - 1].kind == LLKV_TOK_LITERAL && e[i - 2].kind == LLKV_TOK_LITERAL) return fail(LLKV_UNSUPPORTED, "constant sub-expression");
    // get_common_type (llkv-compute/src/kernels.rs:179-242): a 64-bit unsigned side with a signed side → Float64
    if (any_u64) {
      if (n > 1 && !any_other) return fail(LLKV_UNSUPPORTED, "UInt64-only arithmetic");
      if (n > 1) any_float = true;
    }
    std::vector<std::string> st;
    int rc;
    for (uint32_t i = 0; i < n; ++i) {
      if (e[i].kind == LLKV_TOK_COLUMN) {
        const ColumnInfo *ci;
        std::string c;
        if ((rc = expr_col_node(e[i].field_id, &ci, &c))) return rc;
        if (any_float) { if (ci->dtype != LLKV_DT_FLOAT64) c = "ToF64<" + c + ">"; }
        else if (ci->dtype != LLKV_DT_INT64 && ci->dtype != LLKV_DT_UINT64) c = "ToI64<" + c + ">";
        st.push_back(c);
      } else if (e[i].kind == LLKV_TOK_LITERAL) {
        std::string l;
        const llkv_literal &lit = e[i].literal;
        if (any_float) rc = lit_f(lit.tag == LLKV_LIT_FLOAT64 ? lit.f64 : (double)lit_i128(lit), &l);
        else rc = lit_i((int64_t)lit_i128(lit), &l);
        if (rc) return rc;
        st.push_back(l);
      } else {
        if (st.size() < 2) return fail(LLKV_INTERNAL, "fast path stack underflow");
        std::string r = st.back(); st.pop_back();
        std::string l = st.back(); st.pop_back();
        const int op = e[i].binop == LLKV_BIN_ADD ? 1 : e[i].binop == LLKV_BIN_SUB ? 2 : e[i].binop == LLKV_BIN_MUL ? 3 : 4;
        std::string b = "Bin<" + std::to_string(op) + "," + l + "," + r + nan_flag(any_float) + ">";
        if (fit32) b = "Fit32<" + b + "," + (fit32 == 1 ? "1" : "0") + ">";
        st.push_back(b);
      }
    }
    if (st.size() != 1) return fail(LLKV_INTERNAL, "fast path evaluation missing result");
    *node = st[0];
    *is_f64 = any_float;
    return LLKV_OK;
  }

  // Expressions with a Divide take the generic route (try_evaluate_vectorized llkv-compute/src/eval.rs:616-665 →
  // compute_binary kernels.rs:99-177): every Binary node coerces its own two operands to their common type
  // (not the whole expression to one final type), integers stay checked, zeros of a divisor become NULLs.
  // Restated for Int64 / Float64 operands; narrower or unsigned ones keep the query on the caller's route.
  int expr_generic(const llkv_expr_token *e, uint32_t n, std::string *node, bool *is_f64) {
    struct V { std::string s; bool f; };
    std::vector<V> st;
    int rc;
    for (uint32_t i = 0; i < n; ++i) {
      if (e[i].kind == LLKV_TOK_COLUMN) {
        const ColumnInfo *ci;
        std::string c;
        if ((rc = expr_col_node(e[i].field_id, &ci, &c))) return rc;
        if (ci->dtype != LLKV_DT_INT64 && ci->dtype != LLKV_DT_FLOAT64)
          return fail(LLKV_UNSUPPORTED, std::string("division over ") + dtype_name(ci->dtype) + " operands (per-node typing of narrow types)");
        st.push_back({c, ci->dtype == LLKV_DT_FLOAT64});
      } else if (e[i].kind == LLKV_TOK_LITERAL) {
        std::string l;
        const llkv_literal &lit = e[i].literal;
        if (lit.tag == LLKV_LIT_FLOAT64) { if ((rc = lit_f(lit.f64, &l))) return rc; st.push_back({l, true}); }
        else if (lit.tag == LLKV_LIT_INT128) { if ((rc = lit_i((int64_t)lit_i128(lit), &l))) return rc; st.push_back({l, false}); } // `*i as i64`
        else return fail(LLKV_UNSUPPORTED, "literal kind in a computed projection");
      } else {
        if (st.size() < 2) return fail(LLKV_INTERNAL, "expression stack underflow");
        V r = st.back(); st.pop_back();
        V l = st.back(); st.pop_back();
        const bool f = l.f || r.f; // get_common_type: Int64 ⊕ Float64 → Float64
        const std::string a = (f && !l.f) ? "ToF64<" + l.s + ">" : l.s, b = (f && !r.f) ? "ToF64<" + r.s + ">" : r.s;
        if (e[i].binop == LLKV_BIN_DIV) st.push_back({"Div<" + a + "," + b + nan_flag(f) + ">", f});
        else {
          const int op = e[i].binop == LLKV_BIN_ADD ? 1 : e[i].binop == LLKV_BIN_SUB ? 2 : e[i].binop == LLKV_BIN_MUL ? 3 : 4;
          st.push_back({"Bin<" + std::to_string(op) + "," + a + "," + b + nan_flag(f) + ">", f});
        }
      }
    }
    if (st.size() != 1) return fail(LLKV_INTERNAL, "expression evaluation missing result");
    *node = st[0].s;
    *is_f64 = st[0].f;
    return LLKV_OK;
  }

  // What a GROUP BY aggregate argument evaluates to (PlanValue::{Integer, Float, Decimal}); for the integer and decimal
  // classes the interval [lo, hi] the column statistics and literals leave its 64-bit image in.
  struct PV {
    std::string s;
    bool f = false;     // Float
    bool dec = false;   // Decimal(scale): the value is raw / 10^scale
    int scale = 0;
    bool bounded = false;
    i128 lo = 0, hi = 0;
    bool lit = false;   // an integer / decimal literal (lo == hi)
  };
  PlanValueInfo last_planvalue = {}; // (expr_planvalue) class, scale and bounds of the expression just lowered

  static bool fits_i64(i128 v) { return v >= (i128)INT64_MIN && v <= (i128)INT64_MAX; }
  static i128 pow10_i128(int k) { i128 v = 1; for (int i = 0; i < k; ++i) v *= 10; return v; }

  // GROUP BY aggregate argument, PlanValue semantics (llkv-executor/src/lib.rs:7193-7389).
  int expr_planvalue(const llkv_expr_token *e, uint32_t n, std::string *node, bool *is_f64) {
    std::vector<PV> st;
    int rc;
    last_planvalue = PlanValueInfo{};
    for (uint32_t i = 0; i < n; ++i) {
      if (e[i].kind == LLKV_TOK_COLUMN) {
        const ColumnInfo *ci;
        std::string c;
        if ((rc = expr_col_node(e[i].field_id, &ci, &c))) return rc;
        PV v;
        if (ci->dtype == LLKV_DT_FLOAT64) { v.s = c; v.f = true; }
        else if (ci->dtype == LLKV_DT_FLOAT32) { v.s = "ToF64<" + c + ">"; v.f = true; }
        else if (ci->dtype == LLKV_DT_INT64) v.s = c;
        else if (is_int_class(ci->dtype)) v.s = "ToI64<" + c + ">";
        else if (ci->dtype == LLKV_DT_DECIMAL128) {
          // plan_value_from_array (llkv-plan/src/plans.rs:1160-1174): DecimalValue::new(raw, scale) — a 64-bit image has at
          // most 19 digits, so the 38-digit check cannot fail; the scale must lie within ±38
          if (ci->scale < -38 || ci->scale > 38) return fail(LLKV_UNSUPPORTED, "Decimal128 scale outside ±38 in an aggregate expression (the reference fails the conversion)");
          v.s = c; v.dec = true; v.scale = ci->scale;
        }
        else return fail(LLKV_UNSUPPORTED, std::string("aggregate expression over ") + dtype_name(ci->dtype));
        if (!v.f) {
          v.bounded = true;
          if (ci->has_stats) { v.lo = ci->min_i; v.hi = ci->max_i; }
          else if (ci->dtype == LLKV_DT_INT32 || ci->dtype == LLKV_DT_DATE32) { v.lo = INT32_MIN; v.hi = INT32_MAX; }
          else if (ci->dtype == LLKV_DT_UINT32) { v.lo = 0; v.hi = UINT32_MAX; }
          else { v.lo = INT64_MIN; v.hi = INT64_MAX; }
        }
        st.push_back(v);
      } else if (e[i].kind == LLKV_TOK_LITERAL) {
        std::string l;
        const llkv_literal &lit = e[i].literal;
        PV v;
        if (lit.tag == LLKV_LIT_FLOAT64) { if ((rc = lit_f(lit.f64, &l))) return rc; v.f = true; }
        else if (lit.tag == LLKV_LIT_INT128) { // `*v as i64` (:7019)
          const int64_t x = (int64_t)lit_i128(lit);
          if ((rc = lit_i(x, &l))) return rc;
          v.bounded = true; v.lo = v.hi = x; v.lit = true;
        }
        else if (lit.tag == LLKV_LIT_DECIMAL128) { // Literal::Decimal128(DecimalValue) (:7021)
          const i128 raw = lit_i128(lit);
          if (!fits_i64(raw) || lit.scale < -38 || lit.scale > 38) return fail(LLKV_UNSUPPORTED, "decimal literal beyond 64 bits in an aggregate expression");
          if ((rc = lit_i((int64_t)raw, &l))) return rc;
          v.dec = true; v.scale = lit.scale; v.bounded = true; v.lo = v.hi = raw; v.lit = true;
        }
        else return fail(LLKV_UNSUPPORTED, "literal kind in aggregate expression");
        v.s = l;
        st.push_back(v);
      } else {
        if (st.size() < 2) return fail(LLKV_INTERNAL, "expression stack underflow");
        PV r = st.back(); st.pop_back();
        PV l = st.back(); st.pop_back();
        if (l.dec || r.dec) {
          // a Decimal operand: exact decimal arithmetic (:7229-7330).  Int / Int was decided before this arm; a Float operand
          // and Modulo are errors the reference raises on the first row whose operands are both non-NULL — not decidable here
          if (l.f || r.f) return fail(LLKV_UNSUPPORTED, "decimal arithmetic with a Float operand (the reference raises an error on the first non-NULL row)");
          if (e[i].binop == LLKV_BIN_MOD) return fail(LLKV_UNSUPPORTED, "Modulo over Decimal operands (an error in the reference)");
          PV o;
          o.dec = true;
          o.bounded = true;
          // rescale: value · 10^diff (decimal.rs:29-49), admitted when the statistics keep the product inside 64 bits
          auto rescaled = [&](const PV &x, int target, PV *out) -> int {
            *out = x;
            out->scale = target;
            const int diff = target - x.scale;
            if (diff == 0) return LLKV_OK;
            if (diff > 18) return fail(LLKV_UNSUPPORTED, "decimal rescale beyond 64 bits in an aggregate expression");
            const i128 f = pow10_i128(diff);
            out->lo = x.lo * f; out->hi = x.hi * f;
            if (!fits_i64(out->lo) || !fits_i64(out->hi)) return fail(LLKV_UNSUPPORTED, "decimal intermediate beyond 64 bits in an aggregate expression");
            if (x.lit) return lit_i((int64_t)out->lo, &out->s); // a literal is rescaled here
            std::string k;
            int rc2 = lit_i((int64_t)f, &k);
            if (rc2) return rc2;
            out->s = "DecBin<3," + x.s + "," + k + ">";
            return LLKV_OK;
          };
          if (e[i].binop == LLKV_BIN_ADD || e[i].binop == LLKV_BIN_SUB) {
            const int target = std::max(l.scale, r.scale);
            PV a, b;
            if ((rc = rescaled(l, target, &a)) || (rc = rescaled(r, target, &b))) return rc;
            o.scale = target;
            if (e[i].binop == LLKV_BIN_ADD) { o.lo = a.lo + b.lo; o.hi = a.hi + b.hi; }
            else { o.lo = a.lo - b.hi; o.hi = a.hi - b.lo; }
            o.s = std::string("DecBin<") + (e[i].binop == LLKV_BIN_ADD ? "1," : "2,") + a.s + "," + b.s + ">";
          } else if (e[i].binop == LLKV_BIN_MUL) {
            o.scale = l.scale + r.scale;
            if (o.scale < -38 || o.scale > 38) return fail(LLKV_UNSUPPORTED, "decimal product scale outside ±38 (an error in the reference)");
            const i128 c4[4] = {l.lo * r.lo, l.lo * r.hi, l.hi * r.lo, l.hi * r.hi};
            o.lo = std::min(std::min(c4[0], c4[1]), std::min(c4[2], c4[3]));
            o.hi = std::max(std::max(c4[0], c4[1]), std::max(c4[2], c4[3]));
            o.s = "DecBin<3," + l.s + "," + r.s + ">";
          } else { // Divide: to the left operand's scale; numerator · 10^(divisor's scale)
            o.scale = l.scale;
            if (r.scale < 0 || r.scale > 18) return fail(LLKV_UNSUPPORTED, "decimal division by an operand of this scale is not on the GPU path");
            const i128 f = pow10_i128(r.scale);
            const i128 nlo = l.lo * f, nhi = l.hi * f;
            if (!fits_i64(nlo) || !fits_i64(nhi)) return fail(LLKV_UNSUPPORTED, "decimal intermediate beyond 64 bits in an aggregate expression");
            const i128 m = std::max(nlo < 0 ? -nlo : nlo, nhi < 0 ? -nhi : nhi) + 1; // |quotient| ≤ |numerator|, + 1 for the rounding
            o.lo = -m; o.hi = m;
            std::string k;
            if ((rc = lit_i((int64_t)f, &k))) return rc;
            o.s = "DecDiv<" + l.s + "," + r.s + "," + k + ">";
          }
          if (!fits_i64(o.lo) || !fits_i64(o.hi)) return fail(LLKV_UNSUPPORTED, "decimal intermediate beyond 64 bits in an aggregate expression");
          st.push_back(o);
          continue;
        }
        if (e[i].binop == LLKV_BIN_DIV || (e[i].binop == LLKV_BIN_MOD && (l.f || r.f))) {
          // Int / Int truncates but turns Float for i64::MIN / -1 (:7213-7227): the type of the group's temp column
          // would depend on the data — unless the statistics (or a literal divisor) exclude that pair
          if (!l.f && !r.f) {
            const bool never_min = l.bounded && l.lo > (i128)INT64_MIN, never_minus_one = r.bounded && (r.lo > -1 || r.hi < -1);
            if (!never_min && !never_minus_one)
              return fail(LLKV_UNSUPPORTED, "integer division in GROUP BY aggregate arguments whose operands may be i64::MIN / −1 (it turns Float in the reference: the temp column's type would depend on the data)");
            PV o;
            o.s = "DivIntPV<" + l.s + "," + r.s + ">";
            o.bounded = true;
            const i128 m = std::max(l.lo < 0 ? -l.lo : l.lo, l.hi < 0 ? -l.hi : l.hi);
            o.lo = -m; o.hi = m;
            if (!fits_i64(o.lo)) o.lo = INT64_MIN;
            if (!fits_i64(o.hi)) o.hi = INT64_MAX;
            st.push_back(o);
            continue;
          }
          PV o;
          o.s = std::string("DivPV<") + (e[i].binop == LLKV_BIN_MOD ? "1" : "0") + "," + l.s + "," + r.s + ">";
          o.f = true;
          st.push_back(o);
          continue;
        }
        const int op = e[i].binop == LLKV_BIN_ADD ? 1 : e[i].binop == LLKV_BIN_SUB ? 2 : e[i].binop == LLKV_BIN_MUL ? 3 : 4;
        PV o;
        if (!l.f && !r.f) {
          o.s = "BinViaF64<" + std::to_string(op) + "," + l.s + "," + r.s + ">";
          // (computed in f64 and cast back: exact only below 2^53 — beyond that the interval is everything an i64 holds)
          o.bounded = true;
          o.lo = INT64_MIN; o.hi = INT64_MAX;
          const i128 lim = (i128)1 << 53;
          if (op != 4 && l.bounded && r.bounded) {
            i128 a, b;
            if (op == 1) { a = l.lo + r.lo; b = l.hi + r.hi; }
            else if (op == 2) { a = l.lo - r.hi; b = l.hi - r.lo; }
            else {
              const i128 c4[4] = {l.lo * r.lo, l.lo * r.hi, l.hi * r.lo, l.hi * r.hi};
              a = std::min(std::min(c4[0], c4[1]), std::min(c4[2], c4[3]));
              b = std::max(std::max(c4[0], c4[1]), std::max(c4[2], c4[3]));
            }
            if (a > -lim && b < lim && l.lo > -lim && l.hi < lim && r.lo > -lim && r.hi < lim) { o.lo = a; o.hi = b; }
          } else if (op == 4 && l.bounded && r.bounded && l.lo > -lim && l.hi < lim && r.lo > -lim && r.hi < lim) {
            // fmod keeps the dividend's sign and stays below the divisor in magnitude
            const i128 ml = std::max(l.lo < 0 ? -l.lo : l.lo, l.hi < 0 ? -l.hi : l.hi), mr = std::max(r.lo < 0 ? -r.lo : r.lo, r.hi < 0 ? -r.hi : r.hi);
            const i128 m = std::min(ml, mr > 0 ? mr - 1 : (i128)0);
            o.lo = l.lo < 0 ? -m : 0;
            o.hi = l.hi > 0 ? m : 0;
          }
        } else {
          const std::string a = l.f ? l.s : "ToF64<" + l.s + ">", b = r.f ? r.s : "ToF64<" + r.s + ">";
          o.s = "Bin<" + std::to_string(op) + "," + a + "," + b + nan_flag(true) + ">";
          o.f = true;
        }
        st.push_back(o);
      }
    }
    if (st.size() != 1) return fail(LLKV_INTERNAL, "expression evaluation missing result");
    *node = st[0].s;
    *is_f64 = st[0].f;
    last_planvalue.is_decimal = st[0].dec;
    last_planvalue.scale = st[0].scale;
    last_planvalue.lo = st[0].lo;
    last_planvalue.hi = st[0].hi;
    last_planvalue.bounded = st[0].bounded && !st[0].f;
    return LLKV_OK;
  }
};

} // namespace

// Aggregate list → deduplicated lane groups (node strings + lane ops) and one AggOut per aggregate.
static int lower_aggregates(Lowering &L, const ColumnResolver &resolve, const llkv_aggregate_spec *aggs, uint32_t n_aggs, bool grouped,
                            std::vector<std::string> &groups, std::vector<std::vector<uint8_t>> &group_ops, int &next_lane) {
  LoweredPlan &p = L.p;
  int rc;
  std::vector<int> group_lane;     // first lane (relative to base) of each lane group
  auto add_group = [&](const std::string &node, std::vector<uint8_t> lane_ops) -> int {
    std::vector<std::pair<uint8_t, uint8_t>> expand;
    expand.swap(L.next_expand);
    const double narrow = L.next_narrow;
    L.next_narrow = -1.0;
    for (size_t i = 0; i < groups.size(); ++i) if (groups[i] == node) return group_lane[i];
    L.group_narrow.push_back(narrow);
    if (expand.empty()) for (size_t j = 0; j < lane_ops.size(); ++j) expand.emplace_back((uint8_t)j, (uint8_t)0);
    groups.push_back(node);
    group_lane.push_back(next_lane);
    group_ops.push_back(lane_ops);
    L.group_expand.push_back(expand);
    next_lane += (int)lane_ops.size();
    return group_lane.back();
  };
  enum { ADD_F64 = 0, ADD_I64 = 1, MIN_I64 = 2, MAX_I64 = 3, MAX_U64 = 4 };
  // computed decimal arguments of a GROUP BY: the digit count of each group's first non-NULL value types its temp column
  // (FirstDigits, fused_scan.hip.h) — arguments that are NULL in the same rows share a lane, packed after the loop
  struct PendingDigits { size_t agg; std::string valid, node; int dlo, dhi, scale; uint64_t rows; };
  std::vector<PendingDigits> pending_digits;
  auto digits_of = [](i128 v) { int d = 0; u128 m = v < 0 ? (u128)(-v) : (u128)v; do { m /= 10; ++d; } while (m); return d; };

  for (uint32_t a = 0; a < n_aggs; ++a) {
    const llkv_aggregate_spec &s = aggs[a];
    AggOut o{AggFinal::CountRows, -1};
    if (s.distinct && s.kind != LLKV_AGG_MIN && s.kind != LLKV_AGG_MAX && s.kind != LLKV_AGG_COUNT_STAR) {
      // every group runs the reference's distinct accumulator over its own rows (llkv-executor/src/lib.rs:5222-5247):
      // on the sort-based route the argument column is one more sort key and the reduction counts the first row of
      // every run of equal values
      if (!L.allow_sorted_distinct) return L.fail(LLKV_UNSUPPORTED, "DISTINCT aggregates inside GROUP BY run on the sort-based route");
      if (s.kind != LLKV_AGG_COUNT && s.kind != LLKV_AGG_SUM && s.kind != LLKV_AGG_TOTAL && s.kind != LLKV_AGG_AVG)
        return L.fail(LLKV_UNSUPPORTED, "DISTINCT form of aggregate kind " + std::to_string(s.kind));
      if (!s.expr || s.expr_len == 0) return L.fail(LLKV_INVALID_ARGUMENT, "aggregate requires an argument");
      if (s.expr_len != 1 || s.expr[0].kind != LLKV_TOK_COLUMN) {
        // a computed argument: the group's temp column holds the PlanValue of every row (llkv-executor/src/lib.rs:5186-5199) and the
        // distinct accumulator runs over it — Int by value, Float by bits.  The values are computed once for the selected rows by a
        // projection plan of their own and sort as a column's cells would.
        if (p.distinct_field >= 0) return L.fail(LLKV_UNSUPPORTED, "DISTINCT aggregates over more than one argument in a GROUP BY");
        auto dp = std::make_shared<LoweredPlan>();
        Lowering LP{resolve, *dp, L.err, true};
        LP.exact_nan = true; // the values are told apart by their bits
        std::string dnode, dvalid;
        bool df = false;
        if ((rc = LP.expr_planvalue(s.expr, s.expr_len, &dnode, &df))) return rc;
        const PlanValueInfo dpv = LP.last_planvalue;
        if (dpv.is_decimal) return L.fail(LLKV_UNSUPPORTED, "DISTINCT inside GROUP BY over a computed decimal argument");
        if ((rc = LP.valid_of_node(s.expr, s.expr_len, dnode, true, &dvalid))) return rc;
        auto same_tokens = [&]() {
          if (p.distinct_tokens.size() != s.expr_len) return false;
          for (uint32_t k = 0; k < s.expr_len; ++k) {
            const llkv_expr_token &x = p.distinct_tokens[k], &y = s.expr[k];
            if (x.kind != y.kind) return false;
            if (x.kind == LLKV_TOK_COLUMN && x.field_id != y.field_id) return false;
            if (x.kind == LLKV_TOK_BINARY && x.binop != y.binop) return false;
            if (x.kind == LLKV_TOK_LITERAL && (x.literal.tag != y.literal.tag || x.literal.scale != y.literal.scale || x.literal.lo != y.literal.lo || x.literal.hi != y.literal.hi ||
                                               std::memcmp(&x.literal.f64, &y.literal.f64, 8) != 0)) return false;
          }
          return true;
        };
        if (p.distinct_proj && !same_tokens()) return L.fail(LLKV_UNSUPPORTED, "DISTINCT aggregates over more than one argument in a GROUP BY");
        if (!p.distinct_proj) {
          p.distinct_tokens.assign(s.expr, s.expr + s.expr_len);
          for (auto &tk : p.distinct_tokens) tk.literal.str = nullptr; // (numeric literals only: expr_planvalue took nothing else)
          dp->out_dtypes = {df ? LLKV_DT_FLOAT64 : LLKV_DT_INT64};
          dp->out_fields = {-1};
          dp->out_nullable = {!dvalid.empty()};
          dp->type_string = "ProjPlan<" + cols_string(*dp, &dp->bytes_per_row) + ",Outs<" + (dvalid.empty() ? dnode : "OutV<" + dnode + "," + dvalid + ">") + ">>";
          p.distinct_proj = dp;
          p.distinct_node = dnode;
          p.distinct_numeric = 0;
        }
        const int count_lane = add_group("DistinctCount", {ADD_I64});
        if (s.kind == LLKV_AGG_COUNT) { o.fin = AggFinal::CountValid; o.lane = count_lane; p.aggs.push_back(o); continue; }
        if (!df && s.kind != LLKV_AGG_TOTAL) { // the checked_add chain over the distinct values cannot overflow whatever their order
          auto mag = [](i128 v) -> u128 { return v < 0 ? (u128)(-v) : (u128)v; };
          uint64_t rows = 0;
          for (uint32_t k = 0; k < s.expr_len; ++k)
            if (s.expr[k].kind == LLKV_TOK_COLUMN) { const ColumnInfo *ci = resolve(s.expr[k].field_id); if (ci) rows = std::max(rows, ci->rows); }
          const u128 m = std::max(mag(dpv.lo), mag(dpv.hi));
          if (!dpv.bounded || m * (u128)rows > (u128)INT64_MAX) return L.fail(LLKV_UNSUPPORTED, "possible i64 overflow in SUM(DISTINCT) over a computed argument: order-dependent check is not on the GPU path");
        }
        o.count_lane = count_lane;
        o.typed_by_first_value = true; // (a group without a non-NULL value: the temp column is an Int64 column — SUM comes back as an Int64 NULL)
        if (s.kind == LLKV_AGG_TOTAL) { o.fin = AggFinal::TotalF64; o.lane = add_group(df ? "DistinctSumF64" : "DistinctTotalI64", {ADD_F64}); }
        else if (df) { o.fin = s.kind == LLKV_AGG_SUM ? AggFinal::SumF64 : AggFinal::AvgF64; o.lane = add_group("DistinctSumF64", {ADD_F64}); }
        else { o.fin = s.kind == LLKV_AGG_SUM ? AggFinal::SumI64Fast : AggFinal::AvgI64Fast; o.lane = add_group("DistinctSumI64", {ADD_I64}); }
        p.aggs.push_back(o);
        continue;
      }
      if (p.distinct_proj) return L.fail(LLKV_UNSUPPORTED, "DISTINCT aggregates over more than one argument in a GROUP BY");
      const ColumnInfo *dci = resolve(s.expr[0].field_id);
      if (!dci) return L.fail(LLKV_INVALID_ARGUMENT, "unknown column '" + std::to_string(s.expr[0].field_id) + "' in aggregate");
      // DistinctKey::from_array (llkv-aggregate/src/lib.rs:261-331): Int by value, Float by bits, Str by its string (here: its
      // dictionary code — the staged dictionary holds every string once), Bool, Date by its day number; what SUM / TOTAL / AVG
      // add for the last three is their numeric image in the Float64 accumulators (:400-449,889-924,1035-1066,1200-1232)
      const bool keyed = dci->dtype == LLKV_DT_UTF8 || dci->dtype == LLKV_DT_BOOLEAN || dci->dtype == LLKV_DT_DATE32;
      // … and Decimal by its raw value (the 64-bit image; Sum / Total / AvgDistinctDecimal128 :943-967,1089-1112,1260-1284)
      const bool dec = dci->dtype == LLKV_DT_DECIMAL128 && !dci->wide128;
      if (dci->dtype != LLKV_DT_INT64 && dci->dtype != LLKV_DT_FLOAT64 && !keyed && !dec) return L.fail(LLKV_UNSUPPORTED, std::string("DISTINCT aggregate over ") + dtype_name(dci->dtype));
      if (p.distinct_field >= 0 && p.distinct_field != (int64_t)s.expr[0].field_id)
        return L.fail(LLKV_UNSUPPORTED, "DISTINCT aggregates over more than one column in a GROUP BY");
      p.distinct_field = s.expr[0].field_id;
      p.distinct_numeric = dci->dtype == LLKV_DT_UTF8 ? 1u : dci->dtype == LLKV_DT_BOOLEAN ? 2u : dci->dtype == LLKV_DT_DATE32 ? 3u : 0u;
      if (dci->dtype == LLKV_DT_UTF8 && p.distinct_dict_num.empty()) {
        p.distinct_dict_num.assign(256, 0.0);
        for (size_t c = 0; c < dci->dictionary.size() && c < 256; ++c) p.distinct_dict_num[c] = parse_numeric_or_zero(dci->dictionary[c]);
      }
      const bool f = dci->dtype == LLKV_DT_FLOAT64 || keyed;
      const int count_lane = add_group("DistinctCount", {ADD_I64});
      if (s.kind == LLKV_AGG_COUNT) { o.fin = AggFinal::CountValid; o.lane = count_lane; p.aggs.push_back(o); continue; }
      if (!f && (s.kind != LLKV_AGG_TOTAL || dec)) { // the checked_add chain over the distinct values cannot overflow whatever their order (decimals: the i64 lane holds their i128 sum)
        auto mag = [](int64_t v) -> u128 { return v < 0 ? (u128)(-(i128)v) : (u128)v; };
        if (!dci->has_stats) return L.fail(LLKV_UNSUPPORTED, "SUM(DISTINCT) over an integer column without statistics (order-dependent overflow check)");
        const u128 m = mag(dci->min_i) > mag(dci->max_i) ? mag(dci->min_i) : mag(dci->max_i);
        if (m * (u128)dci->rows > (u128)INT64_MAX) return L.fail(LLKV_UNSUPPORTED, "possible i64 overflow in SUM(DISTINCT): order-dependent check is not on the GPU path");
      }
      o.count_lane = count_lane;
      if (dec) { // finalize :1583-1612,1656-1672,1762-1800: i128 sum with the column's (precision, scale); SUM / AVG are NULL without a value
        o.fin = s.kind == LLKV_AGG_SUM ? AggFinal::SumDec : s.kind == LLKV_AGG_TOTAL ? AggFinal::TotalDec : AggFinal::AvgDec;
        o.precision = dci->precision; o.scale = dci->scale;
        o.fast_sum = true;
        o.null_without_values = s.kind != LLKV_AGG_TOTAL;
        o.lane = add_group("DistinctSumI64", {ADD_I64});
        p.aggs.push_back(o);
        continue;
      }
      if (s.kind == LLKV_AGG_TOTAL) { o.fin = AggFinal::TotalF64; o.lane = add_group(f ? "DistinctSumF64" : "DistinctTotalI64", {ADD_F64}); }
      else if (f) { o.fin = s.kind == LLKV_AGG_SUM ? AggFinal::SumF64 : AggFinal::AvgF64; o.lane = add_group("DistinctSumF64", {ADD_F64}); }
      else { o.fin = s.kind == LLKV_AGG_SUM ? AggFinal::SumI64Fast : AggFinal::AvgI64Fast; o.lane = add_group("DistinctSumI64", {ADD_I64}); }
      p.aggs.push_back(o);
      continue;
    }
    if (s.distinct && s.kind == LLKV_AGG_COUNT_STAR) return L.fail(LLKV_UNSUPPORTED, "COUNT(DISTINCT *)");
    if (s.kind == LLKV_AGG_COUNT_STAR) { p.aggs.push_back(o); continue; }
    if (!s.expr || s.expr_len == 0) return L.fail(LLKV_INVALID_ARGUMENT, "aggregate requires an argument");
    const bool simple = s.expr_len == 1 && s.expr[0].kind == LLKV_TOK_COLUMN;
    std::string node;
    bool is_f64 = false;
    const ColumnInfo *simple_ci = nullptr;
    if (simple) {
      simple_ci = resolve(s.expr[0].field_id);
      if (!simple_ci) return L.fail(LLKV_INVALID_ARGUMENT, "unknown column '" + std::to_string(s.expr[0].field_id) + "' in aggregate");
    }
    // rows where the argument is non-NULL (NULL propagates through arithmetic, x / 0 is NULL; accumulators
    // skip NULLs, llkv-aggregate/src/lib.rs:769-786,801-830)
    std::string valid;
    if (simple) {
      std::vector<std::string> vs;
      if ((rc = L.valid_of_expr(s.expr, s.expr_len, &vs))) return rc;
      valid = Lowering::all_of(vs);
    } else {
      rc = grouped ? L.expr_planvalue(s.expr, s.expr_len, &node, &is_f64) : L.expr_fast(s.expr, s.expr_len, &node, &is_f64);
      if (rc) return rc;
      if (!grouped && L.last_fast_32) return L.fail(LLKV_UNSUPPORTED, "aggregate over a 32-bit-only integer expression (the reference has no Int32 accumulator)");
      // (what can make the value NULL is read off the expression the scan evaluates: the simplified one — a division of
      // constants is gone from it)
      std::vector<llkv_expr_token> folded;
      if (!grouped && (rc = L.fold_constants(s.expr, s.expr_len, &folded))) return rc;
      if ((rc = grouped ? L.valid_of_node(s.expr, s.expr_len, node, grouped, &valid) : L.valid_of_node(folded.data(), (uint32_t)folded.size(), node, grouped, &valid))) return rc;
    }
    const PlanValueInfo pvi = (grouped && !simple) ? L.last_planvalue : PlanValueInfo{};
    if (pvi.is_decimal) {
      // every aggregate kind — COUNT too — goes through the group's temp column (llkv-executor/src/lib.rs:5186-5199)
      PendingDigits pd;
      pd.agg = p.aggs.size();
      pd.valid = valid.empty() ? "True" : valid;
      pd.node = node;
      const i128 alo = pvi.lo < 0 ? -pvi.lo : pvi.lo, ahi = pvi.hi < 0 ? -pvi.hi : pvi.hi;
      pd.dlo = (pvi.lo <= 0 && pvi.hi >= 0) ? 1 : digits_of(std::min(alo, ahi));
      pd.dhi = digits_of(std::max(alo, ahi));
      pd.rows = 0;
      pd.scale = pvi.scale;
      for (uint32_t k = 0; k < s.expr_len; ++k)
        if (s.expr[k].kind == LLKV_TOK_COLUMN) { const ColumnInfo *ci = resolve(s.expr[k].field_id); if (ci) pd.rows = std::max(pd.rows, ci->rows); }
      pending_digits.push_back(pd);
    }
    if (s.kind == LLKV_AGG_COUNT || s.kind == LLKV_AGG_COUNT_NULLS) {
      // NULL-free argument: COUNT(x) = rows, COUNT_NULLS(x) = 0
      if (!simple && !grouped) {
        // a computed argument is evaluated for its checked-arithmetic errors even though only its validity counts
        o.fin = s.kind == LLKV_AGG_COUNT ? AggFinal::CountValid : AggFinal::CountNulls;
        o.lane = add_group("CountIfE<" + (valid.empty() ? std::string("True") : valid) + "," + node + ">", {ADD_I64});
      } else if (valid.empty()) o.fin = s.kind == LLKV_AGG_COUNT ? AggFinal::CountRows : AggFinal::CountNullsZero;
      else {
        o.fin = s.kind == LLKV_AGG_COUNT ? AggFinal::CountValid : AggFinal::CountNulls;
        L.next_narrow = 1.0;
        o.lane = add_group("CountIf<" + valid + ">", {ADD_I64});
      }
      p.aggs.push_back(o);
      continue;
    }
    const char *fn = s.kind == LLKV_AGG_SUM ? "SUM" : s.kind == LLKV_AGG_TOTAL ? "TOTAL" : s.kind == LLKV_AGG_AVG ? "AVG" : s.kind == LLKV_AGG_MIN ? "MIN" : "MAX";
    if (simple && !simple_ci->wide128) {
      // validate_aggregate_type llkv-executor/src/lib.rs:5946-5988
      const int32_t dt = simple_ci->dtype;
      // Utf8 / Boolean / Date32 inputs get Float64 accumulators fed by array_value_to_numeric (llkv-aggregate/src/lib.rs:
      // 400-449): strings parse or count as 0, booleans are 0 / 1; Date32 has no arm there — the first non-NULL row fails
      if (dt == LLKV_DT_DATE32)
        return L.fail(LLKV_UNSUPPORTED, std::string(fn) + " over Date32 (the reference fails at the first non-NULL row) is not on the GPU path");
      if (dt != LLKV_DT_INT64 && dt != LLKV_DT_FLOAT64 && dt != LLKV_DT_DECIMAL128 && dt != LLKV_DT_UTF8 && dt != LLKV_DT_BOOLEAN)
        return L.fail(LLKV_INVALID_ARGUMENT, std::string(fn) + " aggregate not supported for column type " + dtype_name(dt));
      const ColumnInfo *ci;
      int slot;
      if ((rc = L.slot_of(s.expr[0].field_id, &ci, &slot))) return rc;
      if (dt == LLKV_DT_UTF8) {
        if (!L.allow_dict_num) return L.fail(LLKV_UNSUPPORTED, std::string(fn) + " over a Utf8 column is not on this route");
        node = "DictNum<" + std::to_string(slot) + ">";
        bool have = false;
        for (auto &d : p.dict_num) have |= d.first == slot;
        if (!have) {
          std::vector<double> image(256, 0.0);
          for (size_t c = 0; c < ci->dictionary.size() && c < 256; ++c) image[c] = parse_numeric_or_zero(ci->dictionary[c]);
          p.dict_num.emplace_back(slot, std::move(image));
        }
      } else if (dt == LLKV_DT_BOOLEAN) {
        node = "ToF64<" + L.col_node(slot, dt) + ">";
      } else {
        node = L.col_node(slot, dt);
      }
      is_f64 = dt == LLKV_DT_FLOAT64 || dt == LLKV_DT_UTF8 || dt == LLKV_DT_BOOLEAN;
    } else {
      o.typed_by_first_value = grouped && !simple; // (a bare column keeps its own type: only a computed argument becomes a temp column)
    }
    // statistics that exclude i64 overflow of any prefix sum: rows · max|v| ≤ i64::MAX
    bool fast_i64 = false;
    if (!is_f64 && simple && simple_ci->has_stats) {
      auto mag = [](int64_t v) -> u128 { return v < 0 ? (u128)(-(i128)v) : (u128)v; };
      u128 m = mag(simple_ci->min_i) > mag(simple_ci->max_i) ? mag(simple_ci->min_i) : mag(simple_ci->max_i);
      fast_i64 = m * (u128)simple_ci->rows <= (u128)INT64_MAX;
    }
    // shared-image plans: SumF64<node> → SumF64X<node, C1, C2> (exact, order-free); needs a bound on |argument|
    bool no_bound = false;
    auto sum_f64 = [&](const std::string &arg) -> std::pair<std::string, std::vector<uint8_t>> {
      if (!L.exact_f64) return {"SumF64<" + arg + ">", {ADD_F64}};
      double absmax = 0.0, nzmin = 0.0, c[4];
      int levels = 0;
      const bool bounded = L.expr_bounds(s.expr, s.expr_len, &absmax, &nzmin);
      const uint64_t n_rows = simple_ci ? simple_ci->rows : L.table_rows;
      int e = 0;
      std::string scale_lit;
      if (!L.image_plan && bounded && L.bounds_all_finite && L.exact_fixed_point(absmax, nzmin, n_rows, &e) && L.lit_f(std::ldexp(1.0, -e), &scale_lit) == 0) {
        o.fixed_point = true;
        o.fixed_exp = e;
        return {"SumF64Q2<" + arg + "," + scale_lit + ">", {ADD_I64, ADD_I64}};
      }
      if (L.image_plan && bounded && L.bounds_all_finite && !std::getenv("LLKV_HIP_IMAGE_NO_FIXED") && L.fixed_point_grid(absmax, nzmin, n_rows, &e) &&
          L.lit_f(std::ldexp(1.0, -e), &scale_lit) == 0) { // one integer lane in the image, two in the exchange image
        o.fixed_point = true;
        o.fixed_exp = e;
        p.image_min_grid = 256;
        L.next_expand = {{0, 1}, {0, 2}};
        return {"SumF64Q<" + arg + "," + scale_lit + ">", {ADD_I64, ADD_I64}};
      }
      if (bounded) levels = L.exact_sum_constants(absmax, nzmin, n_rows, c);
      std::string node_x = "SumF64X<" + arg;
      std::vector<uint8_t> lane_ops;
      for (int j = 0; j < levels; ++j) {
        std::string lit;
        if (L.lit_f(c[j], &lit)) { levels = 0; break; }
        node_x += "," + lit;
        lane_ops.push_back(ADD_F64);
      }
      if (levels == 0) {
        no_bound = true;
        return {"SumF64<" + arg + ">", {ADD_F64}};
      }
      o.exact_levels = levels;
      return {node_x + ">", lane_ops};
    };
    // NULL argument rows contribute each lane's identity; one more lane counts the non-NULL rows
    double narrow_bound = -1.0; // (SumI64Fast over a plain integer column: max |v|)
    if (!is_f64 && simple && fast_i64 && simple_ci->dtype == LLKV_DT_INT64)
      narrow_bound = std::max(std::fabs((double)simple_ci->min_i), std::fabs((double)simple_ci->max_i));
    auto add_agg = [&](const std::string &inner, std::vector<uint8_t> lane_ops) {
      if (inner.rfind("SumI64Fast<", 0) == 0 && narrow_bound >= 0.0) L.next_narrow = std::max(1.0, narrow_bound); // (with IfValid: + a count lane)
      if (valid.empty()) { o.lane = add_group(inner, lane_ops); return; }
      const int n_inner = (int)lane_ops.size();
      lane_ops.push_back(ADD_I64);
      if (!L.next_expand.empty()) { // the count of non-NULL rows is one more kernel lane behind the inner ones
        uint8_t kernel_lanes = 0;
        for (auto &x : L.next_expand) kernel_lanes = std::max<uint8_t>(kernel_lanes, (uint8_t)(x.first + 1));
        L.next_expand.emplace_back(kernel_lanes, (uint8_t)0);
      }
      o.lane = add_group("IfValid<" + valid + "," + inner + ">", lane_ops);
      o.count_lane = o.lane + n_inner;
    };
    if (simple && simple_ci->wide128) {
      // Decimal128 values beyond 64 bits (llkv-aggregate/src/lib.rs:925-943: `sum.checked_add(v)` in i128, row by row):
      // four ADD_I64 lanes over the 32-bit limbs (the top one signed) — exact for < 2^31 rows, order-free.  The
      // reference's overflow check is order dependent (a prefix may leave i128 although the total fits): the plan is
      // taken only when rows · max|v| ≤ i128::MAX excludes that.  AVG: half away from zero (:1720-1742).
      if (s.kind == LLKV_AGG_MIN || s.kind == LLKV_AGG_MAX) {
        // MinDecimal128 / MaxDecimal128 (llkv-aggregate/src/lib.rs:1332-1352,1400-1420: i128 min / max over the non-NULL rows).  A
        // 128-bit compare has no order-free lanes, but a column whose values span less than 2^64 — known from staging — needs
        // none: v − min(column) fits 64 bits and is the low halves' wrapping difference, so MAX is one MAX_U64 lane over
        // lo − min_lo and MIN one over max_lo − lo; the host adds the column's min / max back in i128.
        const i128 vmin = (i128)(((u128)simple_ci->wide_min_hi << 64) | simple_ci->wide_min_lo), vmax = (i128)(((u128)simple_ci->wide_max_hi << 64) | simple_ci->wide_max_lo);
        if (vmax < vmin || (u128)(vmax - vmin) >> 64)
          return L.fail(LLKV_UNSUPPORTED, std::string(fn) + " over Decimal128 values beyond 64 bits that span 2^64 or more (a 128-bit compare has no order-free lanes) is not on the GPU path");
        int lo, hi;
        if ((rc = L.wide_slots_of(s.expr[0].field_id, &lo, &hi, /*want_hi=*/false))) return rc;
        const bool is_min = s.kind == LLKV_AGG_MIN;
        std::string base;
        if ((rc = L.lit_i((int64_t)(is_min ? simple_ci->wide_max_lo : simple_ci->wide_min_lo), &base, "LitU"))) return rc;
        o.precision = simple_ci->precision; o.scale = simple_ci->scale;
        o.wide = true;
        o.wide_delta = is_min ? 2 : 1;
        o.wide_base_hi = is_min ? simple_ci->wide_max_hi : simple_ci->wide_min_hi;
        o.wide_base_lo = is_min ? simple_ci->wide_max_lo : simple_ci->wide_min_lo;
        o.fin = is_min ? AggFinal::MinDec : AggFinal::MaxDec;
        add_agg("MaxWideDelta<" + std::to_string(lo) + "," + base + "," + (is_min ? "1" : "0") + ">", {MAX_U64});
        p.aggs.push_back(o);
        continue;
      }
      if (s.kind != LLKV_AGG_SUM && s.kind != LLKV_AGG_TOTAL && s.kind != LLKV_AGG_AVG) return L.fail(LLKV_UNSUPPORTED, "aggregate kind " + std::to_string(s.kind));
      const u128 absmax = ((u128)simple_ci->wide_absmax_hi << 64) | simple_ci->wide_absmax_lo;
      const u128 i128_max = ~(u128)0 >> 1;
      if (simple_ci->rows >= (1ull << 31) || (absmax != 0 && (u128)simple_ci->rows > i128_max / absmax))
        return L.fail(LLKV_UNSUPPORTED, "possible Decimal128 sum overflow (rows · max|v| exceeds i128): the reference's check is order dependent");
      int lo, hi;
      if ((rc = L.wide_slots_of(s.expr[0].field_id, &lo, &hi))) return rc;
      o.precision = simple_ci->precision; o.scale = simple_ci->scale;
      o.wide = true;
      o.fin = s.kind == LLKV_AGG_SUM ? AggFinal::SumDec : s.kind == LLKV_AGG_TOTAL ? AggFinal::TotalDec : AggFinal::AvgDec;
      add_agg("SumDecWide<" + std::to_string(lo) + "," + std::to_string(hi) + ">", {ADD_I64, ADD_I64, ADD_I64, ADD_I64});
      p.aggs.push_back(o);
      continue;
    }
    if (simple && simple_ci->dtype == LLKV_DT_DECIMAL128) {
      // Decimal128 accumulators (llkv-aggregate/src/lib.rs:925-967,1071-1088,1236-1259,1332-1352,1400-1420) over the
      // 64-bit image: the same exact lanes as Int64, finalized in i128 with the column's (precision, scale)
      o.precision = simple_ci->precision; o.scale = simple_ci->scale;
      o.fast_sum = fast_i64;
      switch (s.kind) {
      case LLKV_AGG_SUM: case LLKV_AGG_TOTAL: case LLKV_AGG_AVG:
        o.fin = s.kind == LLKV_AGG_SUM ? AggFinal::SumDec : s.kind == LLKV_AGG_TOTAL ? AggFinal::TotalDec : AggFinal::AvgDec;
        if (fast_i64) add_agg("SumI64Fast<" + node + ">", {ADD_I64});
        else add_agg("SumI64<" + node + ">", {ADD_I64, ADD_I64, MAX_U64});
        break;
      case LLKV_AGG_MIN: o.fin = AggFinal::MinDec; add_agg("MinI64<" + node + ">", {MIN_I64}); break;
      case LLKV_AGG_MAX: o.fin = AggFinal::MaxDec; add_agg("MaxI64<" + node + ">", {MAX_I64}); break;
      default: return L.fail(LLKV_UNSUPPORTED, "aggregate kind " + std::to_string(s.kind));
      }
      p.aggs.push_back(o);
      continue;
    }
    if (pvi.is_decimal) {
      // Decimal128 accumulators over the computed argument's 64-bit image; precision = digits of the group's first non-NULL
      // value (filled in at finalize from the FirstDigits lane), scale = the expression's
      o.precision = 0; o.scale = pvi.scale;
      const u128 m = (u128)std::max(pvi.lo < 0 ? -pvi.lo : pvi.lo, pvi.hi < 0 ? -pvi.hi : pvi.hi);
      const uint64_t rows = pending_digits.back().rows;
      o.fast_sum = rows != 0 && m * (u128)rows <= (u128)INT64_MAX;
      switch (s.kind) {
      case LLKV_AGG_SUM: case LLKV_AGG_TOTAL: case LLKV_AGG_AVG:
        o.fin = s.kind == LLKV_AGG_SUM ? AggFinal::SumDec : s.kind == LLKV_AGG_TOTAL ? AggFinal::TotalDec : AggFinal::AvgDec;
        if (o.fast_sum) add_agg("SumI64Fast<" + node + ">", {ADD_I64});
        else add_agg("SumI64<" + node + ">", {ADD_I64, ADD_I64, MAX_U64});
        break;
      case LLKV_AGG_MIN: o.fin = AggFinal::MinDec; add_agg("MinI64<" + node + ">", {MIN_I64}); break;
      case LLKV_AGG_MAX: o.fin = AggFinal::MaxDec; add_agg("MaxI64<" + node + ">", {MAX_I64}); break;
      default: return L.fail(LLKV_UNSUPPORTED, "aggregate kind " + std::to_string(s.kind));
      }
      p.aggs.push_back(o);
      continue;
    }
    // MinFloat64 / MaxFloat64 (llkv-aggregate/src/lib.rs:1309-1331,1377-1399) fold sequentially by partial_cmp: a leading NaN sticks,
    // ±0 ties keep the earlier row — which costs two row-order lanes beside the order key.  A bare Float64 column whose staging
    // statistics say "no NaN / ±∞, no −0.0" has neither case: one order-key lane (a third of the DS instructions)
    const bool plain_f64 = simple && simple_ci->dtype == LLKV_DT_FLOAT64 && simple_ci->has_fstats && simple_ci->f_all_finite && simple_ci->f_no_neg_zero &&
                           !std::getenv("LLKV_HIP_MINMAX_ROW_ORDER");
    switch (s.kind) {
    case LLKV_AGG_SUM:
      if (is_f64) { o.fin = AggFinal::SumF64; auto g = sum_f64(node); add_agg(g.first, g.second); }
      else if (fast_i64) { o.fin = AggFinal::SumI64Fast; add_agg("SumI64Fast<" + node + ">", {ADD_I64}); }
      else { o.fin = AggFinal::SumI64; add_agg("SumI64<" + node + ">", {ADD_I64, ADD_I64, MAX_U64}); }
      break;
    case LLKV_AGG_TOTAL: {
      o.fin = AggFinal::TotalF64;
      auto g = sum_f64(is_f64 ? node : "ToF64<" + node + ">");
      add_agg(g.first, g.second);
      break;
    }
    case LLKV_AGG_AVG:
      if (is_f64) { o.fin = AggFinal::AvgF64; auto g = sum_f64(node); add_agg(g.first, g.second); }
      else if (fast_i64) { o.fin = AggFinal::AvgI64Fast; add_agg("SumI64Fast<" + node + ">", {ADD_I64}); }
      else { o.fin = AggFinal::AvgI64; add_agg("SumI64<" + node + ">", {ADD_I64, ADD_I64, MAX_U64}); }
      break;
    case LLKV_AGG_MIN:
      if (is_f64 && plain_f64) { o.fin = AggFinal::MinF64; o.plain_minmax = true; add_agg("MinF64P<" + node + ">", {MIN_I64}); }
      else if (is_f64) { o.fin = AggFinal::MinF64; add_agg("MinF64<" + node + ">", {MIN_I64, MIN_I64, MIN_I64}); }
      else { o.fin = AggFinal::MinI64; add_agg("MinI64<" + node + ">", {MIN_I64}); }
      break;
    case LLKV_AGG_MAX:
      if (is_f64 && plain_f64) { o.fin = AggFinal::MaxF64; o.plain_minmax = true; add_agg("MaxF64P<" + node + ">", {MAX_I64}); }
      else if (is_f64) { o.fin = AggFinal::MaxF64; add_agg("MaxF64<" + node + ">", {MAX_I64, MIN_I64, MIN_I64}); }
      else { o.fin = AggFinal::MaxI64; add_agg("MaxI64<" + node + ">", {MAX_I64}); }
      break;
    default: return L.fail(LLKV_UNSUPPORTED, "aggregate kind " + std::to_string(s.kind));
    }
    if (no_bound) return L.fail(LLKV_UNSUPPORTED, std::string("the column statistics do not bound an f64 sum argument from above and (where non-zero) from below: no exact, order-free sum for ") + (L.image_plan ? "the shared-image GROUP BY" : "the exact-sum option"));
    p.aggs.push_back(o);
  }

  // FirstDigits lanes: per validity, the distinct argument nodes in packs of up to four 6-bit digit fields under the row id
  // (the key must stay below 2^63: four fields leave 39 bits for the row id, one field 57)
  std::vector<bool> placed(pending_digits.size(), false);
  for (size_t i = 0; i < pending_digits.size(); ++i) {
    if (placed[i]) continue;
    std::vector<size_t> nodes; // indices of the first pending entry of every distinct node in this pack
    uint64_t rows = 0;
    for (size_t j = i; j < pending_digits.size(); ++j) rows = std::max(rows, pending_digits[j].rows);
    const size_t per_pack = rows < (1ull << 38) ? 4 : 1;
    for (size_t j = i; j < pending_digits.size(); ++j) {
      if (placed[j] || pending_digits[j].valid != pending_digits[i].valid) continue;
      bool known = false;
      for (size_t k : nodes) known |= pending_digits[k].node == pending_digits[j].node;
      if (!known) { if (nodes.size() == per_pack) continue; nodes.push_back(j); }
      placed[j] = true;
    }
    std::string g = "FirstDigits<" + pending_digits[i].valid;
    for (size_t k : nodes) g += ",DecDigits<" + pending_digits[k].node + "," + std::to_string(pending_digits[k].dlo) + "," + std::to_string(pending_digits[k].dhi) + ">";
    g += ">";
    const int lane = add_group(g, {MIN_I64});
    for (size_t j = i; j < pending_digits.size(); ++j) {
      if (!placed[j] || pending_digits[j].valid != pending_digits[i].valid || p.aggs[pending_digits[j].agg].digits_lane >= 0) continue;
      for (size_t at = 0; at < nodes.size(); ++at)
        if (pending_digits[nodes[at]].node == pending_digits[j].node) {
          p.aggs[pending_digits[j].agg].digits_lane = lane;
          p.aggs[pending_digits[j].agg].digits_shift = 6 * (int)(nodes.size() - 1 - at);
          p.aggs[pending_digits[j].agg].scale = pending_digits[j].scale;
        }
    }
  }
  return LLKV_OK;
}

static std::atomic<bool> g_exact_f64_sums{false};
void plan_set_exact_f64_sums(bool on) { g_exact_f64_sums.store(on); }
bool plan_exact_f64_sums() { return g_exact_f64_sums.load(); }

int lower_plan(const ColumnResolver &resolve, const llkv_filter *filters, uint32_t n_filters,
               const llkv_eval_op *ops, uint32_t n_ops, const uint32_t *key_fields, uint32_t n_keys,
               const llkv_aggregate_spec *aggs, uint32_t n_aggs, bool grouped, bool track_first, LoweredPlan *out, std::string *err, bool image, bool partitioned) {
  *out = LoweredPlan{};
  LoweredPlan &p = *out;
  Lowering L{resolve, p, err, grouped};
  p.grouped = grouped;
  if (image && !grouped) return L.fail(LLKV_INVALID_ARGUMENT, "the shared-image kernel serves GROUP BY plans");
  if (partitioned && !image) return L.fail(LLKV_INVALID_ARGUMENT, "the partitioned route uses the shared-image lowering");
  L.strict_exact = plan_exact_f64_sums();
  L.image_plan = image;
  L.exact_f64 = image || L.strict_exact;
  L.whole_table_image = partitioned;
  const uint32_t max_groups = partitioned ? kMaxPartGroups : image ? kMaxImageGroups : kMaxDenseGroups;
  int rc;
  if (n_aggs == 0 && !grouped) return L.fail(LLKV_INVALID_ARGUMENT, "aggregate query requires at least one aggregate expression");
  if (grouped && n_keys == 0) return L.fail(LLKV_INVALID_ARGUMENT, "GROUP BY requires at least one key");

  std::string pred;
  if ((rc = L.predicate(filters, n_filters, ops, n_ops, &pred))) return rc;
  p.always_false = pred == "False";

  // keys
  std::string keys = "Keys<";
  if (grouped) {
    if (n_keys > (uint32_t)kMaxKeysHost) return L.fail(LLKV_UNSUPPORTED, "more than 4 GROUP BY keys");
    uint64_t ng = 1;
    std::string nodes;
    for (uint32_t k = 0; k < n_keys; ++k) {
      const ColumnInfo *ci;
      int slot;
      const ColumnInfo *probe = resolve(key_fields[k]);
      if (!probe) return L.fail(LLKV_INVALID_ARGUMENT, "column '" + std::to_string(key_fields[k]) + "' not found in GROUP BY input");
      if (probe->dtype == LLKV_DT_FLOAT64 || probe->dtype == LLKV_DT_FLOAT32 || probe->dtype == LLKV_DT_DECIMAL128)
        return L.fail(LLKV_INVALID_ARGUMENT, std::string("GROUP BY does not support column type ") + dtype_name(probe->dtype));
      // GroupKeyValue (llkv-executor/src/lib.rs:99-106, 9362-9456): Utf8 → String, every integer width and
      // Date32 → Int.  Dense ids come from the dictionary code, or from value − column minimum for
      // integer columns whose staging statistics bound the range.
      const bool int_key = probe->dtype == LLKV_DT_INT64 || probe->dtype == LLKV_DT_INT32 || probe->dtype == LLKV_DT_DATE32;
      if (probe->dtype != LLKV_DT_UTF8 && !int_key) return L.fail(LLKV_UNSUPPORTED, std::string("dense GROUP BY over ") + dtype_name(probe->dtype));

      if (int_key) {
        if (!probe->has_stats) return L.fail(LLKV_UNSUPPORTED, "integer GROUP BY key without column statistics (hash path)");
        const unsigned __int128 range = (unsigned __int128)((__int128)probe->max_i - (__int128)probe->min_i) + 1;
        if (range > (image ? max_groups : 256u)) return L.fail(LLKV_UNSUPPORTED, "integer GROUP BY key spans more than " + std::to_string(image ? max_groups : 256u) + " values (sort-based route)");
      }
      if ((rc = L.slot_of(key_fields[k], &ci, &slot))) return rc;
      L.table_rows = std::max(L.table_rows, ci->rows);
      uint32_t card;
      std::string node;
      if (int_key) {
        card = (uint32_t)((__int128)ci->max_i - (__int128)ci->min_i + 1);
        std::string base;
        if ((rc = L.lit_i(ci->min_i, &base))) return rc;
        node = "KeyInt<" + std::to_string(slot) + "," + dtype_tag(ci->dtype) + "," + base + ">";
      } else {
        card = (uint32_t)(ci->dictionary.empty() ? 1 : ci->dictionary.size());
        node = "KeyCode<" + std::to_string(slot) + ">";
      }
      // GroupKeyValue::Null is a group of its own (llkv-executor/src/lib.rs:99-106,9362-9456): one more code
      std::string kv;
      if ((rc = L.valid_of_field(key_fields[k], &kv))) return rc;
      if (!kv.empty()) { node = "KeyOrNull<" + kv + "," + node + "," + std::to_string(card) + ">"; card += 1; }
      p.key_nullable.push_back(kv.empty() ? 0 : 1);
      nodes += "," + node;
      p.key_fields.push_back(key_fields[k]);
      p.key_slots.push_back((uint32_t)slot);
      p.key_cards.push_back(card);
      p.key_bases.push_back(int_key ? ci->min_i : 0);
      p.key_is_int.push_back(int_key ? 1 : 0);
      ng *= card;
      if (ng > max_groups) return L.fail(LLKV_UNSUPPORTED, "more than " + std::to_string(max_groups) + " dense groups");
    }
    p.ng = (uint32_t)ng;
    p.key_strides.assign(n_keys, 1);
    for (int k = (int)n_keys - 2; k >= 0; --k) p.key_strides[k] = p.key_strides[k + 1] * p.key_cards[k + 1];
    keys += std::to_string(p.ng) + "," + (track_first ? "1" : "0") + nodes + ">";
  } else {
    p.ng = 1;
    keys += "1,0>";
  }

  // aggregates → deduplicated lane groups
  const int base = (grouped && track_first) ? 2 : 1;
  std::vector<std::string> groups; // lane-group node strings
  std::vector<std::vector<uint8_t>> group_ops;
  int next_lane = 0;
  const size_t early_slots = p.slot_fields.size(); // what the predicate and the keys read; the slots behind feed aggregate arguments only
  if ((rc = lower_aggregates(L, resolve, aggs, n_aggs, grouped, groups, group_ops, next_lane))) return rc;
  enum { ADD_F64 = 0, ADD_I64 = 1, MIN_I64 = 2, MAX_I64 = 3, MAX_U64 = 4 };

  p.k = base + next_lane;
  p.lanes = (int)p.ng * p.k + 1;
  // the kernel's lanes per group (shared-image plans may keep a lane group in fewer lanes than the exchange image has)
  p.image_src.clear();
  p.image_xf.clear();
  for (int j = 0; j < base; ++j) { p.image_src.push_back((uint8_t)j); p.image_xf.push_back(0); }
  int kernel_lane = base;
  for (auto &ge : L.group_expand) {
    int used = 0;
    for (auto &x : ge) {
      p.image_src.push_back((uint8_t)(kernel_lane + x.first));
      p.image_xf.push_back(x.second);
      used = std::max(used, x.first + 1);
    }
    kernel_lane += used;
  }
  p.k_image = kernel_lane;
  p.lane_ops.clear();
  for (uint32_t g = 0; g < p.ng; ++g) {
    p.lane_ops.push_back(ADD_I64);
    if (grouped && track_first) p.lane_ops.push_back(MIN_I64);
    for (auto &go : group_ops) for (uint8_t op : go) p.lane_ops.push_back(op);
  }
  p.lane_ops.push_back(MAX_U64);
  // accumulator placement: grouped plans keep their state in per-thread LDS slots (DS atomics
  // indexed by the row's group id); ungrouped plans keep it in registers
  p.acc_image = image;
  p.acc_lds = grouped && p.ng > 1 && !image;
  p.track_first = grouped && track_first;
  if (p.acc_lds && (size_t)(p.lanes - 1) * 2048 > 160u * 1024) // one 2 KiB row per group-state lane (the error lane lives in registers)
    return L.fail(LLKV_UNSUPPORTED, "dense group state does not fit the LDS (" + std::to_string(p.lanes) + " lanes)");
  p.acc_part = partitioned;
  // 4-byte cells: every lane a count or a bounded integer sum (and the first-row lane over fewer than 2^32 rows), and what ONE
  // workgroup image can add up — it sees at most rows / 128 + 16 384 rows (fixed_point_grid) — stays below 2^31
  bool narrow = false;
  if (image && !partitioned && !std::getenv("LLKV_HIP_IMAGE_WIDE_CELLS")) {
    uint64_t rows = 0;
    for (uint32_t k = 0; k < n_keys; ++k) if (const ColumnInfo *ci = resolve(key_fields[k])) rows = std::max(rows, ci->rows);
    double bound = 1.0;
    narrow = rows > 0 && rows < (1ull << 32) && L.group_narrow.size() == groups.size();
    for (double b : L.group_narrow) { narrow = narrow && b >= 0.0; bound = std::max(bound, b); }
    narrow = narrow && bound * ((double)(rows / 128 + 16384)) < 2147483648.0;
    if (narrow) p.image_min_grid = 256;
  }
  p.image_cell32 = narrow;
  if (image && !partitioned) {
    p.image_passes = (int)(((size_t)p.ng * p.k_image * (narrow ? 4 : 8) + kMaxImageBytes - 1) / kMaxImageBytes);
    if (p.image_passes < 1) p.image_passes = 1;
    if (p.image_passes > kMaxImagePasses)
      return L.fail(LLKV_UNSUPPORTED, "the group image (" + std::to_string(p.ng) + " groups × " + std::to_string(p.k_image) + " lanes) needs more than " +
                                           std::to_string(kMaxImagePasses) + " LDS-sized slices");
  }
  p.unroll = image ? 2 : (p.acc_lds || p.lanes <= 8) ? 4 : 2;
  if (const char *e = std::getenv("LLKV_HIP_UNROLL")) { // tuning knob (run-time specialised kernels only)
    const int u = std::atoi(e);
    if (u == 1 || u == 2 || u == 4 || u == 8) p.unroll = u;
  }

  std::string cols = "Cols<";
  p.bytes_per_row = 0;
  for (size_t i = 0; i < p.slot_dtypes.size(); ++i) {
    cols += (i ? "," : "") + std::string(dtype_tag(p.slot_dtypes[i]));
    p.bytes_per_row += dtype_width(p.slot_dtypes[i]);
  }
  cols += ">";
  std::string ag = "Aggs<";
  for (size_t i = 0; i < groups.size(); ++i) ag += (i ? "," : "") + groups[i];
  ag += ">";
  // register-state plans with a predicate: the argument-only columns are read for the rows that pass (fused_scan.hip.h:
  // Plan::EARLY, late materialisation — Q6 reads 20 B of every row and the price of the 2 % that pass)
  const bool late = !p.acc_part && !p.acc_image && !p.acc_lds && pred != "True" && pred != "False" && early_slots > 0 && early_slots < p.slot_fields.size() &&
                    !std::getenv("LLKV_HIP_SCAN_NO_LATE");
  p.late_columns = late;
  p.type_string = "Plan<" + cols + "," + pred + "," + keys + "," + ag + "," + std::to_string(p.unroll) + "," + (p.acc_part ? "3" : p.acc_image ? "2" : p.acc_lds ? "1" : "0") +
                  (late ? ",1," + std::to_string(early_slots) : p.image_cell32 ? "," + std::to_string(p.image_passes) + ",-1,1" : p.image_passes > 1 ? "," + std::to_string(p.image_passes) : std::string()) + ">";
  return LLKV_OK;
}

static std::string cols_string(const LoweredPlan &p, uint64_t *bytes) {
  std::string cols = "Cols<";
  uint64_t b = 0;
  for (size_t i = 0; i < p.slot_dtypes.size(); ++i) {
    cols += (i ? "," : "") + std::string(dtype_tag(p.slot_dtypes[i]));
    b += dtype_width(p.slot_dtypes[i]);
  }
  if (bytes) *bytes = b;
  return cols + ">";
}

int lower_reduce(const ColumnResolver &resolve, const llkv_aggregate_spec *aggs, uint32_t n_aggs, LoweredPlan *out, std::string *err) {
  *out = LoweredPlan{};
  LoweredPlan &p = *out;
  Lowering L{resolve, p, err, true};
  L.allow_dict_num = false;
  L.allow_sorted_distinct = true;
  p.grouped = true;
  p.track_first = true;
  std::vector<std::string> groups;
  std::vector<std::vector<uint8_t>> group_ops;
  int next_lane = 0, rc;
  if ((rc = lower_aggregates(L, resolve, aggs, n_aggs, true, groups, group_ops, next_lane))) return rc;
  p.ng = 1;
  p.k = 2 + next_lane;
  p.lanes = p.k;
  p.lane_ops = {1 /*ADD_I64 rows*/, 2 /*MIN_I64 first row*/};
  for (auto &go : group_ops) for (uint8_t op : go) p.lane_ops.push_back(op);
  std::string ag = "Aggs<";
  for (size_t i = 0; i < groups.size(); ++i) ag += (i ? "," : "") + groups[i];
  p.type_string = "ReducePlan<" + cols_string(p, &p.bytes_per_row) + "," + ag + ">>";
  return LLKV_OK;
}

int lower_selection(const ColumnResolver &resolve, const llkv_filter *filters, uint32_t n_filters,
                    const llkv_eval_op *ops, uint32_t n_ops, const uint32_t *drop_null_fields, uint32_t n_drop_null_fields,
                    LoweredPlan *out, std::string *err) {
  *out = LoweredPlan{};
  Lowering L{resolve, *out, err, false};
  std::string pred;
  int rc = L.predicate(filters, n_filters, ops, n_ops, &pred);
  if (rc) return rc;
  // GatherNullPolicy::DropNulls (llkv-column-map/src/store/projection.rs:40-48,1326-1330): a row is dropped when
  // every gathered field is NULL — possible only if all of them have NULL cells
  if (n_drop_null_fields) {
    std::vector<std::string> vs;
    bool some_never_null = false;
    for (uint32_t i = 0; i < n_drop_null_fields; ++i) {
      std::string v;
      if ((rc = L.valid_of_field(drop_null_fields[i], &v))) return rc;
      if (v.empty()) some_never_null = true;
      else if (std::find(vs.begin(), vs.end(), v) == vs.end()) vs.push_back(v);
    }
    if (!some_never_null && pred != "False") {
      const std::string any = vs.size() == 1 ? vs[0] : Lowering::nary("Or", vs);
      pred = pred == "True" ? any : "And<" + pred + "," + any + ">";
    }
  }
  out->always_false = pred == "False";
  out->always_true = pred == "True";
  out->type_string = "SelPlan<" + cols_string(*out, &out->bytes_per_row) + "," + pred + ">";
  return LLKV_OK;
}

int lower_selection_in_set(const ColumnResolver &resolve, const llkv_filter *filters, uint32_t n_filters, uint32_t key_field,
                           LoweredPlan *out, std::string *err) {
  *out = LoweredPlan{};
  Lowering L{resolve, *out, err, false};
  std::string pred, key;
  int rc = L.predicate(filters, n_filters, nullptr, 0, &pred);
  if (rc) return rc;
  const ColumnInfo *ci;
  int slot;
  if ((rc = L.slot_of(key_field, &ci, &slot))) return rc;
  if (ci->nullable) return L.fail(LLKV_UNSUPPORTED, "NULL join keys in the join-aggregate pipeline");
  if (ci->dtype == LLKV_DT_INT64 || ci->dtype == LLKV_DT_UINT64) key = L.col_node(slot, LLKV_DT_INT64);
  else if (ci->dtype == LLKV_DT_INT32 || ci->dtype == LLKV_DT_DATE32 || ci->dtype == LLKV_DT_UINT32) key = "ToI64<" + L.col_node(slot, ci->dtype) + ">";
  else return L.fail(LLKV_UNSUPPORTED, std::string("join key of type ") + dtype_name(ci->dtype));
  out->always_false = pred == "False";
  const std::string in_set = "InKeySet<" + key + ">";
  if (!out->always_false) pred = pred == "True" ? in_set : "AndThen<" + pred + "," + in_set + ">";
  out->type_string = "SelPlan<" + cols_string(*out, &out->bytes_per_row) + "," + pred + ">";
  return LLKV_OK;
}

int lower_emit(const ColumnResolver &resolve, const llkv_filter *filters, uint32_t n_filters, const llkv_eval_op *ops,
               uint32_t n_ops, const llkv_expr_token *expr, uint32_t expr_len, LoweredPlan *out, std::string *err,
               bool allow_f64, bool *is_f64_out, const uint32_t *in_set_field, int32_t *key_dtype, bool int32_value) {
  *out = LoweredPlan{};
  Lowering L{resolve, *out, err, false};
  std::string pred, val;
  int rc = L.predicate(filters, n_filters, ops, n_ops, &pred);
  if (rc) return rc;
  out->always_false = pred == "False";
  if (in_set_field && !out->always_false) { // … AND the key of this field is in the launch's key set (a semi join as a conjunct)
    const ColumnInfo *ci;
    int slot;
    std::string key;
    if ((rc = L.slot_of(*in_set_field, &ci, &slot))) return rc;
    if (ci->nullable) return L.fail(LLKV_UNSUPPORTED, "NULL join keys in the join-aggregate pipeline");
    if (ci->dtype == LLKV_DT_INT64 || ci->dtype == LLKV_DT_UINT64) key = L.col_node(slot, LLKV_DT_INT64);
    else if (ci->dtype == LLKV_DT_INT32 || ci->dtype == LLKV_DT_DATE32 || ci->dtype == LLKV_DT_UINT32) key = "ToI64<" + L.col_node(slot, ci->dtype) + ">";
    else return L.fail(LLKV_UNSUPPORTED, std::string("join key of type ") + dtype_name(ci->dtype));
    const std::string in_set = "InKeySet<" + key + ">";
    pred = pred == "True" ? in_set : "AndThen<" + pred + "," + in_set + ">";
  }
  bool is_f64 = false;
  if (expr_len == 1 && expr[0].kind == LLKV_TOK_COLUMN) {
    const ColumnInfo *ci;
    int slot;
    if ((rc = L.slot_of(expr[0].field_id, &ci, &slot))) return rc;
    // DISTINCT keys (DistinctKey::from_array llkv-aggregate/src/lib.rs:261-331): Int by value, Float by bits — and, where the
    // caller takes them (key_dtype), Str by its dictionary code (the staged dictionary holds every string once), Bool,
    // Date and the 64-bit image of a Decimal by value
    const bool keyed = key_dtype && (ci->dtype == LLKV_DT_UTF8 || ci->dtype == LLKV_DT_BOOLEAN || ci->dtype == LLKV_DT_DATE32 || (ci->dtype == LLKV_DT_DECIMAL128 && !ci->wide128));
    const bool narrow = int32_value && ci->dtype == LLKV_DT_INT32;
    if (ci->dtype != LLKV_DT_INT64 && !(allow_f64 && ci->dtype == LLKV_DT_FLOAT64) && !keyed && !narrow) return L.fail(LLKV_UNSUPPORTED, std::string("value emission over a ") + dtype_name(ci->dtype) + " column");
    val = L.col_node(slot, ci->dtype);
    if ((keyed && ci->dtype != LLKV_DT_DECIMAL128) || narrow) val = "ToI64<" + val + ">";
    is_f64 = ci->dtype == LLKV_DT_FLOAT64;
    if (key_dtype) *key_dtype = ci->dtype;
  } else {
    L.exact_nan = true; // the emitted values are told apart by their bits (DISTINCT)
    rc = L.expr_fast(expr, expr_len, &val, &is_f64);
    L.exact_nan = false;
    if (rc) return rc;
    if (L.last_fast_32) return L.fail(LLKV_UNSUPPORTED, "aggregate over a 32-bit-only integer expression (the reference has no Int32 accumulator)");
    if (is_f64 && !allow_f64) return L.fail(LLKV_INTERNAL, "exact sum check over a float expression");
    if (key_dtype) *key_dtype = is_f64 ? LLKV_DT_FLOAT64 : LLKV_DT_INT64;
  }
  if (is_f64_out) *is_f64_out = is_f64;
  { // NULL argument rows are not part of the accumulator's chain
    std::string v;
    if ((rc = L.valid_of_node(expr, expr_len, val, false, &v))) return rc;
    if (!v.empty() && pred != "False") pred = pred == "True" ? v : "And<" + pred + "," + v + ">";
  }
  out->type_string = "EmitPlan<" + cols_string(*out, &out->bytes_per_row) + "," + pred + "," + val + ">";
  return LLKV_OK;
}

int lower_probe(const ColumnResolver &resolve, const llkv_filter *filters, uint32_t n_filters, uint32_t key_field,
                const llkv_expr_token *expr, uint32_t expr_len, LoweredPlan *out, std::string *err, bool emit_keybit) {
  *out = LoweredPlan{};
  Lowering L{resolve, *out, err, true};
  std::string pred, key, val;
  int rc = L.predicate(filters, n_filters, nullptr, 0, &pred);
  if (rc) return rc;
  out->always_false = pred == "False";
  const ColumnInfo *ci;
  int slot;
  if ((rc = L.slot_of(key_field, &ci, &slot))) return rc;
  if (ci->nullable) return L.fail(LLKV_UNSUPPORTED, "NULL join keys in the join-aggregate pipeline");
  for (uint32_t i = 0; i < expr_len; ++i)
    if (expr && expr[i].kind == LLKV_TOK_COLUMN && resolve(expr[i].field_id) && resolve(expr[i].field_id)->nullable)
      return L.fail(LLKV_UNSUPPORTED, "NULL aggregate arguments in the join-aggregate pipeline");
  if (ci->dtype == LLKV_DT_INT64 || ci->dtype == LLKV_DT_UINT64) key = L.col_node(slot, LLKV_DT_INT64);
  else if (ci->dtype == LLKV_DT_INT32 || ci->dtype == LLKV_DT_DATE32 || ci->dtype == LLKV_DT_UINT32) key = "ToI64<" + L.col_node(slot, ci->dtype) + ">";
  else return L.fail(LLKV_UNSUPPORTED, std::string("join key of type ") + dtype_name(ci->dtype));
  bool is_f64 = false;
  if (!expr || expr_len == 0) return L.fail(LLKV_INVALID_ARGUMENT, "aggregate requires an argument");
  const size_t early = out->slot_fields.size(); // the slots so far feed the predicate and the key; what the value adds is read late
  if ((rc = L.expr_planvalue(expr, expr_len, &val, &is_f64))) return rc;
  if (!is_f64) return L.fail(LLKV_UNSUPPORTED, "integer SUM in the join-aggregate pipeline");
  const bool late = early < out->slot_fields.size() && !std::getenv("LLKV_HIP_JOIN_NO_LATE");
  // (the direct-table probe of a plan with KEYBIT = 1 emits the key's bit position instead of its rank: join_agg.cpp)
  const std::string tail = emit_keybit ? "," + std::to_string(late ? early : out->slot_fields.size()) + ",1" : late ? "," + std::to_string(early) : "";
  out->type_string = "ProbePlan<" + cols_string(*out, &out->bytes_per_row) + "," + pred + "," + key + "," + val + tail + ">";
  return LLKV_OK;
}

int lower_projection(const ColumnResolver &resolve, const llkv_projection *projections, uint32_t n_projections,
                     LoweredPlan *out, std::string *err, bool pad_rows) {
  *out = LoweredPlan{};
  Lowering L{resolve, *out, err, false};
  if (n_projections == 0) return L.fail(LLKV_INVALID_ARGUMENT, "scan requires at least one projection");
  if (n_projections > 8) return L.fail(LLKV_UNSUPPORTED, "more than 8 projections");
  std::string outs = "Outs<";
  int rc;
  for (uint32_t i = 0; i < n_projections; ++i) {
    const llkv_projection &pr = projections[i];
    std::string node;
    // a Decimal128 column with values beyond 64 bits passes through as it was staged: low and high halves → 16 bytes
    auto wide_column = [&](uint32_t field, std::string *n, const ColumnInfo **ci_out) -> int {
      const ColumnInfo *ci = resolve(field);
      if (!ci || !ci->wide128) return -1; // not one
      int lo, hi;
      int r = L.wide_slots_of(field, &lo, &hi);
      if (r) return r;
      *n = "Join128<" + std::to_string(lo) + "," + std::to_string(hi) + ">";
      *ci_out = ci;
      return LLKV_OK;
    };
    if (!pr.computed) {
      const ColumnInfo *ci;
      int slot;
      if ((rc = wide_column(pr.field_id, &node, &ci)) > 0) return rc;
      if (rc < 0) {
        if ((rc = L.slot_of(pr.field_id, &ci, &slot))) return rc;
        if (dtype_width(ci->dtype) == 0) return L.fail(LLKV_UNSUPPORTED, std::string("projection of ") + dtype_name(ci->dtype));
        node = L.col_node(slot, ci->dtype);
        if (ci->dtype == LLKV_DT_DECIMAL128) node = "Widen128<" + node + ">"; // back to arrow's 16-byte raw values
      }
      out->out_dtypes.push_back(ci->dtype);
      out->out_fields.push_back((int32_t)pr.field_id);
      std::string v;
      if ((rc = L.valid_of_field(pr.field_id, &v))) return rc;
      if (pad_rows) v = v.empty() ? "RowPresent" : "And<RowPresent," + v + ">"; // NULL padding of a LEFT join (select.hip.h: ProjPlan PAD)
      if (!v.empty()) node = "OutV<" + node + "," + v + ">";
      out->out_nullable.push_back(!v.empty());
    } else {
      if (pad_rows) return L.fail(LLKV_INVALID_ARGUMENT, "join projections cannot include computed columns yet"); // hash_join.rs:822-826
      bool is_f64 = false;
      if (!pr.expr || pr.expr_len == 0) return L.fail(LLKV_INVALID_ARGUMENT, "computed projection without expression");
      if (pr.expr_len == 1 && pr.expr[0].kind == LLKV_TOK_COLUMN) { // bare column written as an expression
        const ColumnInfo *ci;
        int slot;
        if ((rc = wide_column(pr.expr[0].field_id, &node, &ci)) > 0) return rc;
        if (rc < 0) {
          if ((rc = L.slot_of(pr.expr[0].field_id, &ci, &slot))) return rc;
          node = L.col_node(slot, ci->dtype);
          if (ci->dtype == LLKV_DT_DECIMAL128) node = "Widen128<" + node + ">";
        }
        out->out_dtypes.push_back(ci->dtype);
        out->out_fields.push_back((int32_t)pr.expr[0].field_id);
      } else {
        L.exact_nan = true; // the consumer sees the bits of a projected value
        rc = L.expr_fast(pr.expr, pr.expr_len, &node, &is_f64);
        L.exact_nan = false;
        if (rc) return rc;
        if (L.last_fast_32) node = std::string("Narrow32<") + node + (L.last_fast_32 == 1 ? ",I32>" : ",U32>");
        out->out_dtypes.push_back(is_f64 ? LLKV_DT_FLOAT64 : L.last_fast_32 == 1 ? LLKV_DT_INT32 : L.last_fast_32 == 2 ? LLKV_DT_UINT32 : LLKV_DT_INT64);
        out->out_fields.push_back(-1);
      }
      std::string v;
      if ((rc = L.valid_of_node(pr.expr, pr.expr_len, node, false, &v))) return rc;
      if (!v.empty()) node = "OutV<" + node + "," + v + ">";
      out->out_nullable.push_back(!v.empty());
    }
    outs += (i ? "," : "") + node;
  }
  out->type_string = "ProjPlan<" + cols_string(*out, &out->bytes_per_row) + "," + outs + (pad_rows ? ">,1>" : ">>");
  return LLKV_OK;
}

} // namespace llkv
