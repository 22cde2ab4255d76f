// fused_scan.hip.h — device-side building blocks of the fused
// scan → predicate → (dense group id) → aggregate kernel for gfx950 (CDNA4).
//
// One kernel replaces the reference's five CPU passes (SURVEY.md §2 K1–K6):
//   leaf filters      llkv-column-map/src/store/scan/filter.rs:937-955
//   bitmap AND/OR/NOT llkv-scan/src/predicate.rs:87-186
//   gather            llkv-column-map/src/store/projection.rs:929-1352
//   computed exprs    llkv-compute/src/fast_numeric.rs:69-121,312-356
//   accumulate        llkv-aggregate/src/lib.rs:759-1477
//   GROUP BY          llkv-executor/src/lib.rs:5028-5355
// Nothing but the per-tile partial state is ever written back to HBM.
//
// Shape of the work (HBM-bound, no MFMA — there is no contraction here):
//   * a block owns one tile (a run of rows inside one chunk); a thread owns two
//     consecutive rows per step, so every 8-byte column is read with one 16-byte
//     load per lane (1 KiB contiguous per wave instruction), 4-byte columns with 8-byte
//     loads, 1-byte dictionary codes with 2-byte loads;
//   * predicates, expressions and per-group accumulators live in registers; a plan is
//     a C++ type, so the row body is straight-line code;
//   * partial state is a vector of independent 64-bit lanes, each with one of five
//     combine ops, reduced in a fixed order (thread → LDS transpose → 16-lane
//     butterfly → tile partial), which makes f64 sums bit-reproducible and independent
//     of the GPU count (DESIGN.md "Determinism").
//
// The header is self-contained device code (usable from hiprtc).
#pragma once

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "scan_params.h"

namespace llkv {

__device__ __forceinline__ uint64_t lane_identity(int op) {
  switch (op) {
  case OP_MIN_I64: return 0x7FFFFFFFFFFFFFFFull;
  case OP_MAX_I64: return 0x8000000000000000ull;
  default: return 0ull; // +0.0, 0, 0
  }
}

// Branch-free: `op` is a compile-time constant in the row loop (the selects fold away) and
// a per-16-lane-group value in the reduction phases (no divergent branches there).
__device__ __forceinline__ uint64_t lane_combine(int op, uint64_t a, uint64_t b) {
  const uint64_t fadd = (uint64_t)__double_as_longlong(__longlong_as_double((long long)a) + __longlong_as_double((long long)b));
  const uint64_t iadd = a + b;
  const bool lt = (int64_t)b < (int64_t)a;
  const uint64_t imin = lt ? b : a;
  const uint64_t imax = (int64_t)b > (int64_t)a ? b : a;
  const uint64_t umax = b > a ? b : a;
  return op == OP_ADD_F64 ? fadd : op == OP_ADD_I64 ? iadd : op == OP_MIN_I64 ? imin : op == OP_MAX_I64 ? imax : umax;
}

template <int OP> __device__ __forceinline__ uint64_t lane_combine_s(uint64_t a, uint64_t b) {
  if constexpr (OP == OP_ADD_F64) return (uint64_t)__double_as_longlong(__longlong_as_double((long long)a) + __longlong_as_double((long long)b));
  else if constexpr (OP == OP_ADD_I64) return a + b;
  else if constexpr (OP == OP_MIN_I64) return (int64_t)b < (int64_t)a ? b : a;
  else if constexpr (OP == OP_MAX_I64) return (int64_t)b > (int64_t)a ? b : a;
  else return b > a ? b : a;
}

template <int OP> constexpr uint64_t lane_identity_s() {
  return OP == OP_MIN_I64 ? 0x7FFFFFFFFFFFFFFFull : OP == OP_MAX_I64 ? 0x8000000000000000ull : 0ull;
}

// --------------------------------------------------------------------------
// Storage types of staged columns
// --------------------------------------------------------------------------
struct I32 { using T = int32_t; static constexpr int W = 4; static constexpr bool is_float = false; };
struct U32 { using T = uint32_t; static constexpr int W = 4; static constexpr bool is_float = false; };
struct F32 { using T = float; static constexpr int W = 4; static constexpr bool is_float = true; };
struct I64 { using T = int64_t; static constexpr int W = 8; static constexpr bool is_float = false; };
struct U64 { using T = uint64_t; static constexpr int W = 8; static constexpr bool is_float = false; };
struct F64 { using T = double; static constexpr int W = 8; static constexpr bool is_float = true; };
struct I128 { using T = __int128; static constexpr int W = 16; static constexpr bool is_float = false; }; // outputs only
struct U8 { using T = uint8_t; static constexpr int W = 1; static constexpr bool is_float = false; };

template <class... Ts> struct Cols { static constexpr int N = sizeof...(Ts); };

template <int I, class L> struct ColAt;
template <int I, class T0, class... Ts> struct ColAt<I, Cols<T0, Ts...>> { using type = typename ColAt<I - 1, Cols<Ts...>>::type; };
template <class T0, class... Ts> struct ColAt<0, Cols<T0, Ts...>> { using type = T0; };

// Two consecutive rows of every column, as raw dwords (registers once unrolled).
struct Loaded {
  uint32_t w[kMaxCols][4];
};

// LLKV_NT_LOADS: stream the column data with the non-temporal cache policy.  The columns are
// read exactly once per query, so keeping them out of L2/MALL is free and measurably faster on
// MI355X (Q6 SF10: 5.97 → 6.87 TB/s, Q1 SF10: 5.77 → 6.32 TB/s; profiles/r01/sweep_tile_unroll_nt.txt).
#ifndef LLKV_NT_LOADS
#define LLKV_NT_LOADS 1
#endif
template <class V> __device__ __forceinline__ V stream_load(const V *ptr) {
#if LLKV_NT_LOADS
  return __builtin_nontemporal_load(ptr);
#else
  return *ptr;
#endif
}

template <class Ty> __device__ __forceinline__ void load_pair(const void *base, uint64_t row, uint32_t (&w)[4]) {
  if constexpr (Ty::W == 8) {
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u v = stream_load(reinterpret_cast<const v4u *>(static_cast<const char *>(base) + row * 8));
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
  } else if constexpr (Ty::W == 4) {
    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
    const v2u v = stream_load(reinterpret_cast<const v2u *>(static_cast<const char *>(base) + row * 4));
    w[0] = v.x; w[1] = v.y;
  } else {
    w[0] = stream_load(reinterpret_cast<const uint16_t *>(static_cast<const char *>(base) + row));
  }
}

// FOUR consecutive rows of a column no wider than 4 bytes (one 16-byte / 4-byte load: half the vector-memory instructions of two
// pairs — the scans of narrow columns are bound by the address pipeline, profiles/r04/q3_sq_counters.txt); Ctx::get reads rows
// 0 … 3 of such a column as it reads rows 0 and 1.
template <class Ty> __device__ __forceinline__ void load_quad(const void *base, uint64_t row, uint32_t (&w)[4]) {
  static_assert(Ty::W <= 4, "four rows of an 8-byte column do not fit a column's registers");
  if constexpr (Ty::W == 4) {
    typedef uint32_t v4u __attribute__((ext_vector_type(4)));
    const v4u v = stream_load(reinterpret_cast<const v4u *>(static_cast<const char *>(base) + row * 4));
    w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
  } else {
    w[0] = stream_load(reinterpret_cast<const uint32_t *>(static_cast<const char *>(base) + row));
  }
}
template <class CL, int LO, int HI> __device__ __forceinline__ void load_quad_range(const ScanParams &p, uint64_t row, Loaded &ld) {
  if constexpr (LO < HI && LO < CL::N) {
    load_quad<typename ColAt<LO, CL>::type>(p.col[LO], row, ld.w[LO]);
    load_quad_range<CL, LO + 1, HI>(p, row, ld);
  }
}
template <class CL, int LO, int HI> constexpr bool cols_narrow() {
  if constexpr (LO < HI && LO < CL::N) return ColAt<LO, CL>::type::W <= 4 && ColAt<LO, CL>::type::W != 2 && cols_narrow<CL, LO + 1, HI>();
  else return true;
}

template <class CL, int I = 0> __device__ __forceinline__ void load_all(const ScanParams &p, uint64_t row, Loaded &ld) {
  if constexpr (I < CL::N) {
    load_pair<typename ColAt<I, CL>::type>(p.col[I], row, ld.w[I]);
    load_all<CL, I + 1>(p, row, ld);
  }
}

// Columns [LO, HI) only: the plans that read some columns for the rows that pass and no others (late materialisation).
template <class CL, int LO, int HI> __device__ __forceinline__ void load_range(const ScanParams &p, uint64_t row, Loaded &ld) {
  if constexpr (LO < HI && LO < CL::N) {
    load_pair<typename ColAt<LO, CL>::type>(p.col[LO], row, ld.w[LO]);
    load_range<CL, LO + 1, HI>(p, row, ld);
  }
}

// Per-row evaluation context.
struct Ctx {
  const ScanParams &p;
  const Loaded &ld;
  uint32_t err; // sticky: checked integer arithmetic overflowed on a selected row
  uint64_t row; // logical row id of the row being evaluated
  uint32_t perr; // sticky: ... inside the predicate (Expr::Compare sides), which is evaluated on EVERY row
  uint64_t dval; // sort-based GROUP BY with DISTINCT aggregates: the value of their column in this row …
  uint32_t dhead; // … and 1 when the row is the first of its group with that value

  template <class Ty> __device__ __forceinline__ typename Ty::T get(int s, int j) const {
    if constexpr (Ty::W == 8) {
      const uint64_t bits = ((uint64_t)ld.w[s][2 * j + 1] << 32) | ld.w[s][2 * j];
      if constexpr (Ty::is_float) return __longlong_as_double((long long)bits);
      else return (typename Ty::T)bits;
    } else if constexpr (Ty::W == 4) {
      if constexpr (Ty::is_float) return __uint_as_float(ld.w[s][j]);
      else return (typename Ty::T)ld.w[s][j];
    } else {
      return (uint8_t)((ld.w[s][0] >> (8 * j)) & 0xffu);
    }
  }
};

// --------------------------------------------------------------------------
// Scalar expressions (llkv-compute/src/fast_numeric.rs token programs, lowered to
// a type).  The host lowering inserts the casts the reference performs: every column
// is cast to the program's final type first (fast_numeric.rs:80-87).
// --------------------------------------------------------------------------
// Every expression node also answers `valid`: is the value non-NULL in this row?  NULL propagates through
// arithmetic, a division by zero yields NULL (llkv-compute/src/kernels.rs:121-135), and arrow's checked kernels
// evaluate a node only where its operands are valid — so a node raises its arithmetic error only there.
template <int S, class Ty> struct Col {
  using Type = Ty;
  static __device__ __forceinline__ typename Ty::T eval(Ctx &c, int j) { return c.get<Ty>(S, j); }
  static __device__ __forceinline__ bool valid(Ctx &, int) { return true; }
};
template <int S, class Ty, int SV> struct ColN { // column with NULL cells: validity mask in slot SV
  using Type = Ty;
  static __device__ __forceinline__ typename Ty::T eval(Ctx &c, int j) { return c.get<Ty>(S, j); }
  static __device__ __forceinline__ bool valid(Ctx &c, int j) { return c.get<U8>(SV, j) != 0; }
};
template <int K> struct LitI {
  using Type = I64;
  static __device__ __forceinline__ int64_t eval(Ctx &c, int) { return c.p.lit_i[K]; }
  static __device__ __forceinline__ bool valid(Ctx &, int) { return true; }
};
template <int K> struct LitU {
  using Type = U64;
  static __device__ __forceinline__ uint64_t eval(Ctx &c, int) { return (uint64_t)c.p.lit_i[K]; }
  static __device__ __forceinline__ bool valid(Ctx &, int) { return true; }
};
template <int K> struct LitF {
  using Type = F64;
  static __device__ __forceinline__ double eval(Ctx &c, int) { return c.p.lit_f[K]; }
  static __device__ __forceinline__ bool valid(Ctx &, int) { return true; }
};
template <int S> struct DictNum { // numeric value of the dictionary code in slot S (ScanParams::dict_num)
  using Type = F64;
  static __device__ __forceinline__ double eval(Ctx &c, int j) { return c.p.dict_num[S * 256 + (int)c.get<U8>(S, j)]; }
  static __device__ __forceinline__ bool valid(Ctx &, int) { return true; }
};
template <class E> struct ToF64 { // arrow cast int → f64 / f32 → f64
  using Type = F64;
  static __device__ __forceinline__ double eval(Ctx &c, int j) { return (double)E::eval(c, j); }
  static __device__ __forceinline__ bool valid(Ctx &c, int j) { return E::valid(c, j); }
};
template <class E> struct Widen128 { // Decimal128 column staged as i64 → arrow's 16-byte raw value on the way out
  using Type = I128;
  static __device__ __forceinline__ __int128 eval(Ctx &c, int j) { return (__int128)(int64_t)E::eval(c, j); }
  static __device__ __forceinline__ bool valid(Ctx &c, int j) { return E::valid(c, j); }
};
template <int SLO, int SHI> struct Join128 { // Decimal128 column with values beyond 64 bits (staged as low / high halves) → the 16-byte raw value
  using Type = I128;
  static __device__ __forceinline__ __int128 eval(Ctx &c, int j) { return ((__int128)c.get<I64>(SHI, j) << 64) | (__int128)(unsigned __int128)c.get<U64>(SLO, j); }
  static __device__ __forceinline__ bool valid(Ctx &, int) { return true; }
};
template <class E> struct ToI64 { // widening of the narrow integer types
  using Type = I64;
  static __device__ __forceinline__ int64_t eval(Ctx &c, int j) { return (int64_t)E::eval(c, j); }
  static __device__ __forceinline__ bool valid(Ctx &c, int j) { return E::valid(c, j); }
};

// Rust `f64 as i64`: saturating, NaN → 0 (llkv-executor/src/lib.rs:7385-7389).
__device__ __forceinline__ int64_t f64_as_i64_sat(double x) {
  if (x != x) return 0;
  if (x >= 9223372036854775808.0) return 0x7FFFFFFFFFFFFFFFll;
  if (x <= -9223372036854775808.0) return (int64_t)0x8000000000000000ull;
  return (int64_t)x;
}

// Which MIN lanes whose key grows with the row id skip the DS instruction after a thread's first contribution to a group
// (plan_lane_first_only): 0 none (default), 1 the FirstDigits lane of computed decimal arguments, 2 also the first-row lane and the
// first-row / first-zero lanes of the f64 MIN / MAX.  Built in r04 and measured on one box (profiles/r04/first_only_lanes.txt,
// tools/first_only_ab.sh): the skip is an exec-masked region per lane and row, and it does NOT pay — Q1 over DECIMAL(15,2) 365.1 /
// 364.9 / 363.5 µs under modes 0 / 1 / 2, the Float64 Q1 in first-appearance order 368.3 / 369.6 / 365.1 µs, the 12-aggregate wide
// state 337 / 339 / 421 µs (four such lanes beside nine others: +25 %).  Kept behind the switch as the record of the experiment.
#ifndef LLKV_FIRST_ONLY_MODE
#define LLKV_FIRST_ONLY_MODE 0
#endif
enum BinKind : int { B_ADD = 1, B_SUB = 2, B_MUL = 3, B_REM = 4 };
// sticky error codes of Ctx::err / the error lane (combined with max): 1 = arithmetic overflow, 2 = division by zero
constexpr uint32_t kErrOverflow = 1u, kErrDivZero = 2u;

// arrow-arith numeric::{add,sub,mul,rem}: IEEE for floats; integers checked (overflow is an error,
// fast_numeric.rs:328-334 → Error::Internal) except `%`, which is a zero check ("Divide by zero") followed by
// mod_wrapping; a node is evaluated only where both operands are valid.
// The NaN an f64 operation returns, as the reference's host computes it (SSE2 scalar arithmetic): a NaN operand comes
// back quieted WITH ITS SIGN (the first operand's when both are NaN), an invalid operation (∞ − ∞, 0 · ∞, …) gives the
// negative "real indefinite" quiet NaN.  CDNA would hand back a NaN of the other sign in both cases (a − NaN flips the
// operand's sign, invalid operations give +NaN), and arrow compares floats by totalOrder — where −NaN is below and
// +NaN above everything — so the sign of a computed NaN decides `Expr::Compare` and IN-list results.
// (Only where the bits of a NaN can be observed — template flag EXACT_NAN of Bin / Div, set by the lowering for compare /
// IN-list sides, projected and emitted values; aggregate arguments skip it: no accumulator looks at a NaN's sign, and
// the fix-up costs the Q1 kernel 5 %.)
__device__ __forceinline__ double f64_result_as_sse2(double r, double a, double b) {
  if (r == r) return r;
  const long long quiet = 0x0008000000000000ll;
  if (a != a) return __longlong_as_double(__double_as_longlong(a) | quiet);
  if (b != b) return __longlong_as_double(__double_as_longlong(b) | quiet);
  return __longlong_as_double((long long)0xFFF8000000000000ull);
}

template <int OP, class L, class R, int EXACT_NAN = 0> struct Bin {
  using Type = typename L::Type;
  static __device__ __forceinline__ bool valid(Ctx &c, int j) { return L::valid(c, j) & R::valid(c, j); }
  static __device__ __forceinline__ typename Type::T eval(Ctx &c, int j) {
    const auto a = L::eval(c, j);
    const auto b = R::eval(c, j);
    if constexpr (Type::is_float) {
      double r;
      if constexpr (OP == B_ADD) r = a + b;
      else if constexpr (OP == B_SUB) r = a - b;
      else if constexpr (OP == B_MUL) r = a * b;
      else r = fmod(a, b);
      if constexpr (EXACT_NAN) return f64_result_as_sse2(r, a, b);
      else return r;
    } else if constexpr (OP == B_REM) {
      const bool ok = valid(c, j);
      const int64_t x = (int64_t)a, y = (int64_t)b;
      const bool zero = y == 0; // arrow `rem`: zero check, then mod_wrapping (i64::MIN % -1 = 0)
      c.err = max(c.err, (ok & zero) ? kErrDivZero : 0u);
      return (zero | (y == -1)) ? 0 : x % y;
    } else {
      int64_t z;
      bool o;
      if constexpr (OP == B_ADD) o = __builtin_add_overflow((int64_t)a, (int64_t)b, &z);
      else if constexpr (OP == B_SUB) o = __builtin_sub_overflow((int64_t)a, (int64_t)b, &z);
      else o = __builtin_mul_overflow((int64_t)a, (int64_t)b, &z);
      c.err = max(c.err, (o & valid(c, j)) ? kErrOverflow : 0u);
      return z;
    }
  }
};
// A fast-path expression whose root type is Int32 / UInt32 (every leaf a column of that type, no literal: a literal is Int64) runs on
// arrow's checked 32-bit kernels (llkv-compute/src/fast_numeric.rs:312-356).  The operands are within 32 bits, so the 64-bit result of
// + − * is exact and the reference's overflow is "does not fit 32 bits"; rem as Bin's.  Carried in 64 bits; Narrow32 is the output column.
template <class E, int SIGNED> struct Fit32 {
  using Type = I64;
  static __device__ __forceinline__ bool valid(Ctx &c, int j) { return E::valid(c, j); }
  static __device__ __forceinline__ int64_t eval(Ctx &c, int j) {
    const int64_t z = (int64_t)E::eval(c, j);
    const bool fits = SIGNED ? (z >= -2147483648ll && z <= 2147483647ll) : (z >= 0 && z <= 4294967295ll);
    c.err = max(c.err, (!fits & E::valid(c, j)) ? kErrOverflow : 0u);
    return z;
  }
};
template <class E, class T32> struct Narrow32 {
  using Type = T32;
  static __device__ __forceinline__ bool valid(Ctx &c, int j) { return E::valid(c, j); }
  static __device__ __forceinline__ typename T32::T eval(Ctx &c, int j) { return (typename T32::T)E::eval(c, j); }
};
template <class L, class R> using Add = Bin<B_ADD, L, R>;
template <class L, class R> using Sub = Bin<B_SUB, L, R>;
template <class L, class R> using Mul = Bin<B_MUL, L, R>;

// Divide on the generic (per-node typed) path, compute_binary llkv-compute/src/kernels.rs:99-177: zeros of
// the divisor become NULLs first, then arrow `div` — truncating and checked for integers (i64::MIN / -1
// overflows), IEEE for floats.  The operands arrive coerced to their common type.
template <class L, class R, int EXACT_NAN = 0> struct Div {
  using Type = typename L::Type;
  // "zero" is what arrow's `eq` says equals the cast 0: for floats that compare is totalOrder, so +0.0 only — a −0.0
  // divisor is divided by (∓inf, NaN for 0 / −0.0)
  static __device__ __forceinline__ bool nonzero(typename Type::T b) {
    if constexpr (Type::is_float) return __double_as_longlong(b) != 0;
    else return b != 0;
  }
  static __device__ __forceinline__ bool valid(Ctx &c, int j) { return L::valid(c, j) & R::valid(c, j) & nonzero(R::eval(c, j)); }
  static __device__ __forceinline__ typename Type::T eval(Ctx &c, int j) {
    const auto a = L::eval(c, j);
    const auto b = R::eval(c, j);
    if constexpr (Type::is_float) {
      if constexpr (EXACT_NAN) return f64_result_as_sse2(a / b, a, b);
      else return a / b;
    }
    else {
      const int64_t x = (int64_t)a, y = (int64_t)b;
      const bool ovf = (x == (int64_t)0x8000000000000000ull) & (y == -1);
      c.err = max(c.err, (ovf & valid(c, j)) ? kErrOverflow : 0u);
      return ((y == 0) | ovf) ? 0 : x / y;
    }
  }
};

// GROUP BY aggregate arguments go through the PlanValue interpreter, where Int∘Int for
// + - * % is computed in f64 and cast back (llkv-executor/src/lib.rs:7338-7389); x % 0 is NULL.
template <int OP, class L, class R> struct BinViaF64 {
  using Type = I64;
  static __device__ __forceinline__ bool valid(Ctx &c, int j) {
    if constexpr (OP == B_REM) return L::valid(c, j) & R::valid(c, j) & ((double)R::eval(c, j) != 0.0);
    else return L::valid(c, j) & R::valid(c, j);
  }
  static __device__ __forceinline__ int64_t eval(Ctx &c, int j) {
    const double a = (double)L::eval(c, j), b = (double)R::eval(c, j);
    if constexpr (OP == B_ADD) return f64_as_i64_sat(a + b);
    else if constexpr (OP == B_SUB) return f64_as_i64_sat(a - b);
    else if constexpr (OP == B_MUL) return f64_as_i64_sat(a * b);
    else return f64_as_i64_sat(fmod(a, b));
  }
};
// PlanValue division / modulo with a Float operand: f64 arithmetic, NULL when the divisor is zero.
template <int IS_MOD, class L, class R> struct DivPV {
  using Type = F64;
  static __device__ __forceinline__ bool valid(Ctx &c, int j) { return L::valid(c, j) & R::valid(c, j) & ((double)R::eval(c, j) != 0.0); }
  static __device__ __forceinline__ double eval(Ctx &c, int j) {
    const double a = (double)L::eval(c, j), b = (double)R::eval(c, j);
    if constexpr (IS_MOD) return fmod(a, b);
    else return a / b;
  }
};
// Exact decimal arithmetic of the PlanValue interpreter (llkv-executor/src/lib.rs:7229-7330 over llkv-compute/src/scalar/
// decimal.rs:128-234: add / sub at the larger scale, mul at the sum of the scales, every step checked in i256 and against 38
// digits).  The operands are the 64-bit images of Decimal128 cells, Int64 cells (DecimalValue::from_i64) and literals; the host
// lowering (plan.cpp: expr_planvalue) carries the scales, writes a rescale as a multiplication by the literal 10^k and admits
// the plan only when interval arithmetic over the column statistics keeps EVERY intermediate inside 64 bits — so none of the
// reference's checks can fire and plain wrapping integer arithmetic is exact.
template <int OP, class L, class R> struct DecBin {
  using Type = I64;
  static __device__ __forceinline__ bool valid(Ctx &c, int j) { return L::valid(c, j) & R::valid(c, j); }
  static __device__ __forceinline__ int64_t eval(Ctx &c, int j) {
    const uint64_t a = (uint64_t)(int64_t)L::eval(c, j), b = (uint64_t)(int64_t)R::eval(c, j);
    if constexpr (OP == B_ADD) return (int64_t)(a + b);
    else if constexpr (OP == B_SUB) return (int64_t)(a - b);
    else return (int64_t)(a * b);
  }
};
// Division to the LEFT operand's scale (:7296-7316, decimal.rs:168-234): numerator · 10^(divisor's scale) (P10, the host's
// literal), truncating quotient, then the reference's rounding as written — half = denominator / 2 TRUNCATED, |remainder| ≥
// |half| rounds, and the direction follows the signs of the truncated quotient and the denominator (a quotient of 0 counts as
// positive): 1.00 / 3 = 0.34, −0.01 / 3 = +0.01.  A zero divisor is NULL.
template <class L, class R, class P10> struct DecDiv {
  using Type = I64;
  static __device__ __forceinline__ bool valid(Ctx &c, int j) { return L::valid(c, j) & R::valid(c, j) & ((int64_t)R::eval(c, j) != 0); }
  static __device__ __forceinline__ int64_t eval(Ctx &c, int j) {
    const int64_t num = (int64_t)((uint64_t)(int64_t)L::eval(c, j) * (uint64_t)(int64_t)P10::eval(c, j));
    const int64_t den = (int64_t)R::eval(c, j);
    if (den == 0) return 0;
    const uint64_t an = num < 0 ? 0ull - (uint64_t)num : (uint64_t)num, ad = den < 0 ? 0ull - (uint64_t)den : (uint64_t)den;
    const uint64_t qm = an / ad, rm = an - qm * ad;
    int64_t q = ((num < 0) != (den < 0)) ? -(int64_t)qm : (int64_t)qm;
    if ((rm != 0) & (rm >= ad / 2)) q += ((q >= 0) == (den >= 0)) ? 1 : -1;
    return q;
  }
};
constexpr uint64_t dec_pow10(int k) { uint64_t v = 1; for (int i = 0; i < k; ++i) v *= 10ull; return v; }
// Decimal digits of |E| (digit_count_i256 llkv-types/src/decimal.rs:218-231: 1 for zero), known by the lowering's interval
// arithmetic to lie in [LO, HI]: HI − LO compares against literals.
template <class E, int LO, int HI> struct DecDigits {
  static __device__ __forceinline__ uint64_t digits(Ctx &c, int j) {
    const int64_t v = (int64_t)E::eval(c, j);
    const uint64_t m = v < 0 ? 0ull - (uint64_t)v : (uint64_t)v;
    uint64_t d = LO;
#pragma unroll
    for (int k = LO; k < HI; ++k) d += m >= dec_pow10(k) ? 1u : 0u;
    return d;
  }
};
// The temp column a group's computed decimal argument becomes is typed by the group's FIRST non-NULL value
// (plan_values_to_arrow_array llkv-executor/src/lib.rs:298-330): Decimal128(its digit count, its scale) — the precision the
// finalized cell carries, and a query error when the scale exceeds it.  One MIN lane over (row << 6·n | digits of up to n = 4
// arguments that are NULL in the same rows — V): the smallest key is the first row's.
template <class V, class... Ds> struct FirstDigits {
  static constexpr int N = 1;
  static constexpr int op(int) { return OP_MIN_I64; }
  template <class D0, class... Dr> static __device__ __forceinline__ uint64_t pack(Ctx &c, int j) {
    const uint64_t d = D0::digits(c, j);
    if constexpr (sizeof...(Dr) > 0) return (d << (6 * sizeof...(Dr))) | pack<Dr...>(c, j);
    else return d;
  }
  static __device__ __forceinline__ void contrib(Ctx &c, int j, uint64_t *o) {
    o[0] = V::eval(c, j) ? ((c.row << (6 * sizeof...(Ds))) | pack<Ds...>(c, j)) : 0x7FFFFFFFFFFFFFFFull;
  }
  static constexpr bool first_only(int) { return LLKV_FIRST_ONLY_MODE >= 1; } // the key grows with the row id: a thread's first contribution to a group is its smallest
};
// PlanValue Int / Int (llkv-executor/src/lib.rs:7213-7227): truncating integer division, NULL for a zero divisor.  The one pair
// that turns Float there — i64::MIN / −1 — is excluded by the host (column statistics or a literal divisor), so the group's temp
// column stays Int64 whatever the data.
template <class L, class R> struct DivIntPV {
  using Type = I64;
  static __device__ __forceinline__ bool valid(Ctx &c, int j) { return L::valid(c, j) & R::valid(c, j) & ((int64_t)R::eval(c, j) != 0); }
  static __device__ __forceinline__ int64_t eval(Ctx &c, int j) {
    const int64_t a = (int64_t)L::eval(c, j), b = (int64_t)R::eval(c, j);
    return b == 0 ? 0 : b == -1 ? (int64_t)(0ull - (uint64_t)a) : a / b;
  }
};
// Predicate form of an expression's validity.
template <class E> struct VE {
  static __device__ __forceinline__ bool eval(Ctx &c, int j) { return E::valid(c, j); }
};

// --------------------------------------------------------------------------
// Predicates (llkv-expr/src/typed_predicate.rs:75-146).  Native comparison operators
// give Rust's partial_cmp behaviour: a NaN never matches an ordering or equality test.
// Bound kinds: 0 unbounded, 1 included, 2 excluded.
// --------------------------------------------------------------------------
struct Nil {}; // unbounded side of a Range
struct True {
  static __device__ __forceinline__ bool eval(Ctx &, int) { return true; }
};
struct False {
  static __device__ __forceinline__ bool eval(Ctx &, int) { return false; }
};
template <class E, int LK, class LO, int UK, class HI> struct Range {
  static __device__ __forceinline__ bool eval(Ctx &c, int j) {
    const auto v = E::eval(c, j);
    bool ok = true;
    // `&`, not `&&`: the row body must stay straight-line (no exec-mask branches)
    if constexpr (LK == 1) ok = ok & (v >= LO::eval(c, j));
    if constexpr (LK == 2) ok = ok & (v > LO::eval(c, j));
    if constexpr (UK == 1) ok = ok & (v <= HI::eval(c, j));
    if constexpr (UK == 2) ok = ok & (v < HI::eval(c, j));
    return ok;
  }
};
template <class E, class V> struct Eq {
  static __device__ __forceinline__ bool eval(Ctx &c, int j) { return E::eval(c, j) == V::eval(c, j); }
};
// Any predicate over a dictionary-coded (≤ 256 entries) Utf8 column: the host evaluates it once per dictionary
// string and hands over the 256-bit set of qualifying codes (ordering predicates compare strings as
// Rust's `str::cmp`, llkv-expr/src/typed_predicate.rs:171-185).
template <class E, class M0, class M1, class M2, class M3> struct InMask {
  static __device__ __forceinline__ bool eval(Ctx &c, int j) {
    const uint32_t code = (uint32_t)E::eval(c, j);
    const uint32_t w = code >> 6;
    const uint64_t m = w == 0 ? M0::eval(c, j) : w == 1 ? M1::eval(c, j) : w == 2 ? M2::eval(c, j) : M3::eval(c, j);
    return (m >> (code & 63)) & 1u;
  }
};
template <class E, class... Vs> struct In {
  static __device__ __forceinline__ bool eval(Ctx &c, int j) {
    const auto v = E::eval(c, j);
    return (bool)((int)(v == Vs::eval(c, j)) | ...);
  }
};
// Expr::Compare over two scalar expressions of one common type (llkv-compute/src/kernels.rs:269-297).
// arrow-ord `cmp::{eq,neq,lt,lt_eq,gt,gt_eq}`: integers natively, floats by IEEE totalOrder.
// OP: 1 eq, 2 neq, 3 lt, 4 lteq, 5 gt, 6 gteq.
__device__ __forceinline__ int64_t f64_total_order_key(double v) {
  const int64_t b = __double_as_longlong(v);
  return b ^ (int64_t)((uint64_t)(b >> 63) >> 1);
}
struct AlwaysValid {
  static __device__ __forceinline__ bool eval(Ctx &, int) { return true; }
};
// V: rows where every field of the two sides is present — the compare's domain; elsewhere nothing is
// evaluated (no match, no arithmetic error).
template <int OP, class L, class R, class V = AlwaysValid> struct Cmp {
  template <class T> static __device__ __forceinline__ bool rel(T a, T b) {
    if constexpr (OP == 1) return a == b; else if constexpr (OP == 2) return a != b; else if constexpr (OP == 3) return a < b;
    else if constexpr (OP == 4) return a <= b; else if constexpr (OP == 5) return a > b; else return a >= b;
  }
  static __device__ __forceinline__ bool eval(Ctx &c, int j) {
    // both sides are evaluated over the whole domain, so their arithmetic errors count on rows the rest of
    // the predicate rejects too (evaluate_compare_rows runs before the row sets are intersected)
    const uint32_t outer = c.err;
    c.err = 0;
    const auto a = L::eval(c, j);
    const auto b = R::eval(c, j);
    const bool v = V::eval(c, j);
    c.perr |= v ? c.err : 0u;
    c.err = outer;
    bool m;
    if constexpr (L::Type::is_float) m = rel<int64_t>(f64_total_order_key((double)a), f64_total_order_key((double)b));
    else if constexpr (sizeof(typename L::Type::T) == 8 && !(((typename L::Type::T)-1) < 0)) m = rel<uint64_t>((uint64_t)a, (uint64_t)b); // UInt64 common type
    else if constexpr (sizeof(typename R::Type::T) == 8 && !(((typename R::Type::T)-1) < 0)) m = rel<uint64_t>((uint64_t)a, (uint64_t)b);
    else m = rel<int64_t>((int64_t)a, (int64_t)b);
    return v & m;
  }
};
// A compare whose other side is the NULL literal matches nothing and determines nothing, but the side that is there is still
// evaluated over the rows where its fields are present (evaluate_compare_rows, llkv-scan/src/predicate.rs:562-663): its checked
// arithmetic can fail the scan.
template <class E, class V = AlwaysValid> struct ErrOnly {
  static __device__ __forceinline__ bool eval(Ctx &c, int j) {
    const uint32_t outer = c.err;
    c.err = 0;
    (void)E::eval(c, j);
    c.perr |= V::eval(c, j) ? c.err : 0u;
    c.err = outer;
    return false;
  }
};
// MVCC visibility of a row version (llkv-transaction/src/mvcc.rs:283-333), fused into the scan instead of
// the reference's per-row gather of `_created_by` / `_deleted_by` (helpers.rs:205-244).  UN… = txn ids whose
// status is not Committed; TXN_ID_NONE = u64::MAX has status None; TXN_ID_AUTO_COMMIT = 1 is never "current".
template <class C, class D, class TXN, class SNAP, class... UN> struct Mvcc {
  static __device__ __forceinline__ bool eval(Ctx &c, int j) {
    const uint64_t created = (uint64_t)C::eval(c, j), deleted = (uint64_t)D::eval(c, j);
    const uint64_t txn = (uint64_t)TXN::eval(c, j), snap = (uint64_t)SNAP::eval(c, j);
    const uint64_t none = ~0ull;
    const bool cur = txn != 1ull;
    const bool c_uncommitted = (created == none) | (bool)((int)0 | ... | (int)(created == (uint64_t)UN::eval(c, j)));
    const bool d_uncommitted = (bool)((int)0 | ... | (int)(deleted == (uint64_t)UN::eval(c, j)));
    const bool own = (created == txn) & cur;
    const bool other = !c_uncommitted & (created <= snap) &
                       ((deleted == none) | (!((deleted == txn) & cur) & (d_uncommitted | (deleted > snap))));
    return own ? (deleted != txn) : other;
  }
};
// Semi join against a statistics-bounded key set: bit (key − bm_min) of the launch's bitmap (ScanParams::bm_bits;
// the dimension chains of join → aggregate pipelines, llkv-executor/src/lib.rs:3780-4052 restated as a filter of
// the middle table).
template <class E> struct InKeySet {
  static __device__ __forceinline__ bool eval(Ctx &c, int j) {
    const uint64_t d = (uint64_t)(long long)E::eval(c, j) - (uint64_t)c.p.bm_min; // key < min wraps to a huge value
    if (d > c.p.bm_span) return false;
    return (c.p.bm_bits[d >> 6] >> (d & 63)) & 1ull;
  }
};
// A, and only for the rows that pass it, B (B gathers from a table: skipping it for the rows A rejects saves the
// loads; And<> evaluates every conjunct branch-free)
template <class A, class B> struct AndThen {
  static __device__ __forceinline__ bool eval(Ctx &c, int j) { return A::eval(c, j) && B::eval(c, j); }
};
template <class... Ps> struct And {
  static __device__ __forceinline__ bool eval(Ctx &c, int j) { return (bool)((int)Ps::eval(c, j) & ...); }
};
template <class... Ps> struct Or {
  static __device__ __forceinline__ bool eval(Ctx &c, int j) { return (bool)((int)Ps::eval(c, j) | ...); }
};
// Complement within all rows; the host lowering intersects it with the child's domain when some column of
// the child has NULL cells (llkv-scan/src/predicate.rs:167-186).
template <class P> struct Not {
  static __device__ __forceinline__ bool eval(Ctx &c, int j) { return !P::eval(c, j); }
};
// The field of validity slot S is present in this row (1 B/row mask staged beside the values).
template <int S> struct Valid {
  static __device__ __forceinline__ bool eval(Ctx &c, int j) { return c.get<U8>(S, j) != 0; }
};

// --------------------------------------------------------------------------
// Dense group ids from 1-byte dictionary codes (Utf8 keys staged as codes) or small
// integers.  gid = Σ code_i · stride_i  <  NG.
// --------------------------------------------------------------------------
template <int S> struct KeyCode {
  static __device__ __forceinline__ uint32_t code(Ctx &c, int j) { return c.get<U8>(S, j); }
};
// Integer key with a small statistics-bounded range: code = value − column minimum.
template <int S, class Ty, class Base> struct KeyInt {
  static __device__ __forceinline__ uint32_t code(Ctx &c, int j) { return (uint32_t)((int64_t)c.get<Ty>(S, j) - (int64_t)Base::eval(c, j)); }
};
// Key column with NULL cells: NULL is its own group, coded right after the non-NULL codes.
template <class V, class K, int NULL_CODE> struct KeyOrNull {
  static __device__ __forceinline__ uint32_t code(Ctx &c, int j) { return V::eval(c, j) ? K::code(c, j) : (uint32_t)NULL_CODE; }
};
// FIRST = 1 keeps the row id of each group's first row (first-appearance output order,
// llkv-executor/src/lib.rs:5065-5089); with ORDER BY on the keys it is not needed.
template <int NG_, int FIRST_, class... Ks> struct Keys {
  static constexpr int NG = NG_;
  static constexpr int FIRST = FIRST_;
  static constexpr int NK = sizeof...(Ks);
  template <int I, class K0, class... Kr> static __device__ __forceinline__ uint32_t acc(Ctx &c, int j) {
    uint32_t g = K0::code(c, j) * c.p.key_stride[I];
    if constexpr (sizeof...(Kr) > 0) g += acc<I + 1, Kr...>(c, j);
    return g;
  }
  static __device__ __forceinline__ uint32_t gid(Ctx &c, int j) {
    if constexpr (NK == 0) return 0u;
    else return acc<0, Ks...>(c, j);
  }
};

// --------------------------------------------------------------------------
// Aggregate lane groups (llkv-aggregate/src/lib.rs:759-1477 state machines, restated as
// order-independent lane updates; finalize happens on the host).
// --------------------------------------------------------------------------
__device__ __forceinline__ int64_t f64_order_key(double v) { // total order, -0 canonicalised to +0
  int64_t b = __double_as_longlong(v == 0.0 ? 0.0 : v);
  return b < 0 ? (b ^ 0x7FFFFFFFFFFFFFFFll) : b;
}

template <class E> struct SumF64 { // SumFloat64 :870-888, AvgFloat64 :1177-1199, Total* :968-1034
  static constexpr int N = 1;
  static constexpr int op(int) { return OP_ADD_F64; }
  static __device__ __forceinline__ void contrib(Ctx &c, int j, uint64_t *o) { o[0] = (uint64_t)__double_as_longlong((double)E::eval(c, j)); }
};
// The same sum, EXACT and therefore order-free: every value is cut into pieces on fixed grids — q1 = v rounded to a
// multiple of u1, q2 = (v − q1) rounded to a multiple of u2, … — chosen by the host (plan.cpp: exact_sum_constants)
// from a bound B ≥ |v| (column statistics through the expression) and the table's row count N so that any partial
// sum of the pieces of one level is a multiple of its grid below 2^53 grid units: every f64 addition of a lane is
// exact, so threads, waves, workgroups and ranks may add in ANY order and still produce the same bits.  The last grid
// resolves the smallest non-zero |v| the statistics allow to 2^-30 of itself, so what a row drops is < 2^-31 of its
// own magnitude.  C(j) = 1.5·2^52·u(j): (x + C) − C rounds x to the grid of C's ulp (nearest-even, branch-free).
// ±∞ and NaN travel in the first lane.
template <class E, class... Cs> struct SumF64X {
  static constexpr int N = sizeof...(Cs);
  static constexpr int op(int) { return OP_ADD_F64; }
  template <class C0, class... Cr> static __device__ __forceinline__ void cut(Ctx &c, int j, double r, uint64_t *o) {
    const double cj = C0::eval(c, j);
    const double q = __dsub_rn(__dadd_rn(r, cj), cj);
    o[0] = (uint64_t)__double_as_longlong(q);
    if constexpr (sizeof...(Cr) > 0) cut<Cr...>(c, j, (__builtin_fabs(r) == __builtin_inf()) ? 0.0 : __dsub_rn(r, q), o + 1);
  }
  static __device__ __forceinline__ void contrib(Ctx &c, int j, uint64_t *o) { cut<Cs...>(c, j, (double)E::eval(c, j), o); }
};
// The same idea in ONE lane when the argument's range allows (shared-image plans, where every DS atomic to a scattered
// address costs tens of cycles): the row's value as an integer count of grid steps, q = rint(x · 2^−e) with the grid
// 2^e at 2^-30 of the smallest non-zero |x| — an int64 sum per workgroup image (|q| · rows of a workgroup < 2^62, the
// lowering checks), split into low 32 bits / high part when the images are folded so that no exchange lane overflows.
// S = LitF<2^−e>.  The argument's columns hold no NaN / ±∞ (staging statistic).
template <class E, class S> struct SumF64Q {
  static constexpr int N = 1;
  static constexpr int op(int) { return OP_ADD_I64; }
  static __device__ __forceinline__ void contrib(Ctx &c, int j, uint64_t *o) { o[0] = (uint64_t)__double2ll_rn((double)E::eval(c, j) * S::eval(c, j)); }
};
// The exact form for the register / per-thread-column plans (planning option llkv_hip_set_exact_f64_sums): the grid is the
// ulp of the smallest non-zero |x| the statistics allow, so x · 2^−e IS an integer (below 2^62, the lowering checks) and
// the two lanes — its low 32 bits, the rest — add without overflow below 2^31 rows: the finalize step rounds the exact
// sum once.
template <class E, class S> struct SumF64Q2 {
  static constexpr int N = 2;
  static constexpr int op(int) { return OP_ADD_I64; }
  static __device__ __forceinline__ void contrib(Ctx &c, int j, uint64_t *o) {
    const long long q = __double2ll_rn((double)E::eval(c, j) * S::eval(c, j));
    o[0] = (uint64_t)q & 0xFFFFFFFFull;
    o[1] = (uint64_t)(q >> 32);
  }
};
// DISTINCT forms inside GROUP BY (sort-based route: the argument column is the least significant sort key, so equal
// values of a group are neighbours and group_reduce_body marks the first of each run): every group runs the reference's
// distinct accumulators over its own rows (llkv-executor/src/lib.rs:5222-5247, llkv-aggregate/src/lib.rs:95-249).
struct DistinctCount {
  static constexpr int N = 1;
  static constexpr int op(int) { return OP_ADD_I64; }
  static __device__ __forceinline__ void contrib(Ctx &c, int, uint64_t *o) { o[0] = c.dhead; }
};
struct DistinctSumI64 { // (statistics exclude an i64 overflow of the sum)
  static constexpr int N = 1;
  static constexpr int op(int) { return OP_ADD_I64; }
  static __device__ __forceinline__ void contrib(Ctx &c, int, uint64_t *o) { o[0] = c.dhead ? c.dval : 0ull; }
};
struct DistinctTotalI64 { // TOTAL(DISTINCT int): an f64 sum of the values as f64
  static constexpr int N = 1;
  static constexpr int op(int) { return OP_ADD_F64; }
  static __device__ __forceinline__ void contrib(Ctx &c, int, uint64_t *o) { o[0] = (uint64_t)__double_as_longlong(c.dhead ? (double)(int64_t)c.dval : 0.0); }
};
struct DistinctSumF64 {
  static constexpr int N = 1;
  static constexpr int op(int) { return OP_ADD_F64; }
  static __device__ __forceinline__ void contrib(Ctx &c, int, uint64_t *o) { o[0] = c.dhead ? c.dval : 0ull; } // +0.0 for the others
};
template <class E> struct SumI64 { // SumInt64 :801-830, AvgInt64 :1114-1144 — exact 96-bit split sum + max|v|
  static constexpr int N = 3;
  static constexpr int op(int k) { return k == 2 ? OP_MAX_U64 : OP_ADD_I64; }
  static __device__ __forceinline__ void contrib(Ctx &c, int j, uint64_t *o) {
    const int64_t v = (int64_t)E::eval(c, j);
    o[0] = (uint64_t)(uint32_t)v;       // low 32 bits, unsigned
    o[1] = (uint64_t)(v >> 32);         // high part, signed
    o[2] = v < 0 ? (uint64_t)(-(v + 1)) + 1u : (uint64_t)v;
  }
};
// Same sum when the column statistics gathered at staging prove rows·max|v| < 2^63
// (no prefix of the reference's checked_add chain can overflow): one wrapping lane.
// SUM / TOTAL / AVG over a Decimal128 column with values beyond 64 bits (llkv-aggregate/src/lib.rs:925-943: i128
// checked_add row by row): the column is two 8 B/row buffers (slot SLO: low halves, slot SHI: high halves), the lanes
// are the sums of the four 32-bit limbs — the top one sign-extended — so every lane stays inside 64 bits for < 2^31
// rows and the host rebuilds Σ v mod 2^128 (the lowering admits the plan only when no prefix can leave i128).
template <int SLO, int SHI> struct SumDecWide {
  static constexpr int N = 4;
  static constexpr int op(int) { return OP_ADD_I64; }
  static __device__ __forceinline__ void contrib(Ctx &c, int j, uint64_t *o) {
    const uint64_t lo = c.get<U64>(SLO, j);
    const int64_t hi = c.get<I64>(SHI, j);
    o[0] = lo & 0xFFFFFFFFull;
    o[1] = lo >> 32;
    o[2] = (uint64_t)hi & 0xFFFFFFFFull;
    o[3] = (uint64_t)(hi >> 32); // arithmetic shift: the signed top limb
  }
};
// MIN / MAX over such a column whose values span less than 2^64 (llkv-aggregate/src/lib.rs:1332-1352,1400-1420): v − min(column)
// — or max(column) − v — fits 64 bits and is the wrapping difference of the low halves; the host adds the base back in i128.
template <int SLO, class Base, int NEG> struct MaxWideDelta {
  static constexpr int N = 1;
  static constexpr int op(int) { return OP_MAX_U64; }
  static __device__ __forceinline__ void contrib(Ctx &c, int j, uint64_t *o) {
    const uint64_t lo = c.get<U64>(SLO, j), base = Base::eval(c, j);
    o[0] = NEG ? base - lo : lo - base;
  }
};
template <class E> struct SumI64Fast {
  static constexpr int N = 1;
  static constexpr int op(int) { return OP_ADD_I64; }
  static __device__ __forceinline__ void contrib(Ctx &c, int j, uint64_t *o) { o[0] = (uint64_t)(int64_t)E::eval(c, j); }
};
template <class E> struct MinI64 { // :1285-1308
  static constexpr int N = 1;
  static constexpr int op(int) { return OP_MIN_I64; }
  static __device__ __forceinline__ void contrib(Ctx &c, int j, uint64_t *o) { o[0] = (uint64_t)(int64_t)E::eval(c, j); }
};
template <class E> struct MaxI64 { // :1354-1376
  static constexpr int N = 1;
  static constexpr int op(int) { return OP_MAX_I64; }
  static __device__ __forceinline__ void contrib(Ctx &c, int j, uint64_t *o) { o[0] = (uint64_t)(int64_t)E::eval(c, j); }
};
// MinFloat64 :1309-1331 / MaxFloat64 :1377-1399 — sequential partial_cmp fold: a NaN never
// replaces, a leading NaN sticks, ±0 ties keep the earlier row.  Lanes: best key over
// non-NaN values, first zero (row<<1|sign), first selected row (row<<1|isnan).
template <class E, bool IS_MAX> struct ExtF64 {
  static constexpr int N = 3;
  static constexpr int op(int k) { return k == 0 ? (IS_MAX ? OP_MAX_I64 : OP_MIN_I64) : OP_MIN_I64; }
  static __device__ __forceinline__ void contrib(Ctx &c, int j, uint64_t *o) {
    const double v = (double)E::eval(c, j);
    const bool nan = v != v;
    o[0] = nan ? lane_identity_s<IS_MAX ? OP_MAX_I64 : OP_MIN_I64>() : (uint64_t)f64_order_key(v);
    o[1] = (v == 0.0) ? ((c.row << 1) | (uint64_t)((uint64_t)__double_as_longlong(v) >> 63)) : 0x7FFFFFFFFFFFFFFFull;
    o[2] = (c.row << 1) | (nan ? 1u : 0u);
  }
  static constexpr bool first_only(int k) { return LLKV_FIRST_ONLY_MODE >= 2 && k >= 1; } // lanes 1 and 2 are minima of keys that grow with the row id
};
// The same over a column that holds neither NaN nor −0.0 (staging statistics): nothing sticks, no tie depends on the row order —
// the order key alone.
template <class E, bool IS_MAX> struct ExtF64P {
  static constexpr int N = 1;
  static constexpr int op(int) { return IS_MAX ? OP_MAX_I64 : OP_MIN_I64; }
  static __device__ __forceinline__ void contrib(Ctx &c, int j, uint64_t *o) { o[0] = (uint64_t)f64_order_key((double)E::eval(c, j)); }
};
template <class E> using MinF64P = ExtF64P<E, false>;
template <class E> using MaxF64P = ExtF64P<E, true>;
template <class E> using MinF64 = ExtF64<E, false>;
template <class E> using MaxF64 = ExtF64<E, true>;

// COUNT(x) / COUNT_NULLS(x) over an argument with NULL cells: the rows where it is present (:769-786,1422-1445).
template <class V> struct CountIf {
  static constexpr int N = 1;
  static constexpr int op(int) { return OP_ADD_I64; }
  static __device__ __forceinline__ void contrib(Ctx &c, int j, uint64_t *o) { o[0] = V::eval(c, j) ? 1u : 0u; }
};
// The same over a COMPUTED argument: the reference materialises the projection for every selected row before any
// accumulator sees it (llkv-executor/src/lib.rs:470-501), so a checked-arithmetic error in it (overflow, % by zero)
// fails the query even though COUNT only looks at the validity — the expression is evaluated for that effect.
template <class V, class E> struct CountIfE {
  static constexpr int N = 1;
  static constexpr int op(int) { return OP_ADD_I64; }
  static __device__ __forceinline__ void contrib(Ctx &c, int j, uint64_t *o) {
    (void)E::eval(c, j);
    o[0] = V::eval(c, j) ? 1u : 0u;
  }
};
// Any accumulator over an argument with NULL cells: a NULL row contributes every lane's identity (accumulators
// skip NULLs), and one more lane counts the non-NULL
// rows — the AVG denominator and the "no rows → NULL" test of finalize.
// MIN lanes whose key is (row id << n | …): within one thread the keys of a group only grow, so the thread's FIRST contribution
// to a group that is not the lane's identity is its smallest — the per-thread-column kernel skips the DS operation of every later
// one (agg_first_only; same minimum, fewer LDS instructions per row).
template <class A, class = void> struct AggFirstOnly { static constexpr bool at(int) { return false; } };
template <class A> struct AggFirstOnly<A, decltype((void)A::first_only(0))> { static constexpr bool at(int k) { return A::first_only(k); } };

template <class V, class A> struct IfValid {
  static constexpr int N = A::N + 1;
  static constexpr int op(int k) { return k < A::N ? A::op(k) : OP_ADD_I64; }
  static constexpr bool first_only(int k) { return k < A::N && AggFirstOnly<A>::at(k); }
  static __device__ __forceinline__ void contrib(Ctx &c, int j, uint64_t *o) {
    A::contrib(c, j, o);
    const bool v = V::eval(c, j);
#pragma unroll
    for (int k = 0; k < A::N; ++k) o[k] = v ? o[k] : lane_identity(A::op(k));
    o[A::N] = v ? 1u : 0u;
  }
};

template <class... As> struct Aggs {
  static constexpr int N = (0 + ... + As::N);
};

// Plan = Cols × Pred × Keys × Aggs × unroll × accumulator placement.  Lane layout per group:
//   [0] rows (ADD_I64)   [1] first row id (MIN_I64, only with Keys<.., FIRST=1, ..>)   [..] aggregate lanes
// ACC = 0: accumulators in registers, masked per group (ungrouped plans, NG = 1);
// ACC = 1: accumulators in per-thread private LDS slots updated with DS atomics
//          (ds_add_f64 / ds_add_u64 / ds_min_i64 / ...), indexed by the row's own group id.
// ACC = 2: ONE accumulator image per workgroup in LDS, [lane][group], shared by its 1024 threads and updated with
//          the same DS atomics (image_scan_body): hundreds to thousands of groups.  Every lane op must be order-free,
//          so f64 sums are the exact two-level SumF64X.
template <class CL, class PR, class KS, class AG, int U_ = 2, int ACC_ = 0, int PASSES_ = 1, int EARLY_ = -1, int CELL32_ = 0> struct Plan {
  // shared-image plans whose every lane is a count or an integer sum that the statistics keep below 2^31 per workgroup image
  // (and a first-row lane over fewer than 2^32 rows): 4-byte cells — twice the groups per LDS-sized slice, 32-bit DS atomics
  static constexpr bool CELL32 = CELL32_ != 0;
  // register-state plans: the columns [0, EARLY) feed the predicate (and keys) and are streamed for every row, the ones
  // behind them feed aggregate arguments alone and are read for the row pairs that hold a passing row (EARLY_ < 0: all
  // columns up front)
  static constexpr int EARLY = EARLY_ < 0 ? CL::N : EARLY_;
  static constexpr int PASSES = PASSES_; // shared-image plans: the groups are cut into PASSES slices, one scan each
  using ColList = CL;
  using Pred = PR;
  using KeyT = KS;
  using AggT = AG;
  static constexpr int U = U_;
  static constexpr int ACC = ACC_;
  static constexpr int NG = KS::NG;
  static constexpr bool grouped = KS::NK > 0;
  static constexpr bool first = KS::FIRST != 0;
  static constexpr int BASE = first ? 2 : 1;
  static constexpr int K = BASE + AG::N;
  static constexpr int LANES = NG * K + 1; // + error lane (MAX_U64)
};

template <class AG> struct AggOps;
template <class... As> struct AggOps<Aggs<As...>> {
  template <class A0, class... Ar> static constexpr int at(int k) {
    if (k < A0::N) return A0::op(k);
    if constexpr (sizeof...(Ar) > 0) return at<Ar...>(k - A0::N);
    else return OP_ADD_I64;
  }
  static constexpr int op(int k) {
    if constexpr (sizeof...(As) == 0) return OP_ADD_I64;
    else return at<As...>(k);
  }
  template <class A0, class... Ar> static __device__ __forceinline__ void contrib_all(Ctx &c, int j, uint64_t *o) {
    A0::contrib(c, j, o);
    if constexpr (sizeof...(Ar) > 0) contrib_all<Ar...>(c, j, o + A0::N);
  }
  static __device__ __forceinline__ void contrib(Ctx &c, int j, uint64_t *o) {
    if constexpr (sizeof...(As) > 0) contrib_all<As...>(c, j, o);
  }
  template <class A0, class... Ar> static constexpr bool first_only_at(int k) {
    if (k < A0::N) return AggFirstOnly<A0>::at(k);
    if constexpr (sizeof...(Ar) > 0) return first_only_at<Ar...>(k - A0::N);
    else return false;
  }
  static constexpr bool first_only(int k) {
    if constexpr (sizeof...(As) == 0) return false;
    else return first_only_at<As...>(k);
  }
};

// Lane k of a group's block (0 ≤ k < K): a MIN over keys that grow with the row id (the first-row lane, FirstDigits, the
// first-row / first-zero lanes of the f64 MIN / MAX)?
template <class P> constexpr bool plan_lane_first_only(int k) {
  if (k == 0) return false;
  if (P::first && k == 1) return LLKV_FIRST_ONLY_MODE >= 2;
  return AggOps<typename P::AggT>::first_only(k - P::BASE);
}
template <class P> constexpr int plan_first_only_index(int k) { // how many such lanes sit before lane k
  int n = 0;
  for (int i = 0; i < k; ++i) n += plan_lane_first_only<P>(i) ? 1 : 0;
  return n;
}

template <class P> constexpr int plan_lane_op(int lane) {
  if (lane == P::NG * P::K) return OP_MAX_U64; // error lane
  const int k = lane % P::K;
  if (k == 0) return OP_ADD_I64;
  if (P::first && k == 1) return OP_MIN_I64;
  return AggOps<typename P::AggT>::op(k - P::BASE);
}

template <class P> struct LaneOpTable {
  int v[P::LANES];
  constexpr LaneOpTable() : v{} {
    for (int i = 0; i < P::LANES; ++i) v[i] = plan_lane_op<P>(i);
  }
};

// --------------------------------------------------------------------------
// Octant fold: tile partials → exchange image [kOctants][lanes], one wave per (octant, lane), tiles
// combined in tile order (8 loads in flight, same association as a sequential loop).  Rows of octants
// this rank does not own are written as zero so that an integer-sum all-reduce of the image
// concatenates the ranks' states bit-exactly; an owned octant without tiles yields the lane identity.
// --------------------------------------------------------------------------
// `ppt`: partials per tile (1: register plans, one per workgroup; kLdsParts: LDS-accumulator plans, one per wave).
__device__ __forceinline__ void fold_one_lane(const uint64_t *tile_partials, uint64_t *exchange, const uint32_t *octant_tile_begin,
                                              uint32_t owned_mask, uint32_t n_tiles, uint32_t lanes, uint32_t o, uint32_t lane, int op, uint32_t ppt) {
  const uint32_t l = threadIdx.x & 63;
  if (!((owned_mask >> o) & 1u)) {
    if (l == 0) exchange[(uint64_t)o * lanes + lane] = 0;
    return;
  }
  const uint32_t t0 = octant_tile_begin[o] * ppt, t1 = octant_tile_begin[o + 1] * ppt;
  const uint64_t *src = tile_partials + (uint64_t)lane * n_tiles * ppt;
  uint64_t v = lane_identity(op);
  uint32_t t = t0 + l;
  for (; t + 448 < t1; t += 512) {
    uint64_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = src[t + 64 * i];
#pragma unroll
    for (int i = 0; i < 8; ++i) v = lane_combine(op, v, a[i]);
  }
  for (; t < t1; t += 64) v = lane_combine(op, v, src[t]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)v, off, 64);
    const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), off, 64);
    const uint64_t ov = ((uint64_t)hi << 32) | lo;
    v = (l & off) ? lane_combine(op, ov, v) : lane_combine(op, v, ov);
  }
  if (l == 0) exchange[(uint64_t)o * lanes + lane] = v;
}

// Piggy-back form: workgroups [0, kOctants · ceil(LANES/4)) of a scan launch fold the PREVIOUS execution.
template <class P> constexpr uint32_t piggyback_blocks() { return (uint32_t)kOctants * (uint32_t)((P::LANES + kBlock / 64 - 1) / (kBlock / 64)); }

template <class P> __device__ __forceinline__ void piggyback_fold(const ScanParams &p) {
  if (p.prev_partials == nullptr || blockIdx.x >= piggyback_blocks<P>()) return;
  constexpr LaneOpTable<P> ops{};
  const uint32_t o = blockIdx.x % kOctants;
  const uint32_t lane = (blockIdx.x / kOctants) * (kBlock / 64) + (threadIdx.x >> 6);
  if (lane < (uint32_t)P::LANES)
    fold_one_lane(p.prev_partials, p.prev_exchange, p.octant_tile_begin, p.owned_mask, p.n_tiles, (uint32_t)P::LANES, o, lane, ops.v[lane],
                  P::ACC == 1 ? (uint32_t)(kBlock / 64) : 1u);
}

__device__ __forceinline__ void publish_partial(const ScanParams &p, int lane, uint64_t v, uint32_t part, uint32_t n_parts) {
  p.tile_partials[(uint64_t)lane * n_parts + part] = v;
}

// --------------------------------------------------------------------------
// The kernel
// --------------------------------------------------------------------------
constexpr int kRedBatch = 16;                 // lanes transposed through LDS per round
constexpr int kRedRow = kBlock + kBlock / 16; // 16-element segments padded to 17

template <class P> __device__ __forceinline__ void fused_scan_body_reg(const ScanParams &p) {
  constexpr int NG = P::NG, K = P::K, U = P::U, LANES = P::LANES;
  constexpr LaneOpTable<P> ops{};
  __shared__ uint64_t red[LANES < kRedBatch ? LANES : kRedBatch][kRedRow];
  const uint32_t tid = threadIdx.x;
  piggyback_fold<P>(p);
  // One workgroup per tile, or (scan_grid != 0: tables of a few thousand tiles, where the dispatch of one short workgroup
  // per tile is a visible part of the kernel) workgroup b of g streams the tiles [b·n/g, (b+1)·n/g).  Either way every tile
  // is reduced on its own and publishes its own partial: the association — and so the bits — do not depend on the grid.
  const uint32_t g = p.scan_grid;
  if (g && blockIdx.x >= g) return; // (a launch wider than the grid the host announced must not walk off the tile list)
  const uint32_t t_first = g ? (uint32_t)((uint64_t)blockIdx.x * p.n_tiles / g) : blockIdx.x;
  const uint32_t t_last = g ? (uint32_t)((uint64_t)(blockIdx.x + 1) * p.n_tiles / g) : (blockIdx.x < p.n_tiles ? blockIdx.x + 1 : blockIdx.x);
  for (uint32_t tile = t_first; tile < t_last; ++tile) {
  uint64_t acc[NG][K];
#pragma unroll
  for (int gg = 0; gg < NG; ++gg)
#pragma unroll
    for (int k = 0; k < K; ++k) acc[gg][k] = lane_identity(ops.v[gg * K + k]);
  uint32_t err = 0;

  const TileDesc td = p.tiles[tile];
  const uint32_t nsteps = (td.rows + kStepRows - 1) / kStepRows;

  if constexpr (P::EARLY < P::ColList::N) {
    // Late materialisation.  Per group of U steps: the predicate over the early columns → the argument columns of the row
    // pairs that hold a passing row → the early columns of the NEXT group (requested behind the late loads, so waiting for
    // the late ones does not wait for them: the counter retires loads in order) → the accumulation.  A lane whose rows both
    // fail requests nothing; the same bits as the eager form (same rows, same order of combination).
    Loaded ld[U], nx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int c = P::EARLY; c < P::ColList::N; ++c) ld[u].w[c][0] = ld[u].w[c][1] = ld[u].w[c][2] = ld[u].w[c][3] = 0u;
      load_range<typename P::ColList, 0, P::EARLY>(p, td.dev_row + (uint64_t)u * kStepRows + (uint64_t)tid * kRowsPerThread, ld[u]);
    }
    for (uint32_t s = 0; s < nsteps; s += U) {
      bool pass[U][kRowsPerThread];
      uint32_t gidv[U][kRowsPerThread];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t row0 = (s + u) * kStepRows + tid * kRowsPerThread;
#pragma unroll
        for (int j = 0; j < kRowsPerThread; ++j) {
          Ctx c{p, ld[u], 0u, td.logical_row + row0 + j};
          const bool in_tile = (row0 + j) < td.rows;
          pass[u][j] = in_tile & P::Pred::eval(c, j);
          gidv[u][j] = P::KeyT::gid(c, j);
          err |= in_tile ? c.perr : 0u;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        bool any = false;
#pragma unroll
        for (int j = 0; j < kRowsPerThread; ++j) any |= pass[u][j];
        if (any) load_range<typename P::ColList, P::EARLY, P::ColList::N>(p, td.dev_row + (uint64_t)(s + u) * kStepRows + (uint64_t)tid * kRowsPerThread, ld[u]);
      }
      const bool more = s + U < nsteps;
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) load_range<typename P::ColList, 0, P::EARLY>(p, td.dev_row + (uint64_t)(s + U + u) * kStepRows + (uint64_t)tid * kRowsPerThread, nx[u]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t row0 = (s + u) * kStepRows + tid * kRowsPerThread;
#pragma unroll
        for (int j = 0; j < kRowsPerThread; ++j) {
          Ctx c{p, ld[u], 0u, td.logical_row + row0 + j};
          uint64_t contrib[K];
          contrib[0] = 1;
          if constexpr (P::first) contrib[1] = c.row;
          AggOps<typename P::AggT>::contrib(c, j, contrib + P::BASE);
          err |= pass[u][j] ? c.err : 0u;
#pragma unroll
          for (int gg = 0; gg < NG; ++gg) {
            const bool sel = pass[u][j] & (NG == 1 || gidv[u][j] == (uint32_t)gg);
#pragma unroll
            for (int k = 0; k < K; ++k) {
              const int op = ops.v[gg * K + k];
              const uint64_t x = sel ? contrib[k] : lane_identity(op);
              acc[gg][k] = lane_combine(op, acc[gg][k], x);
            }
          }
        }
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int c = 0; c < P::EARLY; ++c)
#pragma unroll
            for (int wd = 0; wd < 4; ++wd) ld[u].w[c][wd] = nx[u].w[c][wd];
      }
    }
  } else
  for (uint32_t s = 0; s < nsteps; s += U) {
    Loaded ld[U];
    // issue every load of the unrolled group before the first use (column buffers carry
    // slack past the last tile, so the tail steps may read — and discard — past td.rows)
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t row0 = (uint64_t)(s + u) * kStepRows + (uint64_t)tid * kRowsPerThread;
      load_all<typename P::ColList>(p, td.dev_row + row0, ld[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t row0 = (s + u) * kStepRows + tid * kRowsPerThread;
#pragma unroll
      for (int j = 0; j < kRowsPerThread; ++j) {
        Ctx c{p, ld[u], 0u, td.logical_row + row0 + j};
        const bool in_tile = (row0 + j) < td.rows;
        const bool pass = in_tile & P::Pred::eval(c, j);
        const uint32_t gid = P::KeyT::gid(c, j);
        uint64_t contrib[K];
        contrib[0] = 1;
        if constexpr (P::first) contrib[1] = c.row;
        AggOps<typename P::AggT>::contrib(c, j, contrib + P::BASE);
        err |= (pass ? c.err : 0u) | (in_tile ? c.perr : 0u);
#pragma unroll
        for (int gg = 0; gg < NG; ++gg) {
          const bool sel = pass & (NG == 1 || gid == (uint32_t)gg);
#pragma unroll
          for (int k = 0; k < K; ++k) {
            const int op = ops.v[gg * K + k];
            const uint64_t x = sel ? contrib[k] : lane_identity(op);
            acc[gg][k] = lane_combine(op, acc[gg][k], x);
          }
        }
      }
    }
  }

  // ---- block reduction: LDS transpose in rounds of kRedBatch lanes, fixed order ----
  const uint32_t slot = tid + (tid >> 4); // padded position of this thread's element
  const uint32_t ri = tid >> 4;           // lane of the round this thread reduces
  const uint32_t rq = tid & 15;           // 16-element segment it reduces
#pragma unroll
  for (int base = 0; base < LANES; base += kRedBatch) {
#pragma unroll
    for (int i = 0; i < kRedBatch; ++i) {
      const int lane = base + i;
      if (lane < LANES) {
        uint64_t v;
        if (lane == NG * K) v = err;
        else v = acc[lane / K][lane % K];
        red[i][slot] = v;
      }
    }
    __syncthreads();
    const int lane = base + (int)ri;
    if (lane < LANES) {
      const int op = ops.v[lane];
      const uint32_t seg = rq * 17;
      uint64_t v = red[ri][seg];
#pragma unroll
      for (int e = 1; e < 16; ++e) v = lane_combine(op, v, red[ri][seg + e]);
      // butterfly over the 16 segment owners (lanes of one 16-lane row of the wave)
#pragma unroll
      for (int off = 8; off >= 1; off >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, off, 16);
        const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), off, 16);
        const uint64_t o = ((uint64_t)hi << 32) | lo;
        // fixed operand order (lower segment first) keeps f64 adds reproducible
        v = (rq & off) ? lane_combine(op, o, v) : lane_combine(op, v, o);
      }
      if (rq == 0) publish_partial(p, lane, v, tile, p.n_tiles);
    }
    __syncthreads();
  }
  } // tiles of this workgroup
}


// ---- grouped plans: accumulators in per-thread private LDS slots -------------------
// acc[slot][tid], slot = gid * K + k.  Every thread only ever touches its own column, so the DS
// read-modify-write instructions are uncontended, execute in program order and the result is
// deterministic; a row updates K slots of ITS group — no per-group masking, no accumulator VGPRs.
// (Measured against a plain ds_read_b64 + VALU op + ds_write_b64 — the slot is private, no atomicity is needed: Q1 SF10
// 0.377 → 0.391 ms, the 12-lane state 0.333 → 0.523 ms; the compiler has to order every row's reads behind the previous
// row's writes, the atomics carry no such dependency.  profiles/r03/lds_rmw_experiment.txt)
template <int OP> __device__ __forceinline__ void lds_accumulate(uint64_t *slot, uint64_t x) {
  if constexpr (OP == OP_ADD_F64)
    (void)__hip_atomic_fetch_add(reinterpret_cast<double *>(slot), __longlong_as_double((long long)x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else if constexpr (OP == OP_ADD_I64)
    (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(slot), (unsigned long long)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else if constexpr (OP == OP_MIN_I64)
    (void)__hip_atomic_fetch_min(reinterpret_cast<long long *>(slot), (long long)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else if constexpr (OP == OP_MAX_I64)
    (void)__hip_atomic_fetch_max(reinterpret_cast<long long *>(slot), (long long)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else
    (void)__hip_atomic_fetch_max(reinterpret_cast<unsigned long long *>(slot), (unsigned long long)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// `seen[i]`: bit g set = this thread has already given first-only lane i of group g a value in this tile
template <class P, int K0 = 0> __device__ __forceinline__ void lds_accumulate_row(uint64_t *group_base, const uint64_t *contrib, uint32_t gid, uint64_t *seen) {
  if constexpr (K0 < P::K) {
    if constexpr (plan_lane_first_only<P>(K0)) {
      constexpr int i = plan_first_only_index<P>(K0);
      const uint64_t bit = 1ull << gid;
      if (!(seen[i] & bit) && contrib[K0] != lane_identity_s<plan_lane_op<P>(K0)>()) {
        lds_accumulate<plan_lane_op<P>(K0)>(group_base + K0 * kBlock, contrib[K0]);
        seen[i] |= bit;
      }
    } else {
      lds_accumulate<plan_lane_op<P>(K0)>(group_base + K0 * kBlock, contrib[K0]);
    }
    lds_accumulate_row<P, K0 + 1>(group_base, contrib, gid, seen);
  }
}

// Orders the LDS traffic of ONE wave: the waves of a workgroup never touch each other's columns of the accumulator
// image, and the DS unit executes a wave's instructions in order, so all that is needed between the phases of a wave
// (its lanes' atomics → cross-lane reads → re-initialisation) is that the COMPILER keeps them in program order.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A tile descriptor through the scalar cache: the tile list is never written while a scan runs, but the compiler
// cannot know that (the kernel stores partials to global memory) and would use a vector load — whose vmcnt then
// orders it with the column loads.  Reading through the constant address space selects s_load_dwordx4 + x2.
__device__ __forceinline__ TileDesc load_tile_desc(const TileDesc *tiles, uint32_t i) {
  typedef const __attribute__((address_space(4))) uint64_t *ConstU64;
  ConstU64 q = (ConstU64)(unsigned long long)(tiles + i);
  TileDesc d;
  d.dev_row = q[0];
  d.logical_row = q[1];
  const uint64_t w = q[2];
  d.rows = (uint32_t)w;
  d.octant = (uint32_t)(w >> 32);
  return d;
}

constexpr int kLdsParts = kBlock / 64; // partials an LDS-accumulator plan publishes per tile: one per wave

// The positions inside a group's lane block whose combine op is OP (the same for every group): the wave reduction
// runs class by class, so every instruction of a round combines with ONE op (a round over mixed lanes made the
// compiler branch five ways per element).  Lane of class position idx = (idx / n)·K + v[idx % n], computed with
// selects: a table in memory would be a VMEM load, and its s_waitcnt vmcnt(0) would wait for the column loads of the
// next tile that are in flight during the reduction.
template <class P, int OP> struct OpSlots {
  int n;
  int v[P::K];
  constexpr OpSlots() : n(0), v{} {
    for (int k = 0; k < P::K; ++k)
      if (plan_lane_op<P>(k) == OP) v[n++] = k;
  }
};

// One op class of the wave reduction: lane `idx` of a round is summed by four threads (one 16-column segment of the
// wave's 64 columns each, starting at column li of the segment — the 16 lanes of a round then read 16 different
// banks), then a 4-lane butterfly, lower segment first.  Fixed order: a function of (lane, wave) only.
template <class P, int OP>
__device__ __forceinline__ void wave_reduce_class(const ScanParams &p, const uint64_t (*acc)[kBlock], uint32_t wave, uint32_t li, uint32_t sq, uint32_t part, uint32_t n_parts) {
  constexpr OpSlots<P, OP> cls{};
  if constexpr (cls.n > 0) {
    constexpr int total = P::NG * cls.n;
#pragma unroll 1
    for (int base = 0; base < total; base += 16) {
      const int idx = base + (int)li;
      if (idx < total) {
        const int g = idx / cls.n, j = idx % cls.n;
        int k = cls.v[0];
#pragma unroll
        for (int t = 1; t < cls.n; ++t) k = j == t ? cls.v[t] : k;
        const int lane = g * P::K + k;
        const uint64_t *seg = &acc[lane][wave * 64 + sq * 16];
        uint64_t x[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e] = seg[(li + e) & 15];
        uint64_t v = x[0];
#pragma unroll
        for (int e = 1; e < 16; ++e) v = lane_combine_s<OP>(v, x[e]);
#pragma unroll
        for (int off = 2; off >= 1; off >>= 1) {
          const uint32_t lo = __shfl_xor((uint32_t)v, off, 4);
          const uint32_t hi = __shfl_xor((uint32_t)(v >> 32), off, 4);
          const uint64_t o = ((uint64_t)hi << 32) | lo;
          v = (sq & off) ? lane_combine_s<OP>(o, v) : lane_combine_s<OP>(v, o);
        }
        if (sq == 0) publish_partial(p, lane, v, part, n_parts);
      }
    }
  }
}

// Reduction of the wave's 64 columns of the image into the partial of (tile, wave), one op class after the other.
template <class P>
__device__ __forceinline__ void wave_reduce_image(const ScanParams &p, const uint64_t (*acc)[kBlock], uint32_t wave, uint32_t wl, uint32_t part, uint32_t n_parts, uint32_t err) {
  const uint32_t li = wl >> 2, sq = wl & 3;
  wave_lds_sync();
  wave_reduce_class<P, OP_ADD_F64>(p, acc, wave, li, sq, part, n_parts);
  wave_reduce_class<P, OP_ADD_I64>(p, acc, wave, li, sq, part, n_parts);
  wave_reduce_class<P, OP_MIN_I64>(p, acc, wave, li, sq, part, n_parts);
  wave_reduce_class<P, OP_MAX_I64>(p, acc, wave, li, sq, part, n_parts);
  wave_reduce_class<P, OP_MAX_U64>(p, acc, wave, li, sq, part, n_parts);
  // the error lane (codes 1 = overflow, 2 = division by zero; any non-zero code fails the query): two ballots
  const uint32_t wave_err = (__builtin_amdgcn_ballot_w64((err & 1u) != 0) ? 1u : 0u) | (__builtin_amdgcn_ballot_w64((err & 2u) != 0) ? 2u : 0u);
  if (wl == 0) publish_partial(p, P::NG * P::K, wave_err, part, n_parts);
  wave_lds_sync(); // the reads above precede the re-initialisation of the columns they read
}

template <class P> __device__ __forceinline__ void fused_scan_body_lds(const ScanParams &p) {
  constexpr int NG = P::NG, K = P::K, U = P::U;
  constexpr LaneOpTable<P> ops{};
  __shared__ uint64_t acc[NG * K][kBlock]; // acc[lane][thread]: every thread owns a column

  const uint32_t tid = threadIdx.x, wave = tid >> 6, wl = tid & 63;
  piggyback_fold<P>(p);

  // The canonical unit of the reduction is (tile, wave): the rows of a tile that a wave's lanes own (128 consecutive
  // rows of every 512-row step) → one partial per lane of the plan, whatever the launch geometry.  Workgroup b of g
  // streams the consecutive tiles [b·n/g, (b+1)·n/g) — the host picks g (≈ one workgroup per CU, engine.cpp:
  // pick_scan_grid); its four waves never synchronise: each reduces its own 64 columns of the image
  // (DS operations of one wave execute in order) and moves on.  The image of a finished tile is reduced AFTER the
  // first loads of the next tile have been requested (same iteration: nothing loaded lives across the back-edge or a
  // join point), so the reduction hides behind their latency; the next tile's descriptor is fetched a tile ahead.
  uint32_t tile = (uint32_t)((uint64_t)blockIdx.x * p.n_tiles / gridDim.x);
  const uint32_t tile_end = (uint32_t)((uint64_t)(blockIdx.x + 1) * p.n_tiles / gridDim.x);
  if (tile >= tile_end) return;
  const uint32_t n_parts = p.n_tiles * kLdsParts;
#pragma unroll
  for (int l = 0; l < NG * K; ++l) acc[l][tid] = lane_identity(ops.v[l]);
  uint32_t err = 0, done_err = 0, done_tile = 0;
  constexpr int n_first_only = plan_first_only_index<P>(K);
  static_assert(n_first_only == 0 || NG <= 64, "the first-only masks hold one bit per group");
  uint64_t seen[n_first_only > 0 ? n_first_only : 1] = {};
  bool fresh = false; // the image holds a finished tile that awaits its reduction
  // (Measured, r03: requesting the NEXT group of U steps before this one is accumulated — a second Loaded[U], free in
  // registers at two waves per SIMD — is slower here: Q1 0.359 → 0.370 ms, the 12-lane state 0.338 → 0.368 ms; the
  // register-state kernel's late form does gain from it.  profiles/r03/lds_rmw_experiment.txt)

  TileDesc td = load_tile_desc(p.tiles, tile);
  TileDesc td_next = load_tile_desc(p.tiles, tile + 1 < tile_end ? tile + 1 : tile);
  uint32_t nsteps = (td.rows + kStepRows - 1) / kStepRows;
  uint32_t s = 0;
  for (;;) {
    Loaded ld[U];
    // issue every load of the unrolled group before the first use (column buffers carry slack past the last
    // tile, so the tail steps may read — and discard — past td.rows)
#pragma unroll
    for (int u = 0; u < U; ++u) load_all<typename P::ColList>(p, td.dev_row + (uint64_t)(s + u) * kStepRows + (uint64_t)tid * kRowsPerThread, ld[u]);
    if (fresh) {
      wave_reduce_image<P>(p, acc, wave, wl, done_tile * kLdsParts + wave, n_parts, done_err);
#pragma unroll
      for (int l = 0; l < NG * K; ++l) acc[l][tid] = lane_identity(ops.v[l]);
#pragma unroll
      for (int i = 0; i < (n_first_only > 0 ? n_first_only : 1); ++i) seen[i] = 0;
      fresh = false;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t row0 = (s + u) * kStepRows + tid * kRowsPerThread;
#pragma unroll
      for (int j = 0; j < kRowsPerThread; ++j) {
        Ctx c{p, ld[u], 0u, td.logical_row + row0 + j};
        const bool in_tile = (row0 + j) < td.rows;
        const bool pass = in_tile & P::Pred::eval(c, j);
        uint32_t gid = P::KeyT::gid(c, j);
        gid = gid < (uint32_t)NG ? gid : 0u; // rows past the tile end carry arbitrary codes
        uint64_t contrib[K];
        contrib[0] = 1;
        if constexpr (P::first) contrib[1] = c.row;
        AggOps<typename P::AggT>::contrib(c, j, contrib + P::BASE);
        err |= (pass ? c.err : 0u) | (in_tile ? c.perr : 0u);
        if (pass) lds_accumulate_row<P>(&acc[gid * K][tid], contrib, gid, seen);
      }
    }
    s += U;
    if (s >= nsteps) { // the tile is complete: its reduction follows the next loads (or the loop's exit)
      done_tile = tile;
      done_err = err;
      err = 0;
      fresh = true;
      if (++tile >= tile_end) break;
      td = td_next;
      nsteps = (td.rows + kStepRows - 1) / kStepRows;
      s = 0;
      if (tile + 1 < tile_end) td_next = load_tile_desc(p.tiles, tile + 1);
    }
  }
  wave_reduce_image<P>(p, acc, wave, wl, done_tile * kLdsParts + wave, n_parts, done_err);
}

// ---- grouped plans with many groups: one accumulator image per workgroup --------------------------------------
// 65 … ~16 000 groups (GROUP BY l_shipdate: 2 526) do not fit per-thread accumulator columns; sorting the rows by key
// and gathering the arguments through the permutation (group_sort.cpp) moves the table several times.  Here a
// persistent workgroup of 1024 threads (16 waves: with ≥ 80 KB of image only one workgroup fits a CU, and it has to
// hide the HBM latency on its own) keeps ONE image img[lane][group] in LDS and every row updates the K slots of its
// group with native DS atomics.  Rows of one wave instruction that meet in a group are serialised by the LDS, in no
// particular order — which is why every lane of such a plan is order-free: integer adds, min / max, and f64 sums as
// exact two-level SumF64X.  The result is then independent of thread, workgroup, tile and rank geometry by
// construction.  Each workgroup leaves its image in global memory ([workgroup][lane][group], coalesced);
// image_fold_kernel combines them.
constexpr int kImgBlock = 1024;
constexpr int kImgStepRows = kImgBlock * kRowsPerThread; // 2048 rows per workgroup step

// copies of an image of `cells` cells of `cell_bytes`: a power of two ≤ 32 that keeps them within 64 KB
constexpr int image_replicas(int cells, int cell_bytes = 8) {
  int r = 1;
  while (r < 32 && (long)(2 * r) * cells * cell_bytes <= 64 * 1024) r *= 2;
  return r;
}
template <bool NARROW> struct ImageCell { using T = uint64_t; };
template <> struct ImageCell<true> { using T = uint32_t; };
// 4-byte cells (Plan::CELL32): integer adds wrap in 32 bits (the lowering has bounded the image's total), the first-row lane is
// an unsigned minimum of row ids below 2^32
template <int OP> __device__ __forceinline__ void lds_accumulate(uint32_t *slot, uint64_t x) {
  static_assert(OP == OP_ADD_I64 || OP == OP_MIN_I64, "4-byte image cells hold counts, bounded integer sums and first rows");
  if constexpr (OP == OP_ADD_I64) (void)__hip_atomic_fetch_add(slot, (uint32_t)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else (void)__hip_atomic_fetch_min(slot, (uint32_t)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <class P, int NGS, int K0 = 0, class Cell> __device__ __forceinline__ void image_accumulate_row(Cell *img, uint32_t gid, const uint64_t *contrib) {
  if constexpr (K0 < P::K) {
    lds_accumulate<plan_lane_op<P>(K0)>(img + K0 * NGS + gid, contrib[K0]);
    image_accumulate_row<P, NGS, K0 + 1>(img, gid, contrib);
  }
}

// When the image of all groups exceeds the LDS the groups are cut into P::PASSES slices of NG groups and the table is
// scanned once per slice (ScanParams::group_base = first group of the launch's slice; rows of other slices are
// skipped): P× the traffic, still far below what sorting the rows costs.
template <class P> __device__ __forceinline__ void image_scan_body(const ScanParams &p) {
  constexpr int K = P::K, U = P::U;
  constexpr int NG = (P::NG + P::PASSES - 1) / P::PASSES; // groups of one slice
  // A small image (a few groups under many lanes, or a few hundred groups) is kept R times and thread t adds into copy
  // t mod R: the 64 rows of a wave instruction that meet in a cell are serialised by the LDS, and with four groups all of
  // them meet (12 + 3 lanes over the 4 groups of Q1: 3.6 ms for SF10 with one copy).  Every lane is order-free, so the
  // copies are combined in any order when the workgroup leaves its image.  R depends on the plan alone.
  using Cell = typename ImageCell<P::CELL32>::T;
  constexpr int R = image_replicas(K * NG, (int)sizeof(Cell));
  __shared__ Cell img_all[R * K * NG]; // [copy][lane][group of the slice]
  __shared__ uint32_t block_err;

  const uint32_t tid = threadIdx.x;
  Cell *img = img_all + (tid % (uint32_t)R) * (uint32_t)(K * NG);
  for (uint32_t i = tid; i < (uint32_t)(R * K * NG); i += kImgBlock) {
    const int op = plan_lane_op<P>((int)((i % (uint32_t)(K * NG)) / NG));
    if constexpr (P::CELL32) img_all[i] = op == OP_MIN_I64 ? 0xFFFFFFFFu : 0u;
    else img_all[i] = lane_identity(op);
  }
  if (tid == 0) block_err = 0;
  __syncthreads();

  // A tile is U steps of 2 048 rows (engine.cpp: pick_tile_rows — tiles are only a work list here), so one batch of
  // loads covers it.  Two register buffers take turns: the loads of the tile after next are requested before a tile is
  // accumulated — the sixteen waves of the one workgroup a CU holds run in step, and with a single buffer every batch
  // paid its latency AND its transfer (0.37 ms for 28 B/row whatever the aggregates; 0.16 ms for 4 B/row).
  uint32_t err = 0;
  auto issue = [&](const TileDesc &td, Loaded (&ld)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) load_all<typename P::ColList>(p, td.dev_row + (uint64_t)u * kImgStepRows + (uint64_t)tid * kRowsPerThread, ld[u]);
  };
  auto accumulate = [&](const TileDesc &td, const Loaded (&ld)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t row0 = u * kImgStepRows + tid * kRowsPerThread;
#pragma unroll
      for (int j = 0; j < kRowsPerThread; ++j) {
        Ctx c{p, ld[u], 0u, td.logical_row + row0 + j};
        const bool in_tile = (row0 + j) < td.rows;
        uint32_t gid = P::KeyT::gid(c, j) - p.group_base; // groups of other slices (and the arbitrary codes of
        const bool mine = gid < (uint32_t)NG;              // rows past the tile end) wrap past NG
        const bool pass = in_tile & mine & P::Pred::eval(c, j);
        gid = mine ? gid : 0u;
        uint64_t contrib[K];
        contrib[0] = 1;
        if constexpr (P::first) contrib[1] = c.row;
        AggOps<typename P::AggT>::contrib(c, j, contrib + P::BASE);
        err |= (pass ? c.err : 0u) | (in_tile ? c.perr : 0u);
        if (pass) image_accumulate_row<P, NG>(img, gid, contrib);
      }
    }
  };
  const uint32_t n = p.n_tiles, g = gridDim.x;
  uint32_t ta = blockIdx.x, tb = ta + g; // the tiles in buffers A and B (≥ n: none)
  TileDesc da = load_tile_desc(p.tiles, ta < n ? ta : 0), db = load_tile_desc(p.tiles, tb < n ? tb : 0);
  Loaded la[U], lb[U];
  if (ta < n) issue(da, la);
  while (ta < n) {
    if (tb < n) issue(db, lb);
    accumulate(da, la);
    ta = tb + g;
    if (ta < n) { da = load_tile_desc(p.tiles, ta); issue(da, la); }
    if (tb < n) accumulate(db, lb);
    tb = ta + g;
    if (tb < n) db = load_tile_desc(p.tiles, tb);
  }
  if (err) atomicOr(&block_err, err);
  __syncthreads();
  uint64_t *out = p.tile_partials + (uint64_t)blockIdx.x * (uint64_t)(K * NG + 1);
  for (uint32_t i = tid; i < (uint32_t)(K * NG); i += kImgBlock) {
    const int op = plan_lane_op<P>((int)(i / NG));
    if constexpr (P::CELL32) { // the image leaves as 64-bit lanes (the fold and everything behind it are the same)
      uint32_t v = img_all[i];
#pragma unroll 1
      for (int r = 1; r < R; ++r) {
        const uint32_t o = img_all[(uint32_t)r * (uint32_t)(K * NG) + i];
        v = op == OP_MIN_I64 ? (o < v ? o : v) : v + o;
      }
      out[i] = op == OP_MIN_I64 ? (v == 0xFFFFFFFFu ? lane_identity(OP_MIN_I64) : (uint64_t)v) : (uint64_t)(int64_t)(int32_t)v;
    } else {
      uint64_t v = img_all[i];
      if constexpr (R > 1) {
#pragma unroll 1
        for (int r = 1; r < R; ++r) v = lane_combine(op, v, img_all[(uint32_t)r * (uint32_t)(K * NG) + i]);
      }
      out[i] = v;
    }
  }
  if (tid == 0) out[K * NG] = block_err;
}

// ---- partitioned GROUP BY: more groups than LDS-sized slices can cover in a few scans ------------------------------
// The sort-based route orders all selected rows by key (radix passes) and then gathers the argument columns at random:
// 9.5 ms for 60 M rows in 2 M groups.  Accumulating into one image in HBM is no better: device-scope atomics run at
// 23.5 × 10⁹ / s whatever the image size (tools/micro/global_atomics.hip).  Here the rows are cut by group-id range
// into partitions whose image fits the LDS: every tile of 32 768 rows is put in partition order (part_scatter_body: a
// count sweep, a workgroup scan, a scatter sweep that writes each selected row's (group within partition, lane
// contributions) at its exact position — no global atomics), and one workgroup per partition then reduces its cells
// of all the tiles in an LDS image (part_reduce_kernel).  Every lane op is order-free (the shared-image lowering), so
// the order inside a cell is immaterial.
constexpr int kMaxParts = 4096;
constexpr int kPartTileRows = 32768; // rows of one (tile, partition) cell's tile: ≥ 8 rows per cell at 4 096 partitions
constexpr int kPartStageLanes = 7;   // records of up to 7 words are sorted by partition in the LDS before they leave (152 KB of LDS)

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding global load and store
// of the wave (s_waitcnt vmcnt(0)) — with six barriers per 2 048-row step that drained the prefetched loads of the next
// step and the record stores of this one at every barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Exclusive scan of cnt[0 .. MP) in place by the B threads of the workgroup (MP / B entries each); returns the total.
template <int B, int MP = kMaxParts> __device__ __forceinline__ uint32_t part_block_scan(uint32_t *cnt, uint32_t *wave_sum /*[B / 64]*/) {
  constexpr int E = MP / B; // entries per thread
  static_assert(E * B == MP && (E == 1 || E % 4 == 0), "one entry or whole uint4s per thread");
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t c[E];
  uint32_t mine = 0;
  if constexpr (E == 1) {
    c[0] = cnt[tid];
    mine = c[0];
  } else {
#pragma unroll
    for (int q = 0; q < E / 4; ++q) {
      const uint4 v = *reinterpret_cast<const uint4 *>(cnt + E * tid + 4 * q);
      c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w;
      mine += v.x + v.y + v.z + v.w;
    }
  }
  uint32_t incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(incl, off, 64);
    incl += lane >= (uint32_t)off ? o : 0u;
  }
  if (lane == 63) wave_sum[wave] = incl;
  lds_barrier();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (int w = 0; w < B / 64; ++w) {
    const uint32_t x = wave_sum[w];
    before += (uint32_t)w < wave ? x : 0u;
    total += x;
  }
  uint32_t ex = before + incl - mine;
  if constexpr (E == 1) {
    cnt[tid] = ex;
  } else {
#pragma unroll
    for (int q = 0; q < E / 4; ++q) {
      uint4 v;
      v.x = ex; ex += c[4 * q];
      v.y = ex; ex += c[4 * q + 1];
      v.z = ex; ex += c[4 * q + 2];
      v.w = ex; ex += c[4 * q + 3];
      *reinterpret_cast<uint4 *>(cnt + E * tid + 4 * q) = v;
    }
  }
  lds_barrier();
  return total;
}

// One workgroup per tile, two sweeps over its rows (the second finds the columns in the L2).
//   sweep 1: rows of the tile per partition (predicate + keys) → a workgroup scan → where each partition's cell of this
//            tile starts.  The cells of a tile are neighbours: tile t owns the records [t·32 768, (t + 1)·32 768) and
//            lays its cells out in partition order inside — a workgroup writes into one 1.3 MB window instead of into
//            every partition's region of the whole record array (489 cells spread over 2.4 GB: the translation misses
//            of a wide radix fan-out made that scatter 1.74 ms for 60 M rows), and no global scan is needed.  The
//            cell table [tile][partition] goes to global memory for the reduction.
//   sweep 2: a record = K − 1 words: [0] the group within the partition and the row's position in the tile (lanes 0
//            and 1 — rows, smallest row id — need no more), [l − 1] lane l ≥ 2.
//            Writing each record from the thread that made it sends the 64 lanes of every store to 64 different lines;
//            so the 2 048 records of a step are first put in partition order in the LDS — a counter per partition ranks
//            them, a scan places the partitions — and leave as runs of consecutive words (records of ≤ 7 words; wider
//            ones go straight from their thread).
//   LINES    (records of ≤ 4 words, ≤ 512 partitions, 1 024 threads): whole 128-byte lines only.  The runs a step leaves are a few
//            records long and start anywhere; here the words of a partition's cell behind its last complete line wait in the LDS —
//            a ring of 16 words per partition, indexed by the word's position in its line — and leave when their line is complete
//            (or the tile ends): per step the words of the ring whose line completes (phase A) and the new words below the
//            partition's last line boundary (phase B) go out, the rest of the new words enter the ring.
//            Measured (profiles/r03/part_lines.txt, part_store_exp.txt): 1.355 ms against 1.403 — the record stores cost 0.7 ms
//            whether they leave as whole lines or as runs of a few records, and 0.06 ms when they land in a window the L2 holds:
//            what bounds them is the DRAM's rate for lines scattered over a megabyte per workgroup, not partial lines.
template <class P, int B = 1024, bool LINES = false> __device__ __forceinline__ void part_scatter_body(const ScanParams &p) {
  constexpr int kStep = B * kRowsPerThread; // rows of one workgroup step
  constexpr int MP = LINES ? 1024 : kMaxParts; // partition counters in the LDS
  constexpr int kLineParts = 512;              // LINES: partitions at most (the host admits the form by np)
  static_assert(P::first, "partitioned plans keep the first row of every group");
  static_assert(kPartTileRows == 1 << 15, "part_reduce_kernel packs (tile << 15 | row within the tile)");
  constexpr int K = P::K - 1; // words of a record: [0] group within the partition | row within the tile << 32, then lanes 2 …
  constexpr bool STAGED = K <= kPartStageLanes;
  static_assert(!LINES || (B == 1024 && STAGED && K <= 4), "the line form: short records, one partition counter per thread");
  __shared__ __attribute__((aligned(16))) uint32_t cell[MP];  // next record position of each partition's cell of this tile
  __shared__ __attribute__((aligned(16))) uint32_t scnt[STAGED ? MP : 4]; // the step's records per partition → where they start in `stage`
  __shared__ uint32_t wave_sum[B / 64];
  __shared__ uint32_t dest[STAGED ? kStep : 1];       // record position of each staged slot
  __shared__ uint64_t stage[STAGED ? kStep * K : 1];
  // LINES: word positions are relative to the tile's window (tile_base · K is a multiple of 16: a line of the record array)
  __shared__ uint16_t slot_part[LINES ? kStep : 1];           // partition of each staged slot
  __shared__ uint64_t ring[LINES ? kLineParts * 16 : 1];      // [partition][word position & 15]: words waiting for their line
  __shared__ uint32_t carried[LINES ? kLineParts : 1];        // words of the partition in the ring (< 16)
  __shared__ uint32_t lim[LINES ? kLineParts : 1];            // this step: words below leave, the others enter the ring
  __shared__ uint32_t old_from[LINES ? kLineParts : 1];       // this step: the ring's words [old_from, old_from + old_n) leave
  __shared__ uint32_t old_n[LINES ? kLineParts : 1];
  if constexpr (LINES) {
    if (p.part_np > (uint32_t)kLineParts) return; // (the host admits the form by np: no fault if it ever did not)
  }
  const uint32_t tid = threadIdx.x, tile = blockIdx.x, np = p.part_np;
  const TileDesc td = load_tile_desc(p.tiles, tile);
  for (uint32_t i = tid; i < (uint32_t)MP; i += B) {
    cell[i] = 0u;
    if constexpr (STAGED) scnt[i] = 0u;
  }
  if constexpr (LINES)
    for (uint32_t i = tid; i < (uint32_t)kLineParts; i += B) carried[i] = 0u;
  lds_barrier();
  uint32_t err = 0;
  const uint32_t mask = (1u << p.part_shift) - 1u;
  const uint32_t nsteps = (td.rows + kStep - 1) / kStep;
  auto each_step = [&](auto &&rows_of) { // two steps of loads in flight
    uint32_t s = 0;
    for (; s + 1 < nsteps; s += 2) {
      Loaded a, b;
      load_all<typename P::ColList>(p, td.dev_row + (uint64_t)s * kStep + (uint64_t)tid * kRowsPerThread, a);
      load_all<typename P::ColList>(p, td.dev_row + (uint64_t)(s + 1) * kStep + (uint64_t)tid * kRowsPerThread, b);
      rows_of(s, a);
      rows_of(s + 1, b);
    }
    if (s < nsteps) {
      Loaded a;
      load_all<typename P::ColList>(p, td.dev_row + (uint64_t)s * kStep + (uint64_t)tid * kRowsPerThread, a);
      rows_of(s, a);
    }
  };
  // ---- sweep 1 ------------------------------------------------------------------------------------------------------
  each_step([&](uint32_t s, const Loaded &ld) {
    const uint32_t row0 = s * kStep + tid * kRowsPerThread;
#pragma unroll
    for (int j = 0; j < kRowsPerThread; ++j) {
      Ctx c{p, ld, 0u, td.logical_row + row0 + j};
      const bool in_tile = (row0 + j) < td.rows;
      const uint32_t gid = P::KeyT::gid(c, j);
      const bool pass = in_tile & (gid < (uint32_t)P::NG) & P::Pred::eval(c, j);
      if (pass) (void)__hip_atomic_fetch_add(&cell[gid >> p.part_shift], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  });
  lds_barrier();
  const uint32_t tile_rows = part_block_scan<B, MP>(cell, wave_sum);
  const uint32_t tile_base = tile * (uint32_t)kPartTileRows;
  uint32_t *cells_out = p.part_hist + (uint64_t)tile * (np + 1);
  for (uint32_t i = tid; i < (uint32_t)MP; i += B) {
    const uint32_t at = tile_base + cell[i];
    cell[i] = at;
    if (i < np) cells_out[i] = at;
  }
  if (tid == 0) cells_out[np] = tile_base + tile_rows;
  lds_barrier();
  // ---- sweep 2 ------------------------------------------------------------------------------------------------------
  each_step([&](uint32_t s, const Loaded &ld) {
    const uint32_t row0 = s * kStep + tid * kRowsPerThread;
    uint64_t contrib[kRowsPerThread][K];
    uint32_t part[kRowsPerThread], rank[kRowsPerThread];
    bool pass[kRowsPerThread];
#pragma unroll
    for (int j = 0; j < kRowsPerThread; ++j) {
      Ctx c{p, ld, 0u, td.logical_row + row0 + j};
      const bool in_tile = (row0 + j) < td.rows;
      const uint32_t gid = P::KeyT::gid(c, j);
      pass[j] = in_tile & (gid < (uint32_t)P::NG) & P::Pred::eval(c, j);
      part[j] = pass[j] ? gid >> p.part_shift : 0u;
      // lane 0 counts rows and lane 1 is the smallest row id: the row's position inside the tile (15 bits) rides with
      // the group id, and the reduction — which knows the tile of every cell — adds the tile's first row id
      AggOps<typename P::AggT>::contrib(c, j, contrib[j] + 1);
      contrib[j][0] = (uint64_t)(gid & mask) | ((uint64_t)(row0 + j) << 32);
      err |= (pass[j] ? c.err : 0u) | (in_tile ? c.perr : 0u);
      rank[j] = 0;
      if (pass[j]) rank[j] = __hip_atomic_fetch_add(STAGED ? &scnt[part[j]] : &cell[part[j]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if constexpr (!STAGED) {
#pragma unroll
      for (int j = 0; j < kRowsPerThread; ++j) {
        if (!pass[j]) continue;
        uint64_t *rec = p.part_val + (uint64_t)rank[j] * K;
#pragma unroll
        for (int l = 0; l < K; ++l) rec[l] = contrib[j][l];
      }
    } else {
      lds_barrier();
      const uint32_t total = part_block_scan<B, MP>(scnt, wave_sum); // scnt: records per partition → first slot of each partition
#pragma unroll
      for (int j = 0; j < kRowsPerThread; ++j) {
        if (!pass[j]) continue;
        const uint32_t slot = scnt[part[j]] + rank[j];
        dest[slot] = cell[part[j]] + rank[j];
        if constexpr (LINES) slot_part[slot] = (uint16_t)part[j];
#pragma unroll
        for (int l = 0; l < K; ++l) stage[slot * K + l] = contrib[j][l];
      }
      lds_barrier();
      if constexpr (LINES) {
        // per partition: where its words stand after this step, and up to which word they leave (a line boundary; the end of
        // the cell on the tile's last step)
        if (tid < np) {
          const uint32_t n = (tid + 1 < (uint32_t)MP ? scnt[tid + 1] : total) - scnt[tid];
          const uint32_t w0 = (cell[tid] - tile_base) * K, end = w0 + n * K, from = w0 - carried[tid];
          const uint32_t boundary = s + 1 == nsteps ? end : (end & ~15u);
          const bool leaves = boundary > from;
          lim[tid] = leaves ? boundary : from;
          old_from[tid] = from;
          old_n[tid] = leaves ? carried[tid] : 0u;
          carried[tid] = leaves ? end - boundary : end - from;
          cell[tid] += n;
        }
        lds_barrier();
        uint64_t *win = p.part_val + (uint64_t)tile_base * K; // the tile's window of the record array
        for (uint32_t q = tid; q < np * 16u; q += B) { // phase A: the ring's words whose line is complete now
          const uint32_t i = q >> 4, j = q & 15u;
          if (j < old_n[i]) {
            const uint32_t at = old_from[i] + j;
            win[at] = ring[i * 16u + (at & 15u)];
          }
        }
        lds_barrier();
        for (uint32_t w = tid; w < total * K; w += B) { // phase B: this step's words
          const uint32_t slot = w / K, l = w - slot * K, i = slot_part[slot];
          const uint32_t at = (dest[slot] - tile_base) * K + l;
          const uint64_t v = stage[w];
          if (at < lim[i]) win[at] = v;
          else ring[i * 16u + (at & 15u)] = v;
        }
        lds_barrier();
        for (uint32_t i = tid; i < (uint32_t)MP; i += B) scnt[i] = 0u;
        lds_barrier();
        return;
      }
      // the cells advance by what the step put into them; the staged words leave in order
      for (uint32_t i = tid; i < np; i += B) cell[i] += (i + 1 < (uint32_t)kMaxParts ? scnt[i + 1] : total) - scnt[i];
      for (uint32_t w = tid; w < total * K; w += B) {
        const uint32_t slot = w / K, l = w - slot * K;
        p.part_val[(uint64_t)dest[slot] * K + l] = stage[w]; // (a non-temporal store here: 2.8 ms instead of 1.8 — the L2 merges the runs into lines)
      }
      lds_barrier();
      for (uint32_t i = tid; i < (uint32_t)kMaxParts; i += B) scnt[i] = 0u;
      lds_barrier();
    }
  });
  if (err) atomicOr(p.part_err, err);
}

// One workgroup per partition: its cell of every tile → LDS image [kernel lane][group of the partition] → rows of the
// group-major result [group][exchange lane] (the layout finalize_value reads; a fixed-point sum is one lane in the
// image, low 32 bits + high part in the result: lane_src / lane_xf as in image_fold_kernel).
template <int V> struct IntC { static constexpr int value = V; };
struct PartReduceParams {
  const uint32_t *offsets; // [n_tiles][np + 1]: where the cell of (tile, partition) starts; [np]: where the tile's records end
  const uint64_t *val;     // records of kl − 1 words: [0] group within the partition | row within the tile << 32, [l − 1] kernel lane l ≥ 2
  const TileDesc *tiles;   // (the first row id of every tile)
  uint64_t *out;           // [ng][k]
  const uint8_t *lane_ops; // [kl] ops of the kernel lanes
  const uint8_t *lane_src, *lane_xf; // [k]
  uint32_t n_tiles, np, ngs, ng, kl, k;
  uint32_t deep; // eight cells in flight per wave instead of four (an image beyond 64 KB leaves a CU one workgroup: 16 waves)
  uint32_t part0; // first partition of this launch (the copy-out of a range of partitions may run beside the reduction of the next)
};
__global__ __launch_bounds__(1024) void part_reduce_kernel(const PartReduceParams f) {
  // [kl − 1][ngs]: lanes 0 and 1 share the first cell of a group — its rows in the low 32 bits, the smallest (tile << 15 | row
  // within the tile) in the high 32 (a table has < 2^32 record positions; both are 32-bit DS atomics) — then kernel lanes 2 …:
  // one 8-byte cell fewer per group is what lets 4 096 groups of a five-lane state share 128 KB (half the partitions, runs of
  // twice the length in the scatter)
  extern __shared__ uint64_t part_img[];
  const uint32_t tid = threadIdx.x, part = blockIdx.x + f.part0, lane = tid & 63, wave = tid >> 6;
  for (uint32_t i = tid; i < (f.kl - 1) * f.ngs; i += 1024) part_img[i] = i < f.ngs ? 0xFFFFFFFF00000000ull : lane_identity((int)f.lane_ops[i / f.ngs + 1]);
  __syncthreads();
  // a wave per (tile, partition) cell, its words read in order (a thread per record read with a stride of kl words:
  // 1.0 ms for 60 M records of 5 words, against 0.46 ms for the same bytes read in order); the group of a word's record
  // is word 0 of that record — the same or the neighbouring cache line.  Four cells in flight per wave.
  const uint32_t rw = f.kl - 1; // words of a record
  auto words_of = [&](uint32_t t, uint64_t *w_begin, uint32_t *n_words, uint64_t *tile) {
    const uint32_t *cells = f.offsets + (uint64_t)t * (f.np + 1) + part;
    const uint32_t b = cells[0], e = cells[1];
    *w_begin = (uint64_t)b * rw;
    *n_words = (e - b) * rw;
    *tile = t;
  };
  // word j of a record, its head word, the tile of its cell
  auto accumulate = [&](uint32_t j, uint64_t head, uint64_t v, uint64_t tile) {
    const uint32_t g = (uint32_t)head;
    if (j == 0) { // lanes 0 and 1: one more row; the smallest position
      uint32_t *cell = reinterpret_cast<uint32_t *>(part_img + g);
      (void)__hip_atomic_fetch_add(cell, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      (void)__hip_atomic_fetch_min(cell + 1, ((uint32_t)tile << 15) | (uint32_t)(head >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return;
    }
    const uint32_t l = j + 1;
    uint64_t *slot = part_img + (uint64_t)(l - 1) * f.ngs + g;
    switch ((int)f.lane_ops[l]) {
    case OP_ADD_F64: lds_accumulate<OP_ADD_F64>(slot, v); break;
    case OP_ADD_I64: lds_accumulate<OP_ADD_I64>(slot, v); break;
    case OP_MIN_I64: lds_accumulate<OP_MIN_I64>(slot, v); break;
    case OP_MAX_I64: lds_accumulate<OP_MAX_I64>(slot, v); break;
    default: lds_accumulate<OP_MAX_U64>(slot, v); break;
    }
  };
  // cells in flight per wave: four when two workgroups share a CU (eight: 1.04 ms instead of 0.79), eight when the image leaves room for one
  auto walk = [&](auto cells) {
    constexpr int kC = decltype(cells)::value;
    for (uint32_t t = wave; t < f.n_tiles; t += 16 * (uint32_t)kC) {
      uint64_t wb[kC], cell_tile[kC];
      uint32_t nw[kC], n_max = 0;
#pragma unroll
      for (int c = 0; c < kC; ++c) {
        wb[c] = 0;
        nw[c] = 0;
        cell_tile[c] = 0;
        if (t + 16 * c < f.n_tiles) words_of(t + 16 * c, &wb[c], &nw[c], &cell_tile[c]);
        n_max = nw[c] > n_max ? nw[c] : n_max;
      }
      for (uint32_t i = lane; i < n_max; i += 64) {
        const uint32_t j = i % rw;
        uint64_t v[kC], head[kC];
#pragma unroll
        for (int c = 0; c < kC; ++c) {
          const bool live = i < nw[c];
          v[c] = live ? f.val[wb[c] + i] : 0;
          head[c] = live ? f.val[wb[c] + i - j] : 0;
        }
#pragma unroll
        for (int c = 0; c < kC; ++c)
          if (i < nw[c]) accumulate(j, head[c], v[c], cell_tile[c]);
      }
    }
  };
  if (f.deep) walk(IntC<8>{});
  else walk(IntC<4>{});
  __syncthreads();
  const uint64_t g0 = (uint64_t)part * f.ngs;
  for (uint32_t i = tid; i < f.ngs * f.k; i += 1024) {
    const uint32_t g = i / f.k, kk = i % f.k;
    if (g0 + g >= f.ng) break;
    const uint32_t src = f.lane_src[kk];
    uint64_t x;
    if (src == 0) x = (uint32_t)part_img[g]; // rows
    else if (src == 1) { // the smallest row id: the tile's first row id + the row's position in the tile
      const uint32_t at = (uint32_t)(part_img[g] >> 32);
      x = at == 0xFFFFFFFFu ? 0x7FFFFFFFFFFFFFFFull : f.tiles[at >> 15].logical_row + (at & 0x7FFFu);
    } else x = part_img[(uint64_t)(src - 1) * f.ngs + g];
    const uint32_t xf = f.lane_xf[kk];
    f.out[(g0 + g) * f.k + kk] = xf == 1 ? (x & 0xFFFFFFFFull) : xf == 2 ? (uint64_t)((int64_t)x >> 32) : x;
  }
}

// Workgroup images [pass][n_wg][K·NGS + 1] (lane-major, NGS groups per slice) → the exchange image [kOctants][NG·K + 1] (group-major, the layout
// of every other plan): the first octant this rank owns receives the combined image; its other octants hold the lane
// identities and the octants of other ranks zero (both written once, when the query is prepared) — so the int64-sum
// all-reduce and the host's octant fold work unchanged.  Every lane op is order-free here, so the order over
// workgroups (and over ranks) is immaterial.
struct ImageFoldParams {
  const uint64_t *partials;
  uint64_t *exchange;
  const uint8_t *lane_ops; // [K]: op of lane k of a group
  uint32_t n_wg, ng, k, owned_mask;
  uint32_t passes, ngs; // slices of the groups, groups per slice
  uint32_t kl;          // lanes per group of the kernel's image (≤ k: a fixed-point sum is one lane there, two here)
  const uint8_t *lane_src, *lane_xf; // [K]: exchange lane k = xf(kernel lane src): 0 as is, 1 low 32 bits, 2 high part
};
// 32 cells (lane k, group g) per workgroup, 8 threads per cell: thread (cell, part) combines the workgroup images
// part, part + 8, … — four loads in flight — and the 8 parts meet in the LDS.  (One thread per cell walking all the
// images one dependent load at a time took 0.13 ms for 256 images: a third of the whole GROUP BY.)
constexpr uint32_t kFoldCells = 32, kFoldParts = 8;
__global__ __launch_bounds__(256) void image_fold_kernel(const ImageFoldParams f) {
  __shared__ uint64_t part_v[kFoldParts][kFoldCells];
  const uint32_t lanes = f.ng * f.k + 1, slice = f.ngs * f.kl + 1; // exchange lanes; words of one workgroup image
  const uint32_t cell = threadIdx.x % kFoldCells, part = threadIdx.x / kFoldCells;
  const uint32_t i = blockIdx.x * kFoldCells + cell; // (lane k, group g) in lane-major order, or the error lane
  const uint32_t o = (uint32_t)__builtin_ctz(f.owned_mask | (1u << kOctants)); // first owned octant
  if (o >= (uint32_t)kOctants) return;
  const bool live = i < lanes, err_lane = i == lanes - 1;
  const uint32_t k = live && !err_lane ? i / f.ng : 0, g = live && !err_lane ? i % f.ng : 0, pass = g / f.ngs, gs = g % f.ngs;
  const int op = err_lane ? OP_MAX_U64 : (int)f.lane_ops[k];
  const uint32_t xf = err_lane ? 0u : f.lane_xf[k];
  // error lane: the largest code any workgroup of any pass reported (the last word of every image)
  const uint64_t *src = err_lane ? f.partials + slice - 1 : f.partials + (uint64_t)pass * f.n_wg * slice + (uint64_t)f.lane_src[k] * f.ngs + gs;
  const uint32_t n_img = err_lane ? f.passes * f.n_wg : f.n_wg;
  uint64_t v = lane_identity(op);
  if (live) {
    uint32_t w = part;
    for (; w + 3 * kFoldParts < n_img; w += 4 * kFoldParts) {
      uint64_t x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = src[(uint64_t)(w + u * kFoldParts) * slice];
#pragma unroll
      for (int u = 0; u < 4; ++u) v = lane_combine(op, v, xf == 1 ? (x[u] & 0xFFFFFFFFull) : xf == 2 ? (uint64_t)((int64_t)x[u] >> 32) : x[u]);
    }
    for (; w < n_img; w += kFoldParts) {
      const uint64_t x = src[(uint64_t)w * slice];
      v = lane_combine(op, v, xf == 1 ? (x & 0xFFFFFFFFull) : xf == 2 ? (uint64_t)((int64_t)x >> 32) : x);
    }
  }
  part_v[part][cell] = v;
  __syncthreads();
  if (part == 0 && live) {
#pragma unroll
    for (uint32_t q = 1; q < kFoldParts; ++q) v = lane_combine(op, v, part_v[q][cell]);
    f.exchange[(uint64_t)o * lanes + (err_lane ? i : (uint64_t)g * f.k + k)] = v; // → [group][lane]
  }
}

template <class P> __device__ __forceinline__ void fused_scan_body(const ScanParams &p) {
  if constexpr (P::ACC == 2) image_scan_body<P>(p);
  else if constexpr (P::ACC == 1) fused_scan_body_lds<P>(p);
  else fused_scan_body_reg<P>(p);
}

template <class P> __global__ __launch_bounds__(kBlock) void fused_scan_kernel(const ScanParams p) { fused_scan_body<P>(p); }

// Standalone fold (flush of the last execution, and of tables with fewer tiles than fold workgroups).
// grid = (kOctants, ceil(lanes / 4)).
__global__ __launch_bounds__(kBlock) void fold_octants_kernel(const FoldParams f) {
  const uint32_t lane = blockIdx.y * (kBlock / 64) + (threadIdx.x >> 6);
  if (lane < f.lanes) fold_one_lane(f.tile_partials, f.exchange, f.octant_tile_begin, f.owned_mask, f.n_tiles, f.lanes, blockIdx.x, lane, f.lane_ops[lane], f.parts_per_tile);
}

} // namespace llkv
