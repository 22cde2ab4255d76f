"""ctypes mirror of include/llkv_hip.h plus plan builders named after the reference's
expression vocabulary (llkv-expr: ``Filter``, ``Operator``, ``Expr``; llkv-aggregate:
``AggregateSpec``), so tests read like the reference's own tests
(e.g. llkv-table/src/table.rs:2035-2355).

Everything here is plain data; no device or library is touched.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import ClassVar, List, Optional, Sequence, Tuple, Union

# ----------------------------------------------------------------------------- enums
OK, INVALID_ARGUMENT, INTERNAL, NOT_FOUND, UNSUPPORTED, NO_DEVICE, PREDICATE_BUILD = range(7)
STATUS_NAMES = {
    0: "Ok", 1: "InvalidArgumentError", 2: "Internal", 3: "NotFound", 4: "Unsupported",
    5: "NoDevice", 6: "PredicateBuild",
}

DT_NULL, DT_INT64, DT_FLOAT64, DT_INT32, DT_DATE32, DT_UINT64, DT_UINT32, DT_FLOAT32, DT_UTF8, DT_BOOLEAN, DT_DECIMAL128 = range(11)
DT_NAMES = {0: "Null", 1: "Int64", 2: "Float64", 3: "Int32", 4: "Date32", 5: "UInt64", 6: "UInt32",
            7: "Float32", 8: "Utf8", 9: "Boolean", 10: "Decimal128"}
NUMPY_OF_DTYPE = {DT_INT64: "int64", DT_FLOAT64: "float64", DT_INT32: "int32", DT_DATE32: "int32",
                  DT_UINT64: "uint64", DT_UINT32: "uint32", DT_FLOAT32: "float32", DT_BOOLEAN: "uint8"}



def i128_buffer(values) -> "np.ndarray":
    """Python ints → arrow Decimal128 raw buffer: (n, 2) uint64, little endian (lo, hi)."""
    import numpy as np
    out = np.empty((len(values), 2), dtype=np.uint64)
    for i, v in enumerate(values):
        u = int(v) & ((1 << 128) - 1)
        out[i, 0], out[i, 1] = u & 0xFFFFFFFFFFFFFFFF, u >> 64
    return out


def i128_buffer_from_i64(values) -> "np.ndarray":
    """int64 raw values → arrow Decimal128 raw buffer (n, 2) uint64: sign-extended, vectorised (DECIMAL(15,2) money columns)."""
    import numpy as np
    v = np.ascontiguousarray(values, dtype=np.int64)
    out = np.empty((len(v), 2), dtype=np.uint64)
    out[:, 0] = v.view(np.uint64)
    out[:, 1] = (v >> 63).view(np.uint64)
    return out


def i128_from_words(lo: int, hi: int) -> int:
    """(low u64 bits, high i64) → Python int."""
    return (int(hi) << 64) | (int(lo) & 0xFFFFFFFFFFFFFFFF)


LIT_NULL, LIT_INT128, LIT_FLOAT64, LIT_DECIMAL128, LIT_BOOLEAN, LIT_STRING, LIT_DATE32 = range(7)
OP_EQUALS, OP_RANGE, OP_GT, OP_GE, OP_LT, OP_LE, OP_IN, OP_IS_NULL, OP_IS_NOT_NULL, OP_MVCC_VISIBLE, OP_COMPARE, OP_IN_LIST, OP_IS_NULL_EXPR, OP_STARTS_WITH, OP_ENDS_WITH, OP_CONTAINS = range(1, 17)
CMP_EQ, CMP_NOT_EQ, CMP_LT, CMP_LT_EQ, CMP_GT, CMP_GT_EQ = range(1, 7)
BOUND_UNBOUNDED, BOUND_INCLUDED, BOUND_EXCLUDED = range(3)
EVAL_PUSH_PREDICATE, EVAL_PUSH_LITERAL, EVAL_AND, EVAL_OR, EVAL_NOT = range(1, 6)
TOK_COLUMN, TOK_LITERAL, TOK_BINARY = range(1, 4)
BIN_ADD, BIN_SUB, BIN_MUL, BIN_DIV, BIN_MOD = range(1, 6)
AGG_COUNT_STAR, AGG_COUNT, AGG_SUM, AGG_TOTAL, AGG_AVG, AGG_MIN, AGG_MAX, AGG_COUNT_NULLS = range(1, 9)
JOIN_INNER, JOIN_LEFT, JOIN_RIGHT, JOIN_FULL, JOIN_SEMI, JOIN_ANTI = range(6)
JOIN_KEYS_TABLE, JOIN_KEYS_EXECUTOR = 0, 1  # llkv_join_key_rules


# --------------------------------------------------------------------------- structs
class CLiteral(C.Structure):
    _fields_ = [("tag", C.c_int32), ("scale", C.c_int32), ("lo", C.c_uint64), ("hi", C.c_int64),
                ("f64", C.c_double), ("str", C.c_char_p)]


class CExprToken(C.Structure):
    _fields_ = [("kind", C.c_int32), ("binop", C.c_int32), ("field_id", C.c_uint32), ("literal", CLiteral)]


class CFilter(C.Structure):
    _fields_ = [("field_id", C.c_uint32), ("op", C.c_int32), ("value", CLiteral),
                ("lower_kind", C.c_int32), ("lower", CLiteral), ("upper_kind", C.c_int32),
                ("upper", CLiteral), ("in_list", C.POINTER(CLiteral)), ("in_len", C.c_uint32),
                ("cmp_op", C.c_int32), ("cmp_left", C.POINTER(CExprToken)), ("cmp_left_len", C.c_uint32),
                ("cmp_right", C.POINTER(CExprToken)), ("cmp_right_len", C.c_uint32),
                ("list_exprs", C.POINTER(C.POINTER(CExprToken))), ("list_expr_lens", C.POINTER(C.c_uint32)), ("list_len", C.c_uint32),
                ("negated", C.c_int32), ("case_sensitive", C.c_int32)]


class CEvalOp(C.Structure):
    _fields_ = [("op", C.c_int32), ("arg", C.c_uint32)]


class CAggregateSpec(C.Structure):
    _fields_ = [("kind", C.c_int32), ("distinct", C.c_int32), ("expr", C.POINTER(CExprToken)),
                ("expr_len", C.c_uint32), ("alias", C.c_char_p)]


class CValue(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("is_null", C.c_int32), ("i64", C.c_int64), ("f64", C.c_double),
                ("str", C.c_char_p), ("i64_hi", C.c_int64), ("precision", C.c_int32), ("scale", C.c_int32)]


class CProjection(C.Structure):
    _fields_ = [("computed", C.c_int32), ("field_id", C.c_uint32), ("expr", C.POINTER(CExprToken)),
                ("expr_len", C.c_uint32), ("alias", C.c_char_p)]


class CScanOptions(C.Structure):
    _fields_ = [("include_nulls", C.c_int32), ("include_row_ids", C.c_int32), ("order_enabled", C.c_int32), ("order_field", C.c_uint32),
                ("order_descending", C.c_int32), ("order_nulls_first", C.c_int32), ("order_transform", C.c_int32)]


ORDER_IDENTITY_INT64, ORDER_IDENTITY_INT32, ORDER_IDENTITY_UTF8, ORDER_CAST_UTF8_TO_INTEGER = range(4)


def scan_options(include_nulls=False, include_row_ids=False, order=None) -> "CScanOptions":
    """order = (field_id, descending, nulls_first, transform) or None (ScanOrderSpec)."""
    o = CScanOptions(int(include_nulls), int(include_row_ids))
    if order is not None:
        o.order_enabled, o.order_field, o.order_descending, o.order_nulls_first, o.order_transform = 1, order[0], int(order[1]), int(order[2]), order[3]
    return o


class CColumnView(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("values", C.c_void_p), ("validity", C.POINTER(C.c_uint8)),
                ("dictionary", C.POINTER(C.c_char_p)), ("precision", C.c_int32), ("scale", C.c_int32)]


class CBatchView(C.Structure):
    _fields_ = [("num_rows", C.c_uint64), ("num_columns", C.c_uint32), ("columns", C.POINTER(CColumnView)),
                ("row_ids", C.POINTER(C.c_uint64))]


class CJoinKey(C.Structure):
    _fields_ = [("left_field", C.c_uint32), ("right_field", C.c_uint32), ("null_equals_null", C.c_int32)]


class CJoinOptions(C.Structure):
    _fields_ = [("join_type", C.c_int32), ("batch_size", C.c_uint64), ("key_rules", C.c_int32)]


class CJoinColumn(C.Structure):
    _fields_ = [("field_id", C.c_uint32), ("name", C.c_char_p)]


class CJoinOutput(C.Structure):
    _fields_ = [("left_columns", C.POINTER(CJoinColumn)), ("n_left", C.c_uint32),
                ("right_columns", C.POINTER(CJoinColumn)), ("n_right", C.c_uint32)]


def join_output(left_columns, right_columns):
    """llkv_join_output from two lists of (field_id, name); returns (struct, keep-alive)."""
    la = (CJoinColumn * max(1, len(left_columns)))()
    ra = (CJoinColumn * max(1, len(right_columns)))()
    for arr, cols in ((la, left_columns), (ra, right_columns)):
        for i, (fid, name) in enumerate(cols):
            arr[i].field_id, arr[i].name = fid, name.encode()
    return CJoinOutput(la, len(left_columns), ra, len(right_columns)), (la, ra)


class CColumnChunks(C.Structure):
    _fields_ = [("field_id", C.c_uint32), ("values", C.POINTER(C.c_void_p)), ("offsets", C.POINTER(C.c_void_p)), ("data", C.POINTER(C.c_void_p)),
                ("validity", C.POINTER(C.c_void_p))]


class CJoinOrderKey(C.Structure):
    _fields_ = [("kind", C.c_int32), ("index", C.c_uint32), ("descending", C.c_int32), ("nulls_first", C.c_int32)]


JOIN_ORDER_AGGREGATE, JOIN_ORDER_PAYLOAD, JOIN_ORDER_KEY = 0, 1, 2


class CColumnDesc(C.Structure):
    _fields_ = [("field_id", C.c_uint32), ("dtype", C.c_int32), ("rows", C.c_uint64), ("has_stats", C.c_int32),
                ("min_i", C.c_int64), ("max_i", C.c_int64), ("dict_size", C.c_uint32),
                ("dictionary", C.POINTER(C.c_char_p)), ("nullable", C.c_int32),
                ("precision", C.c_int32), ("scale", C.c_int32), ("has_fstats", C.c_int32), ("f_absmax", C.c_double), ("f_absmin_nz", C.c_double),
                ("f_all_finite", C.c_int32)]


class CJoinSide(C.Structure):
    _fields_ = [("table", C.c_void_p), ("filters", C.POINTER(CFilter)), ("n_filters", C.c_uint32), ("key_field", C.c_uint32)]


class CJoinGroupRow(C.Structure):
    _fields_ = [("key", C.c_int64), ("sum", C.c_double), ("count", C.c_uint64), ("payload", C.c_int64 * 4),
                ("group_index", C.c_uint64)]


class CArr0Desc(C.Structure):
    _fields_ = [("layout", C.c_int32), ("type_code", C.c_int32), ("dtype", C.c_int32), ("reserved", C.c_int32), ("len", C.c_uint64),
                ("payload_offset", C.c_uint64), ("values_offset", C.c_uint64), ("values_len", C.c_uint64), ("offsets_len", C.c_uint64)]


class CChunkMeta(C.Structure):
    _fields_ = [("row_count", C.c_uint64), ("min_val_u64", C.c_uint64), ("max_val_u64", C.c_uint64)]


ON_BATCH = C.CFUNCTYPE(None, C.POINTER(CBatchView), C.c_void_p)
ON_JOIN_BATCH = C.CFUNCTYPE(None, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_uint64, C.c_void_p)
ON_JOIN_RECORD_BATCH = C.CFUNCTYPE(None, C.POINTER(CBatchView), C.POINTER(C.c_char_p), C.c_void_p)


class LlkvError(Exception):
    """Mirror of llkv_result::Error: ``kind`` is the variant name."""

    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status
        self.kind = STATUS_NAMES.get(status, str(status))
        self.message = message


# -------------------------------------------------------------------------- literals
@dataclass(frozen=True)
class Literal:
    """llkv_types::Literal (llkv-types/src/literal.rs)."""
    tag: int
    int_value: int = 0
    float_value: float = 0.0
    scale: int = 0
    string: Optional[str] = None

    @staticmethod
    def of(v) -> "Literal":
        if isinstance(v, Literal):
            return v
        if v is None:
            return Literal(LIT_NULL)
        if isinstance(v, bool):
            return Literal(LIT_BOOLEAN, int_value=int(v))
        if isinstance(v, int):
            return Literal(LIT_INT128, int_value=v)
        if isinstance(v, float):
            return Literal(LIT_FLOAT64, float_value=v)
        if isinstance(v, str):
            return Literal(LIT_STRING, string=v)
        try:
            import numpy as np
            if isinstance(v, np.integer):
                return Literal(LIT_INT128, int_value=int(v))
            if isinstance(v, np.floating):
                return Literal(LIT_FLOAT64, float_value=float(v))
        except ImportError:  # pragma: no cover
            pass
        raise TypeError(f"cannot build a Literal from {type(v)}")

    @staticmethod
    def decimal(raw: int, scale: int) -> "Literal":
        return Literal(LIT_DECIMAL128, int_value=raw, scale=scale)

    @staticmethod
    def date32(days: int) -> "Literal":
        return Literal(LIT_DATE32, int_value=days)

    def to_c(self, keep: list) -> CLiteral:
        c = CLiteral()
        c.tag = self.tag
        c.scale = self.scale
        v = self.int_value & ((1 << 128) - 1)
        c.lo = v & 0xFFFFFFFFFFFFFFFF
        hi = (v >> 64) & 0xFFFFFFFFFFFFFFFF
        c.hi = hi - (1 << 64) if hi >= (1 << 63) else hi
        c.f64 = self.float_value
        if self.string is not None:
            b = self.string.encode()
            keep.append(b)
            c.str = b
        return c


# --------------------------------------------------------------------------- filters
class Bound:
    """std::ops::Bound<Literal>."""

    def __init__(self, kind: int, value=None):
        self.kind = kind
        self.value = None if value is None else Literal.of(value)

    @staticmethod
    def Included(v):
        return Bound(BOUND_INCLUDED, v)

    @staticmethod
    def Excluded(v):
        return Bound(BOUND_EXCLUDED, v)

    Unbounded: "Bound"


Bound.Unbounded = Bound(BOUND_UNBOUNDED)


@dataclass(frozen=True)
class Operator:
    """llkv_expr::Operator (fixed-width subset + IsNull/IsNotNull)."""
    kind: int
    value: Optional[Literal] = None
    lower: Optional[Bound] = None
    upper: Optional[Bound] = None
    values: Tuple[Literal, ...] = ()
    cmp: Optional[tuple] = None  # (left ScalarExpr, CMP_*, right ScalarExpr)
    in_list: Optional[tuple] = None  # (target ScalarExpr, [item ScalarExpr], negated)
    is_null_expr: Optional[tuple] = None  # (ScalarExpr, negated)
    case_sensitive: bool = True  # StartsWith / EndsWith / Contains

    @staticmethod
    def StartsWith(pattern: str, case_sensitive: bool = True):
        return Operator(OP_STARTS_WITH, value=Literal.of(pattern), case_sensitive=case_sensitive)

    @staticmethod
    def EndsWith(pattern: str, case_sensitive: bool = True):
        return Operator(OP_ENDS_WITH, value=Literal.of(pattern), case_sensitive=case_sensitive)

    @staticmethod
    def Contains(pattern: str, case_sensitive: bool = True):
        return Operator(OP_CONTAINS, value=Literal.of(pattern), case_sensitive=case_sensitive)

    @staticmethod
    def Equals(v):
        return Operator(OP_EQUALS, Literal.of(v))

    @staticmethod
    def GreaterThan(v):
        return Operator(OP_GT, Literal.of(v))

    @staticmethod
    def GreaterThanOrEquals(v):
        return Operator(OP_GE, Literal.of(v))

    @staticmethod
    def LessThan(v):
        return Operator(OP_LT, Literal.of(v))

    @staticmethod
    def LessThanOrEquals(v):
        return Operator(OP_LE, Literal.of(v))

    @staticmethod
    def Range(lower: Bound = Bound.Unbounded, upper: Bound = Bound.Unbounded):
        return Operator(OP_RANGE, lower=lower, upper=upper)

    @staticmethod
    def In(values: Sequence):
        return Operator(OP_IN, values=tuple(Literal.of(v) for v in values))

    @staticmethod
    def MvccVisible(deleted_by_field: int, txn_id: int, snapshot_id: int, uncommitted: Sequence[int] = ()):
        """MvccRowIdFilter as a leaf on the `created_by` column (llkv-transaction/src/helpers.rs:259-312)."""
        return Operator(OP_MVCC_VISIBLE, Literal.of(deleted_by_field), Bound(BOUND_INCLUDED, txn_id), Bound(BOUND_INCLUDED, snapshot_id),
                        tuple(Literal.of(u) for u in uncommitted))

    IsNull: ClassVar["Operator"]
    IsNotNull: ClassVar["Operator"]


Operator.IsNull = Operator(OP_IS_NULL)
Operator.IsNotNull = Operator(OP_IS_NOT_NULL)


@dataclass(frozen=True)
class Filter:
    """llkv_expr::Filter { field_id, op }."""
    field_id: int
    op: Operator


class Expr:
    """llkv_expr::Expr over leaf filters: Pred / And / Or / Not / Literal(bool)."""

    def __init__(self, kind: str, children=(), filter: Optional[Filter] = None, value: bool = True):
        self.kind, self.children, self.filter, self.value = kind, tuple(children), filter, value

    @staticmethod
    def pred(f: Filter) -> "Expr":
        return Expr("pred", filter=f)

    @staticmethod
    def all_of(items: Sequence[Union[Filter, "Expr"]]) -> "Expr":
        return Expr("and", [i if isinstance(i, Expr) else Expr.pred(i) for i in items])

    @staticmethod
    def any_of(items: Sequence[Union[Filter, "Expr"]]) -> "Expr":
        return Expr("or", [i if isinstance(i, Expr) else Expr.pred(i) for i in items])

    @staticmethod
    def not_(e: Union[Filter, "Expr"]) -> "Expr":
        return Expr("not", [e if isinstance(e, Expr) else Expr.pred(e)])

    @staticmethod
    def literal(v: bool) -> "Expr":
        return Expr("lit", value=v)

    @staticmethod
    def compare(left, op: int, right) -> "Expr":
        """Expr::Compare { left, op, right } over scalar expressions (CMP_EQ … CMP_GT_EQ)."""
        return Expr.pred(Filter(0, Operator(OP_COMPARE, cmp=(_scalar(left), op, _scalar(right)))))

    @staticmethod
    def in_list(target, items, negated: bool = False) -> "Expr":
        """Expr::InList { expr, list, negated } over scalar expressions."""
        return Expr.pred(Filter(0, Operator(OP_IN_LIST, in_list=(_scalar(target), [_scalar(i) for i in items], bool(negated)))))

    @staticmethod
    def is_null(expr, negated: bool = False) -> "Expr":
        """Expr::IsNull { expr, negated } over a scalar expression."""
        return Expr.pred(Filter(0, Operator(OP_IS_NULL_EXPR, is_null_expr=(_scalar(expr), bool(negated)))))

    @staticmethod
    def true() -> "Expr":
        return Expr("lit", value=True)


def pred_expr(f: Filter) -> Expr:
    return Expr.pred(f)


def compile_predicate(expr: Optional[Expr]) -> Tuple[List[Filter], List[Tuple[int, int]]]:
    """ProgramCompiler::compile (llkv-compute/src/program.rs:280-297): Expr → postfix EvalOps."""
    filters: List[Filter] = []
    ops: List[Tuple[int, int]] = []
    if expr is None:
        return filters, ops

    def visit(e: Expr):
        if e.kind == "pred":
            filters.append(e.filter)
            ops.append((EVAL_PUSH_PREDICATE, len(filters) - 1))
        elif e.kind == "lit":
            ops.append((EVAL_PUSH_LITERAL, 1 if e.value else 0))
        elif e.kind in ("and", "or"):
            if not e.children:
                raise LlkvError(INVALID_ARGUMENT, f"{e.kind.upper()} expression requires at least one predicate")
            for c in e.children:
                visit(c)
            ops.append((EVAL_AND if e.kind == "and" else EVAL_OR, len(e.children)))
        elif e.kind == "not":
            visit(e.children[0])
            ops.append((EVAL_NOT, 0))
        else:  # pragma: no cover
            raise ValueError(e.kind)

    visit(expr)
    return filters, ops


# --------------------------------------------------------------------- scalar exprs
class ScalarExpr:
    """llkv_expr::ScalarExpr restricted to Column / Literal / Binary, held in postfix form."""

    def __init__(self, tokens):
        self.tokens = list(tokens)

    @staticmethod
    def column(field_id: int) -> "ScalarExpr":
        return ScalarExpr([("col", field_id)])

    @staticmethod
    def literal(v) -> "ScalarExpr":
        return ScalarExpr([("lit", Literal.of(v))])

    @staticmethod
    def binary(left, op: int, right) -> "ScalarExpr":
        l, r = _scalar(left), _scalar(right)
        return ScalarExpr(l.tokens + r.tokens + [("bin", op)])

    def __add__(self, o):
        return ScalarExpr.binary(self, BIN_ADD, o)

    def __radd__(self, o):
        return ScalarExpr.binary(o, BIN_ADD, self)

    def __sub__(self, o):
        return ScalarExpr.binary(self, BIN_SUB, o)

    def __rsub__(self, o):
        return ScalarExpr.binary(o, BIN_SUB, self)

    def __mul__(self, o):
        return ScalarExpr.binary(self, BIN_MUL, o)

    def __rmul__(self, o):
        return ScalarExpr.binary(o, BIN_MUL, self)

    def __truediv__(self, o):
        return ScalarExpr.binary(self, BIN_DIV, o)

    def __mod__(self, o):
        return ScalarExpr.binary(self, BIN_MOD, o)

    def to_c(self, keep: list):
        arr = (CExprToken * len(self.tokens))()
        for i, t in enumerate(self.tokens):
            if t[0] == "col":
                arr[i].kind, arr[i].field_id = TOK_COLUMN, t[1]
            elif t[0] == "lit":
                arr[i].kind = TOK_LITERAL
                arr[i].literal = t[1].to_c(keep)
            else:
                arr[i].kind, arr[i].binop = TOK_BINARY, t[1]
        keep.append(arr)
        return arr


def _scalar(v) -> ScalarExpr:
    return v if isinstance(v, ScalarExpr) else ScalarExpr.literal(v)


def col(field_id: int) -> ScalarExpr:
    return ScalarExpr.column(field_id)


# ------------------------------------------------------------------------ aggregates
@dataclass
class AggregateSpec:
    """llkv_aggregate::AggregateSpec { alias, kind } with the argument as an expression."""
    kind: int
    expr: Optional[ScalarExpr] = None
    alias: str = ""
    distinct: bool = False

    @staticmethod
    def count_star(alias="count"):
        return AggregateSpec(AGG_COUNT_STAR, None, alias)

    @staticmethod
    def count(e, alias="count"):
        return AggregateSpec(AGG_COUNT, _colexpr(e), alias)

    @staticmethod
    def sum(e, alias="sum"):
        return AggregateSpec(AGG_SUM, _colexpr(e), alias)

    @staticmethod
    def total(e, alias="total"):
        return AggregateSpec(AGG_TOTAL, _colexpr(e), alias)

    @staticmethod
    def avg(e, alias="avg"):
        return AggregateSpec(AGG_AVG, _colexpr(e), alias)

    @staticmethod
    def min(e, alias="min"):
        return AggregateSpec(AGG_MIN, _colexpr(e), alias)

    @staticmethod
    def max(e, alias="max"):
        return AggregateSpec(AGG_MAX, _colexpr(e), alias)

    @staticmethod
    def count_nulls(e, alias="count_nulls"):
        return AggregateSpec(AGG_COUNT_NULLS, _colexpr(e), alias)


def _colexpr(e) -> ScalarExpr:
    return ScalarExpr.column(e) if isinstance(e, int) else e


@dataclass
class Value:
    """One finalized aggregate cell (the 1-element Arrow array of finalize())."""
    dtype: int
    is_null: bool
    value: object
    precision: int = 0  # Decimal128(precision, scale); value = raw i128
    scale: int = 0

    @staticmethod
    def from_c(c: CValue) -> "Value":
        if c.dtype == DT_DECIMAL128:
            return Value(c.dtype, bool(c.is_null), None if c.is_null else i128_from_words(c.i64, c.i64_hi), c.precision, c.scale)
        if c.is_null:
            return Value(c.dtype, True, None)
        if c.dtype == DT_FLOAT64:
            return Value(c.dtype, False, c.f64)
        if c.dtype == DT_UTF8:
            return Value(c.dtype, False, c.str.decode() if c.str is not None else "")
        return Value(c.dtype, False, c.i64)


# ------------------------------------------------------------------- plan marshalling
class CPlan:
    """C arrays for (filters, program, keys, aggregates); keeps every buffer alive."""

    def __init__(self, predicate: Optional[Union[Expr, Sequence[Filter]]], aggs: Sequence[AggregateSpec] = (),
                 keys: Sequence[int] = ()):
        self.keep: list = []
        if predicate is None:
            filters, ops = [], []
        elif isinstance(predicate, Expr):
            filters, ops = compile_predicate(predicate)
        else:  # plain list of filters = Expr::all_of with an empty program
            filters, ops = list(predicate), []
        self.n_filters, self.n_ops = len(filters), len(ops)
        self.filters = (CFilter * max(1, len(filters)))()
        for i, f in enumerate(filters):
            cf = self.filters[i]
            cf.field_id, cf.op = f.field_id, f.op.kind
            if f.op.value is not None:
                cf.value = f.op.value.to_c(self.keep)
            if f.op.kind in (OP_RANGE, OP_MVCC_VISIBLE):
                cf.lower_kind = f.op.lower.kind
                if f.op.lower.value is not None:
                    cf.lower = f.op.lower.value.to_c(self.keep)
                cf.upper_kind = f.op.upper.kind
                if f.op.upper.value is not None:
                    cf.upper = f.op.upper.value.to_c(self.keep)
            if f.op.kind == OP_COMPARE:
                l, op, r = f.op.cmp
                la, ra = l.to_c(self.keep), r.to_c(self.keep)
                cf.cmp_op, cf.cmp_left, cf.cmp_left_len, cf.cmp_right, cf.cmp_right_len = op, la, len(l.tokens), ra, len(r.tokens)
            if f.op.kind in (OP_STARTS_WITH, OP_ENDS_WITH, OP_CONTAINS):
                cf.case_sensitive = int(f.op.case_sensitive)
            if f.op.kind == OP_IN_LIST:
                tgt, items, neg = f.op.in_list
                arrs = [it.to_c(self.keep) for it in items]
                ptrs = (C.POINTER(CExprToken) * max(1, len(arrs)))(*[C.cast(a, C.POINTER(CExprToken)) for a in arrs])
                lens = (C.c_uint32 * max(1, len(arrs)))(*[len(it.tokens) for it in items])
                self.keep += [ptrs, lens]
                cf.cmp_left, cf.cmp_left_len = tgt.to_c(self.keep), len(tgt.tokens)
                cf.list_exprs, cf.list_expr_lens, cf.list_len, cf.negated = ptrs, lens, len(arrs), int(neg)
            if f.op.kind == OP_IS_NULL_EXPR:
                ex, neg = f.op.is_null_expr
                cf.cmp_left, cf.cmp_left_len, cf.negated = ex.to_c(self.keep), len(ex.tokens), int(neg)
            if f.op.kind in (OP_IN, OP_MVCC_VISIBLE):
                lst = (CLiteral * max(1, len(f.op.values)))()
                for j, v in enumerate(f.op.values):
                    lst[j] = v.to_c(self.keep)
                self.keep.append(lst)
                cf.in_list, cf.in_len = lst, len(f.op.values)
        self.ops = (CEvalOp * max(1, len(ops)))()
        for i, (op, arg) in enumerate(ops):
            self.ops[i].op, self.ops[i].arg = op, arg
        self.n_aggs = len(aggs)
        self.aggs = (CAggregateSpec * max(1, len(aggs)))()
        for i, a in enumerate(aggs):
            ca = self.aggs[i]
            ca.kind, ca.distinct = a.kind, int(a.distinct)
            if a.expr is not None:
                arr = a.expr.to_c(self.keep)
                ca.expr, ca.expr_len = arr, len(a.expr.tokens)
            alias = a.alias.encode()
            self.keep.append(alias)
            ca.alias = alias
        self.n_keys = len(keys)
        self.keys = (C.c_uint32 * max(1, len(keys)))(*keys)
