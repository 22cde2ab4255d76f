import importlib
import json
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "rust-llkv_amd"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def mod(name: str):
    return importlib.import_module(f"{PKG}.{name}")


@pytest.fixture(scope="session")
def abi():
    return mod("abi")


@pytest.fixture(scope="session")
def tpch():
    return mod("tpch")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def rt():
    """The product binding, bound to cuda:0.  GPU tests only — fails loudly without a device."""
    r = mod("runtime")
    r.init(0)
    return r


def golden(name: str):
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return json.load(f)


DTYPES = {"Int64": 1, "Float64": 2, "Int32": 3, "Date32": 4, "UInt64": 5, "UInt32": 6, "Float32": 7, "Utf8": 8, "Decimal128": 10}


def fval(v):
    if v == "nan":
        return float("nan")
    if v == "-0.0":
        return -0.0
    return v


def build_predicate(abi, spec):
    """JSON predicate tree → abi.Expr."""
    if spec is None:
        return None
    if "pred" in spec:
        return abi.Expr.pred(build_filter(abi, spec["pred"]))
    if "and" in spec:
        return abi.Expr.all_of([build_predicate(abi, s) for s in spec["and"]])
    if "or" in spec:
        return abi.Expr.any_of([build_predicate(abi, s) for s in spec["or"]])
    if "not" in spec:
        return abi.Expr.not_(build_predicate(abi, spec["not"]))
    if "compare" in spec:
        c = spec["compare"]
        op = {"eq": abi.CMP_EQ, "ne": abi.CMP_NOT_EQ, "lt": abi.CMP_LT, "le": abi.CMP_LT_EQ, "gt": abi.CMP_GT, "ge": abi.CMP_GT_EQ}[c["op"]]
        return abi.Expr.compare(build_expr(abi, c["left"]), op, build_expr(abi, c["right"]))
    raise ValueError(spec)


def build_filter(abi, f):
    op = f["op"]
    O, B = abi.Operator, abi.Bound
    if op == "range":
        def bound(b):
            return B.Unbounded if b is None else (B.Included(b[1]) if b[0] == "included" else B.Excluded(b[1]))
        return abi.Filter(f["field"], O.Range(bound(f.get("lower")), bound(f.get("upper"))))
    if op == "in":
        return abi.Filter(f["field"], O.In(f["values"]))
    if op == "starts_with":
        return abi.Filter(f["field"], O.StartsWith(f["value"], f.get("case_sensitive", True)))
    ctor = {"eq": O.Equals, "gt": O.GreaterThan, "ge": O.GreaterThanOrEquals, "lt": O.LessThan, "le": O.LessThanOrEquals}[op]
    return abi.Filter(f["field"], ctor(f["value"]))


def build_string_operator(abi, op):
    """tests/golden/string_predicates.json operator → abi.Operator."""
    O, B = abi.Operator, abi.Bound
    k = op["kind"]
    if k == "eq":
        return O.Equals(op["value"])
    if k == "in":
        return O.In(op["values"])
    if k == "range":
        bound = lambda b: B.Unbounded if b is None else (B.Included(b[1]) if b[0] == "included" else B.Excluded(b[1]))
        return O.Range(bound(op.get("lower")), bound(op.get("upper")))
    return {"starts_with": O.StartsWith, "ends_with": O.EndsWith, "contains": O.Contains}[k](op["pattern"], op["case_sensitive"])


def build_expr(abi, e):
    if "col" in e:
        return abi.ScalarExpr.column(e["col"])
    if "lit" in e:
        return abi.ScalarExpr.literal(e["lit"])
    if "lit_date32" in e:
        return abi.ScalarExpr.literal(abi.Literal.date32(e["lit_date32"]))
    for name, op in (("add", abi.BIN_ADD), ("sub", abi.BIN_SUB), ("mul", abi.BIN_MUL), ("div", abi.BIN_DIV), ("mod", abi.BIN_MOD)):
        if name in e:
            return abi.ScalarExpr.binary(build_expr(abi, e[name][0]), op, build_expr(abi, e[name][1]))
    raise ValueError(e)


def build_aggs(abi, specs):
    A = abi.AggregateSpec
    out = []
    for s in specs:
        k = s["kind"]
        if k == "count_star":
            out.append(A.count_star())
        else:
            spec = getattr(A, k)(build_expr(abi, s["expr"]))
            spec.distinct = bool(s.get("distinct", False))
            out.append(spec)
    return out


def column_values(c):
    """The cells of a fixture column: `values` as written, or `values_blocks` expanded — {range: n} = 0 … n−1, {value: v} = v
    (None = a NULL cell), {blocks: […]} nested — each repeated `times` times, in order (the SLT tables of hundreds of thousands
    of rows, transcribed as the statements that built them)."""
    if "values" in c:
        return c["values"]

    def expand(blocks):
        out = []
        for b in blocks:
            one = list(range(b["range"])) if "range" in b else expand(b["blocks"]) if "blocks" in b else [b["value"]]
            out += one * b.get("times", 1)
        return out
    return expand(c["values_blocks"])


def oracle_table(orc, abi, columns, rows=None):
    rows = len(column_values(columns[0])) if rows is None else rows
    t = orc.OracleTable(rows)
    for c in columns:
        dt = DTYPES[c["dtype"]]
        vals = [fval(v) for v in column_values(c)]
        valid = None
        if any(v is None for v in vals):
            valid = [v is not None for v in vals]
            vals = [0 if v is None else v for v in vals]
        if dt == abi.DT_DECIMAL128:
            t.add(c["field_id"], dt, vals, valid, precision=c["precision"], scale=c["scale"])
        elif dt == abi.DT_UTF8:
            t.add(c["field_id"], dt, [None if (valid and not valid[i]) else vals[i] for i in range(len(vals))])
        else:
            t.add(c["field_id"], dt, np.array(vals, dtype=abi.NUMPY_OF_DTYPE[dt]), valid)
    return t


def same_value(got, want, rel=0.0):
    """Compare a finalized cell with an expectation (None = NULL, 'nan' = NaN)."""
    want = fval(want)
    if want is None:
        return got is None
    if got is None:
        return False
    if isinstance(want, float) and math.isnan(want):
        return isinstance(got, float) and math.isnan(got)
    if isinstance(want, float) or isinstance(got, float):
        if rel == 0.0 or math.isinf(float(want)) or math.isinf(float(got)):
            return float(got) == float(want)
        return abs(float(got) - float(want)) <= rel * max(abs(float(want)), 1e-300)
    return got == want
