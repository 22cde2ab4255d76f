"""TPC-H qualification harness semantics — the caller side of the path.

Restates `llkv-tpch qualify` (llkv-tpch/src/qualification.rs): answer sets are `|`-separated with a
header line (:785-849), every column has a kind from the `colprecision` tokens (:920-931), rows are
compared order-insensitively (`diff_rows` :672-697) with exact equality for strings / integers /
decimals and an ABSOLUTE tolerance of 1e-9 for floats (`values_equal` :708-745, FLOAT_TOLERANCE :39);
the query's wall time is measured around the engine call (`main.rs:1306-1334`).
The official answer sets are not in the reference tree (llkv-tpch/.gitignore:1-3); when supplied they
can be checked with `qualify()`.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from decimal import Decimal, InvalidOperation
from typing import Callable, List, Optional, Sequence, Tuple

FLOAT_TOLERANCE = 1e-9

KIND_OF_TOKEN = {"str": "string", "int": "integer", "cnt": "integer", "sum": "decimal", "num": "decimal",
                 "avg": "float", "rat": "float"}


def kind_from_token(token: str) -> str:
    """ValueKind::from_token (qualification.rs:920-931)."""
    try:
        return KIND_OF_TOKEN[token]
    except KeyError:
        raise ValueError(f"unrecognized colprecision token '{token}'")


NULL = None


def parse_expected_value(text: str, kind: str):
    """parse_expected_value (qualification.rs:826-849): returns (tag, value)."""
    if text.upper() == "NULL":
        return ("null", None)
    if kind == "string":
        return ("string", text)
    if kind == "integer":
        try:
            return ("int", int(text))
        except ValueError as e:
            raise ValueError(f"unable to parse integer '{text}': {e}")
    if kind == "decimal":
        try:
            return ("decimal", Decimal(text).normalize())
        except InvalidOperation as e:
            raise ValueError(f"unable to parse decimal '{text}': {e}")
    try:
        return ("float", float(text))
    except ValueError as e:
        raise ValueError(f"unable to parse float '{text}': {e}")


def date32_to_string(days: int) -> str:
    """date32_to_string (qualification.rs:598-603)."""
    import datetime
    return (datetime.date(1970, 1, 1) + datetime.timedelta(days=int(days))).isoformat()


def engine_value(v, kind: str, is_date: bool = False):
    """extract_value (qualification.rs:342-353): the engine cell is read ACCORDING TO THE EXPECTED KIND —
    string kind formats (dates as YYYY-MM-DD, integers as text), integer kind → int, decimal kind → exact
    Decimal (Int64 exactly, Float64 through Decimal::from_f64), float kind → f64."""
    if v is None:
        return ("null", None)
    if kind == "string":
        if is_date:
            return ("string", date32_to_string(v))
        return ("string", v if isinstance(v, str) else str(v))
    if kind == "integer":
        return ("int", int(v))
    if kind == "decimal":
        if isinstance(v, Decimal):
            return ("decimal", v.normalize())
        if isinstance(v, int):
            return ("decimal", Decimal(v).normalize())
        return ("decimal", decimal_from_f64(float(v)))
    return ("float", float(v))


def decimal_from_f64(x: float) -> Decimal:
    """`Decimal::from_f64` of rust_decimal 1.39.0 (Cargo.lock:2635-2638; the dependency is not in the reference tree):
    the value is converted exactly and the "excess bits of precision" are removed down to the 15 significant decimal
    digits an f64 guarantees, rounding half up.  This is what extract_decimal (qualification.rs:532-540) applies to a
    Float64 result cell before a `sum`-kind column is compared EXACTLY with the answer set."""
    import math
    if not math.isfinite(x):
        raise ValueError("unable to convert float to decimal")
    d = Decimal(x)  # exact binary value
    if d == 0:
        return Decimal(0)
    from decimal import ROUND_HALF_UP, Context
    digits = d.adjusted()  # position of the leading digit
    q = Decimal(1).scaleb(digits - 14)  # keep 15 significant digits
    return d.quantize(q, rounding=ROUND_HALF_UP, context=Context(prec=60)).normalize()


def decimal_15_digits(d: Decimal) -> Decimal:
    """An exact decimal cut to the 15 significant digits Decimal::from_f64 keeps (half up): what an engine whose f64 result
    is the correctly rounded exact value presents to the `sum`-kind compare."""
    from decimal import ROUND_HALF_UP, Context
    if d == 0:
        return Decimal(0)
    return d.quantize(Decimal(1).scaleb(d.adjusted() - 14), rounding=ROUND_HALF_UP, context=Context(prec=60)).normalize()


def values_equal(expected, actual, kind: str) -> bool:
    """values_equal (qualification.rs:708-745)."""
    (te, ve), (ta, va) = expected, actual
    if te == "null" and ta == "null":
        return True
    if te == "null" or ta == "null":
        return False
    if te == ta == "string" or te == ta == "int" or te == ta == "decimal":
        return ve == va
    if te == ta == "float":
        return abs(ve - va) <= FLOAT_TOLERANCE
    if {te, ta} == {"decimal", "float"}:
        d = ve if te == "decimal" else va
        f = va if te == "decimal" else ve
        try:
            return abs(d - decimal_from_f64(f)) <= decimal_from_f64(FLOAT_TOLERANCE)
        except (InvalidOperation, ValueError):
            return False
    return False


def rows_equal(a: Sequence, b: Sequence, kinds: Sequence[str]) -> bool:
    return all(values_equal(x, y, k) for x, y, k in zip(a, b, kinds))


@dataclass
class RowDiff:
    missing: List
    extra: List

    @property
    def ok(self) -> bool:
        return not self.missing and not self.extra


def diff_rows(expected: Sequence[Sequence], actual: Sequence[Sequence], kinds: Sequence[str]) -> RowDiff:
    """diff_rows (qualification.rs:672-697): greedy, order-insensitive."""
    missing = [True] * len(expected)
    extra = []
    for row in actual:
        for i, exp in enumerate(expected):
            if missing[i] and rows_equal(exp, row, kinds):
                missing[i] = False
                break
        else:
            extra.append(row)
    return RowDiff([r for r, m in zip(expected, missing) if m], extra)


def parse_answer_set(text: str, kinds: Sequence[str]) -> List[List]:
    """load_answer_set / parse_answer_row (qualification.rs:785-824): header line skipped, '|' separated."""
    rows, header_skipped = [], False
    for line in text.splitlines():
        t = line.strip()
        if not t:
            continue
        if not header_skipped:
            header_skipped = True
            continue
        parts = [p.strip() for p in t.split("|")]
        if len(parts) != len(kinds):
            raise ValueError(f"answer row has {len(parts)} fields but query expects {len(kinds)}")
        rows.append([parse_expected_value(p, k) for p, k in zip(parts, kinds)])
    return rows


def qualify(run: Callable[[], Sequence[Sequence]], answer_text: str, tokens: Sequence[str]) -> Tuple[RowDiff, float]:
    """Runs the query, times it like the reference harness (wall clock around the engine call) and diffs the
    rows against the answer set."""
    kinds = [kind_from_token(t) for t in tokens]
    expected = parse_answer_set(answer_text, kinds)
    t0 = time.perf_counter()
    rows = run()
    elapsed = time.perf_counter() - t0
    actual = [[engine_value(v, k) for v, k in zip(r, kinds)] for r in rows]
    return diff_rows(expected, actual, kinds), elapsed


# TPC-H colprecision tokens of the queries on this path (answer-set column kinds)
Q1_TOKENS = ["str", "str", "sum", "sum", "sum", "sum", "avg", "avg", "avg", "cnt"]
Q6_TOKENS = ["sum"]
Q3_TOKENS = ["int", "sum", "str", "int"]


# ---------------------------------------------------------------------------------------------------------------
# Query rendering with the default substitution parameters (llkv-tpch/src/queries.rs:60-121,203-267).  The
# reference substitutes `:1`, `:2`, … in the TPC templates with the `defaults` table of the toolkit's varsub.c
# (queries.rs:384-470; the toolkit is not in the tree, llkv-tpch/.gitignore:1-3) unless the caller overrides them
# (QueryOptions.parameter_overrides), parses the SQL and hands it to the engine.  The SQL front end is not part of
# this path, so a "rendered query" here is the PLAN the executor would receive; the defaults below are the
# validation values of the TPC-H specification (clauses 2.4.1.3, 2.4.3.3, 2.4.6.3), which varsub.c's table holds
# for stream 0.
# ---------------------------------------------------------------------------------------------------------------
DEFAULT_PARAMETERS = {
    1: ["90"],                         # :1 DELTA — l_shipdate <= date '1998-12-01' - interval ':1' day
    3: ["BUILDING", "1995-03-15"],     # :1 SEGMENT, :2 DATE
    6: ["1994-01-01", "0.06", "24"],   # :1 DATE, :2 DISCOUNT, :3 QUANTITY
}


def _date32(text: str) -> int:
    import datetime
    return (datetime.date.fromisoformat(text) - datetime.date(1970, 1, 1)).days


def render_parameters(number: int, overrides: Optional[dict] = None) -> List[str]:
    """build_parameter_values (queries.rs:203-232): defaults first, then the caller's overrides by 1-based index;
    a placeholder without a value is an error."""
    if number not in DEFAULT_PARAMETERS:
        raise ValueError(f"TPC-H query {number} is not rendered on this path (Q1, Q3, Q6 are)")
    values = list(DEFAULT_PARAMETERS[number])
    for idx, v in (overrides or {}).items():
        if not 1 <= int(idx) <= len(values):
            raise ValueError(f"query {number}: no placeholder :{idx}")
        values[int(idx) - 1] = str(v)
    return values


def render_query(tpch, abi, number: int, overrides: Optional[dict] = None):
    """The plan of TPC-H Q1 / Q6 (a tpch.QueryPlan) or Q3 (the keyword arguments of runtime.join_groupby_topk, minus the
    tables) with the substitution parameters applied, literal bounds rendered the way SURVEY.md §8d describes (dates as
    Date32 day numbers, so the leaf-predicate route is taken)."""
    p = render_parameters(number, overrides)
    F, O, B, col = abi.Filter, abi.Operator, abi.Bound, abi.col
    if number == 1:
        q = tpch.q1()
        q.predicate = [F(tpch.L_SHIPDATE, O.LessThanOrEquals(_date32("1998-12-01") - int(p[0])))]
        return q
    if number == 6:
        q = tpch.q6()
        d0 = _date32(p[0])
        import datetime
        d1 = _date32(datetime.date.fromisoformat(p[0]).replace(year=datetime.date.fromisoformat(p[0]).year + 1).isoformat())
        # `between :2 - 0.01 and :2 + 0.01`: numeric SQL literals are decimals in the reference (llkv-sql/src/
        # sql_engine.rs:11026-11038), folded exactly, then cast for the Float64 column as raw / 10^scale
        # (llkv-types/src/literal.rs:487-492): 0.06 ∓ 0.01 are the f64 values 0.05 and 0.07
        lo_d, hi_d = Decimal(p[1]) - Decimal("0.01"), Decimal(p[1]) + Decimal("0.01")
        as_f64 = lambda x: float(int(x.scaleb(-x.as_tuple().exponent))) / 10.0 ** (-x.as_tuple().exponent) if x.as_tuple().exponent < 0 else float(x)
        q.predicate = [F(tpch.L_SHIPDATE, O.Range(B.Included(d0), B.Excluded(d1))),
                       F(tpch.L_DISCOUNT, O.Range(B.Included(as_f64(lo_d)), B.Included(as_f64(hi_d)))),
                       F(tpch.L_QUANTITY, O.LessThan(int(p[2])))]
        return q
    d = _date32(p[1])
    return dict(fact_filters=[F(tpch.L_SHIPDATE, O.GreaterThan(d))], fact_key=tpch.L_ORDERKEY, dim_filters=[F(tpch.O_ORDERDATE, O.LessThan(d))],
                dim_key=tpch.O_ORDERKEY, sum_expr=col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT)), payload_fields=[tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY],
                limit=10, dim_fk=tpch.O_CUSTKEY, dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals(p[0]))], dim2_key=tpch.C_CUSTKEY)


def format_answer_set(header: Sequence[str], rows: Sequence[Sequence], kinds: Sequence[str], is_date: Sequence[bool] = ()) -> str:
    """Rows in the `|`-separated layout of the TPC answer sets (one header line; qualification.rs:785-824 reads it
    back): strings verbatim, dates as YYYY-MM-DD, integers as such; a Float64 cell of a `sum` / `num` column is
    written with the 15 significant digits `Decimal::from_f64` keeps (what the reference would write for its own
    result), a float-kind cell with full precision."""
    out = ["|".join(header)]
    for r in rows:
        cells = []
        for i, (v, k) in enumerate(zip(r, kinds)):
            if v is None:
                cells.append("NULL")
            elif k == "string":
                cells.append(date32_to_string(v) if (i < len(is_date) and is_date[i]) else str(v))
            elif k == "integer" or isinstance(v, int):
                cells.append(str(int(v)))
            elif k == "decimal":
                cells.append(format(decimal_from_f64(float(v)), "f"))
            else:
                cells.append(repr(float(v)))
        out.append("|".join(cells))
    return "\n".join(out) + "\n"


def compare_report(expected_rows: Sequence[Sequence], actual_rows: Sequence[Sequence], tokens: Sequence[str]) -> dict:
    """Column by column, for two row lists in the same order: does the column pass the reference's rule for its kind
    (values_equal), and how far apart are the numeric cells (largest absolute and relative difference)."""
    kinds = [kind_from_token(t) for t in tokens]
    report = []
    for c, k in enumerate(kinds):
        ok, max_abs, max_rel = True, 0.0, 0.0
        for e, a in zip(expected_rows, actual_rows):
            ok &= values_equal(engine_value(e[c], k), engine_value(a[c], k), k)
            if isinstance(e[c], (int, float, Decimal)) and isinstance(a[c], (int, float, Decimal)) and not isinstance(e[c], bool):
                d = abs(float(e[c]) - float(a[c]))
                max_abs = max(max_abs, d)
                if e[c]:
                    max_rel = max(max_rel, d / abs(float(e[c])))
        report.append({"column": c, "token": tokens[c], "kind": k, "passes_reference_rule": bool(ok), "max_abs_diff": max_abs, "max_rel_diff": max_rel})
    return {"rows": len(expected_rows), "columns": report, "passes": all(r["passes_reference_rule"] for r in report)}
