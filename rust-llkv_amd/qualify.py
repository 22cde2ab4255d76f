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
        return ("decimal", Decimal(repr(float(v))).normalize())
    return ("float", float(v))


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
            return abs(d - Decimal(repr(f))) <= Decimal(repr(FLOAT_TOLERANCE))
        except InvalidOperation:
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
