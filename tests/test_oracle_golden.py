"""CPU: pins the oracle against every known-answer test the reference holds for the path
(tests/golden/, transcribed by value — SURVEY.md §8c)."""
import numpy as np
import pytest

from conftest import build_aggs, build_filter, build_predicate, build_expr, build_string_operator, golden, oracle_table, same_value, DTYPES

TABLE = golden("table_scan.json")
JOINS = golden("joins.json")
AGGS = golden("aggregates.json")
STRINGS = golden("string_predicates.json")
STRING_SCANS = golden("string_scans.json")
JOIN_FILTERS = golden("join_filters.json")


@pytest.mark.parametrize("case", TABLE["cases"], ids=lambda c: c["name"])
def test_table_scan_cases(case, orc, abi):
    tdef = TABLE["tables"][case["table"]]
    t = oracle_table(orc, abi, tdef["columns"], tdef["rows"])
    pred = build_predicate(abi, case["predicate"])
    projections = [p if isinstance(p, int) else build_expr(abi, p) for p in case["project"]]
    if "expect_error" in case:
        with pytest.raises(abi.LlkvError) as e:
            orc.scan_stream(t, projections, pred)
        assert e.value.kind == case["expect_error"]
        return
    batches = orc.scan_stream(t, projections, pred, include_nulls=case.get("include_nulls", False))
    cols = [[] for _ in projections]
    for bcols, _ in batches:
        assert len(bcols[0]) > 0, "empty batches are never emitted (llkv-scan/src/execute.rs:289-291)"
        for i, c in enumerate(bcols):
            cols[i].extend(c)
    if "expect_sorted" in case:  # the reference scans this column in value order; this path in row order: the same set
        assert sorted(cols[0]) == case["expect_sorted"]
        return
    assert cols == case["expect"]
    if "expect_sum" in case:
        assert sum(v for v in cols[0] if v is not None) == case["expect_sum"]
    if "expect_min" in case:
        assert min(cols[0]) == case["expect_min"] and max(cols[0]) == case["expect_max"]
    if "expect_sqrt" in case:  # what the reference's consumer does with the batch: cast to f64, sqrt
        assert [float(np.sqrt(np.float64(v))) for v in cols[0]] == case["expect_sqrt"]


def _join_tables(orc, abi, case):
    def mk(rows):
        t = orc.OracleTable(len(rows))
        t.add(1, abi.DT_INT32, np.array([r[0] for r in rows], dtype=np.int32))
        t.add(2, abi.DT_UTF8, [r[1] for r in rows])
        return t
    return mk(case["left"]), mk(case["right"])


@pytest.mark.parametrize("case", JOINS["cases"], ids=lambda c: c["name"])
def test_join_cases(case, orc, abi):
    left, right = _join_tables(orc, abi, case)
    jt = {"inner": abi.JOIN_INNER, "left": abi.JOIN_LEFT, "semi": abi.JOIN_SEMI, "anti": abi.JOIN_ANTI}[case["type"]]
    if "expect_error" in case:
        with pytest.raises(abi.LlkvError) as e:
            orc.hash_join(left, right, [] if case.get("cross") else [(1, 1)], jt, case.get("batch_size", 8192))
        assert e.value.kind == case["expect_error"]
        return
    batches = orc.hash_join(left, right, [] if case.get("cross") else [(1, 1)], jt, case.get("batch_size", 8192))
    ls = [x for b in batches for x in b[0]]
    assert len(ls) == case["expect_rows"]
    if "expect_pairs" in case:
        rs = [x for b in batches for x in b[1]]
        pairs = [[l, None if r == 2**64 - 1 else r] for l, r in zip(ls, rs)]
        assert pairs == case["expect_pairs"]
    if "expect_left" in case:
        assert ls == case["expect_left"]
        assert all(b[1] is None for b in batches), "semi/anti joins deliver left columns only"


JOIN_COLS = [(1, "user_id"), (2, "name")]  # create_test_table join_tests.rs:18-53


@pytest.mark.parametrize("case", JOINS["cases"], ids=lambda c: c["name"])
def test_join_record_batches(case, orc, abi):
    """The same reference tests through the delivery the reference really has — joined RecordBatches: column count
    (`expect_columns`), output names (left, right, `_1`), and every cell = the cell of the paired source rows."""
    left, right = _join_tables(orc, abi, case)
    jt = {"inner": abi.JOIN_INNER, "left": abi.JOIN_LEFT, "semi": abi.JOIN_SEMI, "anti": abi.JOIN_ANTI}[case["type"]]
    keys = [] if case.get("cross") else [(1, 1)]
    if "expect_error" in case:
        with pytest.raises(abi.LlkvError) as e:
            orc.hash_join_batches(left, right, keys, JOIN_COLS, JOIN_COLS, jt, case.get("batch_size", 8192))
        assert e.value.kind == case["expect_error"]
        return
    batches = orc.hash_join_batches(left, right, keys, JOIN_COLS, JOIN_COLS, jt, case.get("batch_size", 8192))
    check_join_batches(case, batches)


def check_join_batches(case, batches):
    assert sum(len(cols[0]) for _, cols in batches) == case["expect_rows"]
    assert all(len(cols[0]) > 0 for _, cols in batches)
    left_only = case["type"] in ("semi", "anti")
    for names, cols in batches:
        assert len(names) == len(cols) == (2 if left_only else 4)
        if "expect_columns" in case:
            assert len(cols) == case["expect_columns"]
        assert names == (["user_id", "name"] if left_only else ["user_id", "name", "user_id_1", "name_1"])
        if "expect_names" in case:
            assert names == case["expect_names"]
    rows = [list(r) for _, cols in batches for r in zip(*cols)]
    if "expect_pairs" in case:
        want = [case["left"][l] + (case["right"][r] if r is not None else [None, None]) for l, r in case["expect_pairs"]]
        assert rows == want
    if "expect_left" in case:
        assert rows == [case["left"][l] for l in case["expect_left"]]


def test_join_record_batches_expression_filters(orc, abi):
    """join_tests.rs:470-558: the reference reads columns 0, 2, 5 and 6 of the joined batches — customer_id, annual_revenue
    of the left table, customer_id, avg_order_value of the right one — and counts the rows by the two filters."""
    c = JOIN_FILTERS["expression_filters"]
    left, right = oracle_table(orc, abi, c["left"]["columns"]), oracle_table(orc, abi, c["right"]["columns"])
    lcols = [(col["field_id"], nm) for col, nm in zip(c["left"]["columns"], ["customer_id", "segment", "annual_revenue", "loyalty_score"])]
    rcols = [(col["field_id"], nm) for col, nm in zip(c["right"]["columns"], ["order_id", "customer_id", "avg_order_value", "trailing_spend"])]
    batches = orc.hash_join_batches(left, right, [tuple(k) for k in c["join_keys"]], lcols, rcols, abi.JOIN_INNER)
    check_expression_filter_batches(c, batches)


def check_expression_filter_batches(c, batches):
    assert sum(len(cols[0]) for _, cols in batches) == c["expect_join_rows"]
    counts, both = {"both": 0, "left": 0, "right": 0, "neither": 0}, set()
    for names, cols in batches:
        assert names == ["customer_id", "segment", "annual_revenue", "loyalty_score", "order_id", "customer_id_1", "avg_order_value", "trailing_spend"]
        for cid_l, rev, cid_r, avg in zip(cols[0], cols[2], cols[5], cols[6]):
            assert cid_l == cid_r
            lp, rp = rev >= 900, cid_r in {1002, 1003, 1005, 1010, 1011}  # customers with an order of avg value >= 120
            counts["both" if lp and rp else "left" if lp else "right" if rp else "neither"] += 1
            if lp and rp:
                both.add(cid_l)
                assert rev >= 900 and avg >= 120
    assert (counts["both"], counts["left"], counts["right"], counts["neither"]) == (c["expect_both"], c["expect_left_only"], c["expect_right_only"], c["expect_neither"])
    assert sorted(both) == c["expect_both_customers"]


def test_generic_join_key_rules(orc, abi):
    """The generic typed-key path (llkv-join/src/hash_join.rs:62-148,377-505; any key list that is not one
    fast integer pair), expectations derived by hand from its rules: all parts equal; NULL equals nothing, or —
    under null_equals_null — the marker Utf8("<NULL>"), which also equals a real string "<NULL>"; values of two
    different types never match; a VALUE of a type extract_key_value does not list (Date32) drops the row."""
    def pairs(batches):
        return [(l, None if r == 2**64 - 1 else r) for b in batches for l, r in zip(b[0], b[1])]

    # two-part key (Int64, Int32): NULLs in either part
    left = orc.OracleTable(5).add(1, abi.DT_INT64, np.array([1, 1, 0, 2, 1]), [True, True, False, True, True]) \
                             .add(2, abi.DT_INT32, np.array([1, 2, 1, 0, 1], dtype=np.int32), [True, True, True, False, True])
    right = orc.OracleTable(5).add(1, abi.DT_INT64, np.array([1, 1, 2, 0, 1]), [True, True, True, False, True]) \
                              .add(2, abi.DT_INT32, np.array([1, 1, 0, 1, 2], dtype=np.int32), [True, True, False, True, True])
    assert pairs(orc.hash_join(left, right, [(1, 1), (2, 2)], abi.JOIN_INNER)) == [(0, 0), (0, 1), (1, 4), (4, 0), (4, 1)]
    # NULL = NULL on the first part only: (NULL,1) meets (NULL,1); (2,NULL) still matches nothing
    assert pairs(orc.hash_join(left, right, [(1, 1, True), (2, 2)], abi.JOIN_INNER)) == [(0, 0), (0, 1), (1, 4), (2, 3), (4, 0), (4, 1)]
    assert pairs(orc.hash_join(left, right, [(1, 1, True), (2, 2, True)], abi.JOIN_LEFT)) == [(0, 0), (0, 1), (1, 4), (2, 3), (3, 2), (4, 0), (4, 1)]
    assert [x for b in orc.hash_join(left, right, [(1, 1), (2, 2)], abi.JOIN_ANTI) for x in b[0]] == [2, 3]
    # Utf8 key: by string; the NULL marker is the string "<NULL>"
    ls = orc.OracleTable(4).add(1, abi.DT_UTF8, ["a", None, "<NULL>", "b"])
    rs = orc.OracleTable(4).add(1, abi.DT_UTF8, ["b", "<NULL>", None, "c"])
    assert pairs(orc.hash_join(ls, rs, [(1, 1)], abi.JOIN_INNER)) == [(2, 1), (3, 0)]
    assert pairs(orc.hash_join(ls, rs, [(1, 1, True)], abi.JOIN_INNER)) == [(1, 1), (1, 2), (2, 1), (2, 2), (3, 0)]
    # Int32 against Int64: never equal — but NULLs meet under null_equals_null, and so does a string "<NULL>"
    l32 = orc.OracleTable(3).add(1, abi.DT_INT32, np.array([7, 0, 8], dtype=np.int32), [True, False, True])
    r64 = orc.OracleTable(3).add(1, abi.DT_INT64, np.array([7, 8, 0]), [True, True, False])
    assert pairs(orc.hash_join(l32, r64, [(1, 1)], abi.JOIN_INNER)) == []
    assert pairs(orc.hash_join(l32, r64, [(1, 1, True)], abi.JOIN_LEFT)) == [(0, None), (1, 2), (2, None)]
    assert pairs(orc.hash_join(ls, r64, [(1, 1, True)], abi.JOIN_INNER)) == [(1, 2), (2, 2)]
    # Date32 values fail the key extraction (row skipped); Float64 keys compare by bit pattern (0.0 != -0.0, NaN == NaN)
    ld = orc.OracleTable(3).add(1, abi.DT_DATE32, np.array([10, 0, 11], dtype=np.int32), [True, False, True])
    rd = orc.OracleTable(3).add(1, abi.DT_DATE32, np.array([10, 11, 0], dtype=np.int32), [True, True, False])
    assert pairs(orc.hash_join(ld, rd, [(1, 1)], abi.JOIN_INNER)) == []
    assert pairs(orc.hash_join(ld, rd, [(1, 1, True)], abi.JOIN_INNER)) == [(1, 2)]
    lf = orc.OracleTable(4).add(1, abi.DT_FLOAT64, np.array([0.0, -0.0, np.nan, 1.5]))
    rf = orc.OracleTable(4).add(1, abi.DT_FLOAT64, np.array([1.5, np.nan, 0.0, 1.5]))
    assert pairs(orc.hash_join(lf, rf, [(1, 1)], abi.JOIN_INNER)) == [(0, 2), (2, 1), (3, 0), (3, 3)]
    # batches: the generic path cuts the probe into slices of batch_size rows (hash_join.rs:228-246)
    n = 10
    lb = orc.OracleTable(n).add(1, abi.DT_FLOAT64, np.arange(n, dtype=np.float64))
    rb = orc.OracleTable(n).add(1, abi.DT_FLOAT64, np.arange(n, dtype=np.float64) * 2)
    assert [b[0] for b in orc.hash_join(lb, rb, [(1, 1)], abi.JOIN_INNER, 4)] == [[0, 2], [4, 6], [8]]
    li = orc.OracleTable(n).add(1, abi.DT_INT64, np.arange(n))
    ri = orc.OracleTable(n).add(1, abi.DT_INT64, np.arange(n) * 2)
    assert [b[0] for b in orc.hash_join(li, ri, [(1, 1)], abi.JOIN_INNER, 4)] == [[0, 2, 4, 6], [8]]  # fast path: no slices


def test_executor_join_key_rules(orc, abi):
    """The executor's SQL joins (hash_join_table_batches / normalize_join_column / build_join_match_indices,
    llkv-executor/src/lib.rs:12218-12581), expectations derived by hand: integer types meet after the cast to
    Int64 (a UInt64 ≥ 2^63 turns NULL), Float32 meets Float64, Date32 only Date32; NULL parts never match; one
    batch; LEFT pads unmatched and NULL-key rows; other join types are refused."""
    def pairs(batches):
        assert len(batches) <= 1
        return [(l, None if r == 2**64 - 1 else r) for b in batches for l, r in zip(b[0], b[1])]
    X = abi.JOIN_KEYS_EXECUTOR
    l32 = orc.OracleTable(4).add(1, abi.DT_INT32, np.array([7, -1, 8, 9], dtype=np.int32), [True, True, True, False])
    r64 = orc.OracleTable(4).add(1, abi.DT_INT64, np.array([8, 7, -1, 7]))
    assert pairs(orc.hash_join(l32, r64, [(1, 1)], abi.JOIN_INNER, key_rules=X)) == [(0, 1), (0, 3), (1, 2), (2, 0)]
    assert pairs(orc.hash_join(l32, r64, [(1, 1)], abi.JOIN_LEFT, key_rules=X)) == [(0, 1), (0, 3), (1, 2), (2, 0), (3, None)]
    assert pairs(orc.hash_join(l32, r64, [(1, 1)], abi.JOIN_INNER)) == []  # llkv-join: Int32 and Int64 never equal
    ru = orc.OracleTable(3).add(1, abi.DT_UINT64, np.array([7, 2**64 - 1, 2**63], dtype=np.uint64))
    assert pairs(orc.hash_join(l32, ru, [(1, 1)], abi.JOIN_INNER, key_rules=X)) == [(0, 0)]  # 2^64-1 is NULL after the cast, not -1
    lf = orc.OracleTable(3).add(1, abi.DT_FLOAT32, np.array([0.5, 0.1, -0.0], dtype=np.float32))
    rf = orc.OracleTable(4).add(1, abi.DT_FLOAT64, np.array([0.5, 0.1, float(np.float32(0.1)), 0.0]))
    assert pairs(orc.hash_join(lf, rf, [(1, 1)], abi.JOIN_INNER, key_rules=X)) == [(0, 0), (1, 2)]
    ld = orc.OracleTable(2).add(1, abi.DT_DATE32, np.array([10, 11], dtype=np.int32))
    rd = orc.OracleTable(2).add(1, abi.DT_DATE32, np.array([11, 10], dtype=np.int32))
    ri = orc.OracleTable(2).add(1, abi.DT_INT64, np.array([11, 10]))
    assert pairs(orc.hash_join(ld, rd, [(1, 1)], abi.JOIN_INNER, key_rules=X)) == [(0, 1), (1, 0)]
    assert pairs(orc.hash_join(ld, ri, [(1, 1)], abi.JOIN_INNER, key_rules=X)) == []
    ls = orc.OracleTable(3).add(1, abi.DT_UTF8, ["a", None, "b"]).add(2, abi.DT_INT64, np.array([1, 2, 3]))
    rs = orc.OracleTable(3).add(1, abi.DT_UTF8, ["b", "a", "a"]).add(2, abi.DT_INT32, np.array([3, 1, 9], dtype=np.int32))
    assert pairs(orc.hash_join(ls, rs, [(1, 1), (2, 2)], abi.JOIN_INNER, key_rules=X)) == [(0, 1), (2, 0)]
    for jt in (abi.JOIN_SEMI, abi.JOIN_ANTI, abi.JOIN_RIGHT):
        with pytest.raises(abi.LlkvError) as e:
            orc.hash_join(ls, rs, [(1, 1)], jt, key_rules=X)
        assert e.value.kind in ("Internal", "InvalidArgumentError")


@pytest.mark.parametrize("case", AGGS["cases"], ids=lambda c: c["name"])
def test_aggregate_cases(case, orc, abi):
    t = oracle_table(orc, abi, case["columns"])
    pred = [build_filter(abi, case["filter"])] if "filter" in case else None
    aggs = build_aggs(abi, case["aggs"])
    if "group_by" in case:  # one group expected: its values are the expectation
        rows = orc.groupby(t, pred, case["group_by"], aggs)
        assert len(rows) == 1
        for g, w in zip(rows[0].values, case["expect"]):
            assert same_value(g.value, w), (g, w)
        return
    if "expect_error" in case:
        with pytest.raises(abi.LlkvError) as e:
            orc.aggregate(t, pred, aggs)
        assert e.value.kind == case["expect_error"]
        return
    got = orc.aggregate(t, pred, aggs)
    for g, w in zip(got, case["expect"]):
        assert same_value(g.value, w), (g, w)


def test_group_by_first_appearance_and_order(orc, abi):
    """GROUP BY emits groups in first-appearance order (llkv-executor/src/lib.rs:5065-5089); ORDER BY sorts them;
    Date32 keys collapse to Int (group_by_handles_date32_columns, :13964-13976); NULL is its own group."""
    t = orc.OracleTable(6)
    t.add(1, abi.DT_DATE32, np.array([3, 0, -7, 3, -7, 0], dtype=np.int32), [True, False, True, True, True, False])
    t.add(2, abi.DT_INT64, np.array([10, 20, 30, 40, 50, 60], dtype=np.int64))
    A = abi.AggregateSpec
    rows = orc.groupby(t, None, [1], [A.sum(2), A.count_star()])
    assert [(r.keys[0].value, r.values[0].value, r.values[1].value) for r in rows] == [(3, 50, 2), (None, 80, 2), (-7, 80, 2)]
    rows = orc.groupby(t, None, [1], [A.sum(2)], order_by_keys=True)
    assert [r.keys[0].value for r in rows] == [None, -7, 3]


def test_group_by_int_expression_goes_through_f64(orc, abi):
    """Int∘Int inside a GROUP BY aggregate argument is computed in f64 and cast back
    (llkv-executor/src/lib.rs:7338-7389): exactness is lost above 2^53 — reproduced, not fixed."""
    big = 2**53 + 1
    t = orc.OracleTable(2)
    t.add(1, abi.DT_UTF8, ["g", "g"])
    t.add(2, abi.DT_INT64, np.array([big, 1], dtype=np.int64))
    A = abi.AggregateSpec
    rows = orc.groupby(t, None, [1], [A.sum(abi.col(2) + 0)])
    assert rows[0].values[0].value == int(float(big)) + 1  # 2^53 + 1 rounds to 2^53 through f64
    assert orc.aggregate(t, None, [A.sum(abi.col(2) + 0)])[0].value == big + 1  # fast path stays exact (checked i64)


def test_aggregate_expression_div_by_zero_is_null(orc, abi):
    """Scan projections: x / 0 → NULL (zeros of the divisor are nullified first, llkv-compute/src/kernels.rs:121-135)
    but x % 0 is arrow's `rem` → "Divide by zero" (:136-138, fast_numeric.rs:328-334).  GROUP BY arguments go through
    the PlanValue interpreter, where both are NULL (llkv-executor/src/lib.rs:7193-7389, tests :14020-14048)."""
    t = orc.OracleTable(3)
    t.add(1, abi.DT_INT64, np.array([10, 20, 7], dtype=np.int64))
    t.add(2, abi.DT_INT64, np.array([0, 5, 2], dtype=np.int64))
    t.add(3, abi.DT_UTF8, ["g", "g", "g"])
    A, col = abi.AggregateSpec, abi.col
    got = orc.aggregate(t, None, [A.sum(col(1) / col(2)), A.count(col(1) / col(2)), A.sum(col(1) / 4.0)])
    assert [g.value for g in got] == [7, 2, 9.25]  # 20/5 + 7/2 (truncating); 10/0 is NULL
    with pytest.raises(abi.LlkvError) as e:
        orc.aggregate(t, None, [A.count(col(1) % col(2))])
    assert e.value.kind == "Internal" and "Divide by zero" in e.value.message
    got = orc.aggregate(t, [abi.Filter(2, abi.Operator.GreaterThan(0))], [A.sum(col(1) % col(2))])
    assert got[0].value == 1
    rows = orc.groupby(t, None, [3], [A.count(col(1) % col(2)), A.sum(col(1) % col(2)), A.sum(col(1) / 4.0)])
    assert [v.value for v in rows[0].values] == [2, 1, 9.25]


def test_constant_subexpressions_fold_before_the_projection_is_typed(orc, abi):
    """ScalarEvaluator::simplify (llkv-compute/src/eval.rs:761-791, applied by llkv-scan/src/execute.rs:91): literal ⊕ literal
    is one compute_binary call over one-element arrays — integers checked and truncating, x / 0 the NULL literal, floats
    IEEE — and only then does the projection pick its route: `col * (1 / 4)` has no Divide left (fast path, times 0).
    Expected values derived by hand from those rules."""
    t = orc.OracleTable(3)
    t.add(1, abi.DT_INT64, np.array([10, -3, 7], dtype=np.int64))
    t.add(2, abi.DT_FLOAT64, np.array([0.5, -2.0, 4.0]))
    L, col = abi.ScalarExpr.literal, abi.col
    projs = [col(1) * (L(2) + 3), col(1) + (L(7) / 2), col(1) * (L(1) / 4), col(2) * (L(1) / 4.0), (L(10) % 4) + col(1),
             (L(2) * 3 + 1) * col(2) - (L(1.5) - 0.25), (L(6) / 3) / col(1), col(1) + (L(-7) / 2), col(1) - (L(-7) % 3)]
    (cols, _), = orc.scan_stream(t, projs, None)
    assert cols == [[50, -15, 35], [13, 0, 10], [0, 0, 0], [0.125, -0.5, 1.0], [12, -1, 9],
                    [2.25, -15.25, 26.75], [0, 0, 0], [7, -6, 4], [11, -2, 8]]
    assert all(isinstance(v, int) for c in (cols[0], cols[1], cols[2], cols[4], cols[6]) for v in c)
    # a fold that errors stays unfolded in the reference (fold_binary_literals → None): neither side takes such plans
    for bad in (col(1) + (L(2**62) + 2**62), col(1) + (L(5) % 0), col(1) * (L(-2**63) / -1)):
        with pytest.raises(abi.LlkvError) as e:
            orc.scan_stream(t, [bad], None)
        assert e.value.kind == "Unsupported"


def test_a_negative_zero_divisor_is_not_the_zero_that_becomes_null(orc, abi):
    """compute_binary's Divide nullifies the rows where `eq(rhs, cast(0))` (llkv-compute/src/kernels.rs:121-135); arrow-ord's
    `eq` (57.x, Cargo.lock) compares floats by totalOrder, so only +0.0 is nullified and x / −0.0 is IEEE: ∓inf, NaN for 0."""
    t = orc.OracleTable(4)
    t.add(1, abi.DT_FLOAT64, np.array([1.0, 1.0, -1.0, 0.0]))
    t.add(2, abi.DT_FLOAT64, np.array([0.0, -0.0, -0.0, -0.0]))
    (cols, _), = orc.scan_stream(t, [abi.col(1) / abi.col(2)], None, include_nulls=True)
    assert cols[0][0] is None and cols[0][1] == float("-inf") and cols[0][2] == float("inf") and np.isnan(cols[0][3])
    t.add(3, abi.DT_UTF8, ["g"] * 4)
    rows = orc.groupby(t, None, [3], [abi.AggregateSpec.count(abi.col(1) / abi.col(2))])  # the PlanValue interpreter compares with ==: NULL for both zeros
    assert [r.values[0].value for r in rows] == [0]


def test_distinct_accumulators_over_string_boolean_date_and_decimal_keys(orc, abi):
    """DistinctKey (llkv-aggregate/src/lib.rs:252-331): Str by value — "1" and "1.0" are two keys although both parse to 1.0 —,
    Bool, Date and the raw Decimal by value; SUM / TOTAL / AVG over a non-float key add each NEW key's numeric image in order
    of first appearance (:889-924: strings parse or count 0, booleans 1 / 0, dates their day number), over Decimal128 they
    run in i128 and AVG rounds half away from zero (:1762-1800).  Expected values derived by hand from those rules."""
    t = orc.OracleTable(7)
    t.add(1, abi.DT_UTF8, ["1", "1.0", "x", "1", " 2.5 ", None, "x"])
    t.add(2, abi.DT_BOOLEAN, np.array([1, 0, 1, 1, 0, 0, 1], dtype=np.uint8), [True, True, True, True, True, False, True])
    t.add(3, abi.DT_DATE32, np.array([10, 10, -3, 7, 7, 7, 10], dtype=np.int32))
    t.add(4, abi.DT_DECIMAL128, [150, 150, -25, 1, 1, 150, 2], precision=10, scale=2)
    A = abi.AggregateSpec
    D = lambda k, f: A(k, abi._colexpr(f), "d", True)
    got = orc.aggregate(t, None, [D(abi.AGG_COUNT, 1), D(abi.AGG_SUM, 1), D(abi.AGG_TOTAL, 1), D(abi.AGG_AVG, 1),
                                  D(abi.AGG_COUNT, 2), D(abi.AGG_SUM, 2), D(abi.AGG_COUNT, 3), D(abi.AGG_SUM, 3), D(abi.AGG_AVG, 3),
                                  D(abi.AGG_COUNT, 4), D(abi.AGG_SUM, 4), D(abi.AGG_TOTAL, 4), D(abi.AGG_AVG, 4)])
    vals = [None if g.is_null else g.value for g in got]
    assert vals[:9] == [4, 4.5, 4.5, 4.5 / 4, 2, 1.0, 3, 14.0, 14.0 / 3]
    assert [(g.dtype, g.precision, g.scale) for g in got[10:]] == [(abi.DT_DECIMAL128, 10, 2)] * 3
    assert vals[9:] == [4, 128, 128, 32]  # 150 − 25 + 1 + 2 = 128 (raw, scale 2); 128 / 4 = 32 exactly
    none = orc.aggregate(t, [abi.Filter(3, abi.Operator.GreaterThan(100))], [D(abi.AGG_COUNT, 1), D(abi.AGG_SUM, 1), D(abi.AGG_TOTAL, 1), D(abi.AGG_SUM, 4), D(abi.AGG_TOTAL, 4), D(abi.AGG_AVG, 4)])
    assert [bool(g.is_null) for g in none] == [False, True, False, True, False, True]
    assert none[0].value == 0 and none[2].value == 0.0 and none[4].value == 0
    t2 = orc.OracleTable(3)
    t2.add(1, abi.DT_DECIMAL128, [1, 2, 4], precision=5, scale=1)
    assert orc.aggregate(t2, None, [D(abi.AGG_AVG, 1)])[0].value == 2   # 7 / 3 = 2.33 → 2
    t3 = orc.OracleTable(2)
    t3.add(1, abi.DT_DECIMAL128, [-1, -2], precision=5, scale=1)
    assert orc.aggregate(t3, None, [D(abi.AGG_AVG, 1)])[0].value == -2  # −3 / 2 = −1.5 → −2 (half away from zero)


def test_q6_against_numpy(orc, abi, tpch):
    """Independent cross-check of the restatement (pyarrow/numpy are NOT the reference)."""
    n = tpch.LINEITEM_ROWS["sf0.01"]
    d = tpch.gen_lineitem(n, 0.01)
    t = orc.OracleTable(n)
    for name, (fid, dt) in tpch.LINEITEM_SCHEMA.items():
        t.add(fid, dt, d[name])
    q = tpch.q6()
    got = orc.aggregate(t, q.predicate, q.aggs)[0].value
    m = (d["l_shipdate"] >= 8766) & (d["l_shipdate"] < 9131) & (d["l_discount"] >= 0.05) & (d["l_discount"] <= 0.07) & (d["l_quantity"] < 24)
    seq = 0.0
    for v in (d["l_extendedprice"][m] * d["l_discount"][m]):
        seq += v
    assert got == seq  # strict left-to-right order, bit-exact
    par = orc.aggregate_parallel(t, q.predicate, q.aggs, 4)[0].value
    assert abs(par - got) <= 1e-9 * abs(got)
    q1 = tpch.q1()
    rows = orc.groupby(t, q1.predicate, q1.keys, q1.aggs, True)
    assert [tuple(k.value for k in r.keys) for r in rows] == [("A", "F"), ("N", "F"), ("N", "O"), ("R", "F")]
    for r in rows:
        sel = (d["l_returnflag"] == ord(r.keys[0].value)) & (d["l_linestatus"] == ord(r.keys[1].value)) & (d["l_shipdate"] <= 10471)
        assert r.values[0].value == int(d["l_quantity"][sel].sum())
        assert r.values[7].value == int(sel.sum())
        assert abs(r.values[1].value - d["l_extendedprice"][sel].sum()) <= 1e-9 * r.values[1].value
    # the chunk-parallel fused mode (the "best-effort parallel" CPU baseline) against the reference-faithful one
    par = orc.groupby_parallel(t, q1.predicate, q1.keys, q1.aggs, 4)
    assert [tuple(k) for k, _ in par] == [tuple(k.value for k in r.keys) for r in rows]
    for (_, pv), r in zip(par, rows):
        for a, b in zip(pv, r.values):
            assert a.value == b.value if isinstance(b.value, int) else abs(a.value - b.value) <= 1e-9 * abs(b.value)


MVCC = golden("mvcc.json")


def mvcc_case_rows(m, abi, case):
    """One (created_by, deleted_by) row and the snapshot of one assertion of test_row_visibility_simple."""
    t = m.OracleTable(1) if hasattr(m, "OracleTable") else None
    F, O = abi.Filter, abi.Operator
    pred = [F(1, O.MvccVisible(2, txn_id=case["txn_id"], snapshot_id=case["snapshot_id"], uncommitted=case["uncommitted"]))]
    return np.array([case["created_by"]], dtype=np.uint64), np.array([case["deleted_by"]], dtype=np.uint64), pred


@pytest.mark.parametrize("case", MVCC["cases"], ids=lambda c: c["name"])
def test_mvcc_reference_visibility_sequence(case, orc, abi):
    """llkv-transaction/src/mvcc.rs:528-556 (test_row_visibility_simple): the five is_visible_for assertions with the
    transaction ids, snapshots and Active sets the manager has at each of them."""
    created, deleted, pred = mvcc_case_rows(orc, abi, case)
    t = orc.OracleTable(1).add(1, abi.DT_UINT64, created).add(2, abi.DT_UINT64, deleted)
    assert (orc.filter_row_ids(t, pred).tolist() == [0]) == case["expect"]


def mvcc_version_columns(versions):
    created = np.concatenate([np.full(v["rows"], v["created_by"], dtype=np.uint64) for v in versions])
    deleted = np.concatenate([np.full(v["rows"], v["deleted_by"], dtype=np.uint64) for v in versions])
    return created, deleted


@pytest.mark.parametrize("case", MVCC["count_cases"], ids=lambda c: c["name"])
def test_mvcc_count_star_with_transaction_local_changes(case, orc, abi):
    """llkv-slt-tester/tests/slt/duckdb/transactions/count_star_transactions.slt: COUNT(*) under the MVCC row filter while
    another connection's deletes / appends are uncommitted, and after they commit (the answers are the reference's)."""
    created, deleted = mvcc_version_columns(case["versions"])
    t = orc.OracleTable(len(created)).add(1, abi.DT_UINT64, created).add(2, abi.DT_UINT64, deleted)
    vis = abi.Filter(1, abi.Operator.MvccVisible(2, txn_id=case["txn_id"], snapshot_id=case["snapshot_id"], uncommitted=case["uncommitted"]))
    assert orc.aggregate(t, [vis], [abi.AggregateSpec.count_star()])[0].value == case["expect"]


def test_mvcc_reference_basic_visibility():
    """mvcc.rs:535,540-541: RowVersion::is_visible — numeric ordering only (:276-279); held by the same reference test."""
    NONE = 2**64 - 1
    for c in MVCC["basic"]:
        assert (c["created_by"] <= c["snapshot_txn_id"] and (c["deleted_by"] == NONE or c["deleted_by"] > c["snapshot_txn_id"])) == c["expect"]


def test_mvcc_visibility_rules(orc, abi):
    """RowVersion::is_visible_for (llkv-transaction/src/mvcc.rs:283-333) case by case:
    (created_by, deleted_by) under snapshot {txn_id 7, snapshot_id 5}, txn 4 Active (not committed)."""
    NONE = 2**64 - 1
    rows = [
        (1, NONE),   # auto-commit creator, never deleted                     → visible
        (3, NONE),   # committed creator ≤ snapshot                           → visible
        (6, NONE),   # creator committed AFTER the snapshot (6 > 5)           → invisible
        (4, NONE),   # creator not committed                                  → invisible
        (7, NONE),   # created by the current transaction                     → visible
        (7, 7),      # created and deleted by the current transaction         → invisible
        (3, 7),      # deleted by the current transaction                     → invisible
        (3, 4),      # deleter not committed                                  → still visible
        (3, 5),      # deleter committed, ≤ snapshot                          → invisible
        (3, 6),      # deleter committed after the snapshot                   → visible
        (NONE, NONE),  # creator status None                                  → invisible
        (1, 1),      # deleted by an auto-commit statement (1 ≤ 5)            → invisible
    ]
    t = orc.OracleTable(len(rows))
    t.add(1, abi.DT_UINT64, np.array([r[0] for r in rows], dtype=np.uint64)).add(2, abi.DT_UINT64, np.array([r[1] for r in rows], dtype=np.uint64))
    F, O = abi.Filter, abi.Operator
    got = orc.filter_row_ids(t, [F(1, O.MvccVisible(2, txn_id=7, snapshot_id=5, uncommitted=[4]))])
    assert got.tolist() == [0, 1, 4, 7, 9]
    # auto-commit snapshot (txn_id = 1) is never "the current transaction" (mvcc.rs:296)
    got = orc.filter_row_ids(t, [F(1, O.MvccVisible(2, txn_id=1, snapshot_id=9, uncommitted=[4]))])
    assert got.tolist() == [0, 1, 2, 4, 7]


def test_sum_int64_overflow_is_order_dependent(orc, abi):
    """checked_add chain (llkv-aggregate/src/lib.rs:816-829): a prefix that overflows is an error even when the total fits."""
    A = abi.AggregateSpec
    big = 2**62
    t = orc.OracleTable(4).add(1, abi.DT_INT64, np.array([big, big, -big, -big], dtype=np.int64))
    with pytest.raises(abi.LlkvError) as e:
        orc.aggregate(t, None, [A.sum(1)])
    assert e.value.kind == "InvalidArgumentError"
    t = orc.OracleTable(4).add(1, abi.DT_INT64, np.array([big, -big, big, -big], dtype=np.int64))
    assert orc.aggregate(t, None, [A.sum(1)])[0].value == 0


def test_compare_uses_total_order_and_common_types(orc, abi):
    """Expr::Compare → arrow-ord cmp kernels (llkv-compute/src/kernels.rs:269-297): floats by IEEE totalOrder
    (NaN above +inf, -0.0 below +0.0), unlike the leaf predicate's partial_cmp; NOT is taken over the rows
    where both sides are determined (llkv-scan/src/predicate.rs:779-818)."""
    E, col = abi.Expr, abi.col
    t = orc.OracleTable(6)
    t.add(1, abi.DT_FLOAT64, np.array([1.0, float("nan"), -0.0, 0.0, float("inf"), -1.0]))
    t.add(2, abi.DT_FLOAT64, np.array([0.0, float("inf"), 0.0, -0.0, float("nan"), -1.0]))
    t.add(3, abi.DT_INT64, np.array([2**53 + 1, -1, 0, 5, 7, -1], dtype=np.int64))
    t.add(4, abi.DT_UINT64, np.array([2**53, 2**64 - 1, 0, 5, 6, 1], dtype=np.uint64))
    ids = lambda e: list(orc.filter_row_ids(t, e))
    assert ids(E.compare(col(1), abi.CMP_GT, col(2))) == [0, 1, 3]      # NaN > inf, 0.0 > -0.0
    assert ids(E.compare(col(1), abi.CMP_EQ, col(2))) == [5]            # -0.0 ≠ 0.0 under totalOrder
    assert ids(E.not_(E.compare(col(1), abi.CMP_GT, col(2)))) == [2, 4, 5]
    # the leaf route (column ⋈ literal) keeps partial_cmp: NaN matches nothing, -0.0 == 0.0
    assert ids(E.compare(col(1), abi.CMP_GT, 0.5)) == [0, 4]
    assert ids(E.compare(col(1), abi.CMP_EQ, 0.0)) == [2, 3]
    # but <> is evaluated row-wise with totalOrder: only +0.0 equals the literal
    assert ids(E.compare(col(1), abi.CMP_NOT_EQ, 0.0)) == [0, 1, 2, 4, 5]
    # Int64 ⋈ UInt64 → Float64: 2^53 + 1 rounds onto 2^53, -1 < 2^64
    assert ids(E.compare(col(3), abi.CMP_EQ, col(4))) == [0, 2, 3]
    assert ids(E.compare(col(3), abi.CMP_LT, col(4))) == [1, 5]
    # checked arithmetic on a side is an error for the whole scan, whatever the other conjuncts select
    with pytest.raises(abi.LlkvError) as e:
        ids(E.all_of([E.compare(col(3) * 2**12, abi.CMP_GT, col(3)), abi.Filter(3, abi.Operator.Equals(5))]))
    assert e.value.kind == "Internal" and "overflow" in e.value.message.lower()


def test_in_list_and_is_null_over_expressions(orc, abi):
    """Expr::InList (llkv-scan/src/predicate.rs:443-560) and Expr::IsNull over scalar expressions (:249-331):
    domains, item-by-item coercion of the target, totalOrder equality, the union-of-fields quirk of IS NULL."""
    E, col = abi.Expr, abi.col
    t = orc.OracleTable(6)
    t.add(1, abi.DT_INT64, np.array([1, 2, 3, 4, 5, 6], dtype=np.int64), [True, True, False, True, True, False])
    t.add(2, abi.DT_FLOAT64, np.array([1.0, float("nan"), 3.0, -0.0, 5.5, 0.0]), [True, True, True, True, False, False])
    ids = lambda e: list(orc.filter_row_ids(t, e))
    assert ids(E.in_list(col(1), [1, 4, 9])) == [0, 3]
    assert ids(E.in_list(col(1), [1, 4, 9], negated=True)) == [1, 4]          # NULL targets are outside the domain
    assert ids(E.not_(E.in_list(col(1), [1, 4, 9]))) == [1, 4]
    assert ids(E.in_list(col(1), [], negated=True)) == [0, 1, 3, 4]            # empty list: false, negated true
    assert ids(E.in_list(col(1) + 0, [col(2), 4])) == [0, 3]                   # Int64 target coerced to Float64 by the first item
    assert ids(E.in_list(col(2), [float("nan"), 0.0])) == [1]                  # NaN = NaN, -0.0 ≠ 0.0 (row 5 has NULLs)
    # IS NULL over an expression: rows where at least one field is present and the value is NULL
    assert ids(E.is_null(col(1) + col(2))) == [2, 4]                           # row 5 (both absent) is never scanned
    assert ids(E.is_null(col(1) + col(2), negated=True)) == [0, 1, 3]
    assert ids(E.not_(E.is_null(col(1) + col(2)))) == [0, 1, 3]                # NOT: within the rows where every field is present
    assert ids(E.is_null(col(1))) == [2, 5] and ids(E.is_null(col(1), negated=True)) == [0, 1, 3, 4]  # bare column: the leaf
    assert ids(E.is_null(col(2) / (col(1) - 1))) == [0, 2, 4]                  # x / 0 is NULL


@pytest.mark.parametrize("case", STRINGS["cases"], ids=lambda c: c["name"])
def test_string_predicate_cases(case, orc, abi):
    """The reference's string predicate known answers (typed_predicate.rs:538-600) through the oracle's leaf filter."""
    t = orc.OracleTable(len(case["values"])).add(1, abi.DT_UTF8, case["values"])
    ids = set(orc.filter_row_ids(t, [abi.Filter(1, build_string_operator(abi, case["op"]))]).tolist())
    assert [i in ids for i in range(len(case["values"]))] == case["expect"]


def _string_scan_values(case):
    if "generate" in case:
        g = case["generate"]
        return [f"row-{i}-payload-needle" if i % g["needle_every"] == 0 else f"row-{i}-payload" for i in range(g["rows"])]
    return case["values"]


@pytest.mark.parametrize("case", STRING_SCANS["cases"], ids=lambda c: c["name"])
def test_string_scan_cases(case, orc, abi):
    """llkv-table/tests/fusion_tests.rs and table.rs:1725-1771: string predicates (and their same-field AND, which the
    reference fuses) through scan_stream."""
    values = _string_scan_values(case)
    t = orc.OracleTable(len(values)).add(1, abi.DT_UTF8, values)
    filters = [abi.Filter(1, build_string_operator(abi, op)) for op in case["ops"]]
    got = [v for cols, _ in orc.scan_stream(t, [1], filters) for v in cols[0]]
    if "expect_in_order" in case:
        assert got == case["expect_in_order"]
    if "expect_sorted" in case:
        assert sorted(got) == case["expect_sorted"]
    if case.get("property") == "fused_count_equals_intersection":
        ids = [set(orc.filter_row_ids(t, [f]).tolist()) for f in filters]
        assert len(got) == len(set.intersection(*ids)) and len(got) > 0


def test_join_with_expression_filters(orc, abi):
    """llkv-join/tests/join_tests.rs:299-561: two Expr::Compare-filtered scans and an inner join on Int32 keys; the
    reference asserts the filtered id sets, 8 joined rows and how they split over the two filters."""
    c = JOIN_FILTERS["expression_filters"]
    left, right = oracle_table(orc, abi, c["left"]["columns"]), oracle_table(orc, abi, c["right"]["columns"])
    lids = [v for cols, _ in orc.scan_stream(left, [c["left_filter_project"]], build_predicate(abi, c["left_filter"])) for v in cols[0]]
    rids = [v for cols, _ in orc.scan_stream(right, [c["right_filter_project"]], build_predicate(abi, c["right_filter"])) for v in cols[0]]
    assert sorted(set(lids)) == c["expect_left_ids"]
    assert len(set(rids)) == c["expect_right_id_count"] and set(c["expect_right_ids_include"]) <= set(rids)
    batches = orc.hash_join(left, right, [tuple(k) for k in c["join_keys"]], abi.JOIN_INNER)
    pairs = [(l, r) for b in batches for l, r in zip(b[0], b[1])]
    assert len(pairs) == c["expect_join_rows"]
    lk, rk = c["left"]["columns"][0]["values"], c["right"]["columns"][1]["values"]
    counts, both = {"both": 0, "left": 0, "right": 0, "neither": 0}, set()
    for l, r in pairs:
        assert lk[l] == rk[r]
        lp, rp = lk[l] in set(lids), rk[r] in set(rids)
        counts["both" if lp and rp else "left" if lp else "right" if rp else "neither"] += 1
        if lp and rp:
            both.add(lk[l])
    assert (counts["both"], counts["left"], counts["right"], counts["neither"]) == (c["expect_both"], c["expect_left_only"], c["expect_right_only"], c["expect_neither"])
    assert sorted(both) == c["expect_both_customers"]


@pytest.mark.parametrize("case", JOIN_FILTERS["cartesian"], ids=lambda c: c["name"])
def test_cartesian_known_answers(case, orc, abi):
    """join_tests.rs:711-806 and the executor's three-table product (llkv-executor/src/lib.rs:14065-14150)."""
    if "tables" in case:  # chained products of single-column tables: left-major row order
        cols = [[v] for v in case["tables"][0]]
        for nxt in case["tables"][1:]:
            lt = orc.OracleTable(len(cols)).add(1, abi.DT_INT64, np.arange(len(cols)))
            rt_ = orc.OracleTable(len(nxt)).add(1, abi.DT_INT64, np.array(nxt, dtype=np.int64))
            pairs = [(l, r) for b in orc.hash_join(lt, rt_, [], abi.JOIN_INNER) for l, r in zip(b[0], b[1])]
            cols = [cols[l] + [nxt[r]] for l, r in pairs]
        assert len(cols) == case["expect_rows"]
        assert [[row[i] for row in cols] for i in range(len(case["tables"]))] == case["expect_columns"]
        return
    left, right = _join_tables(orc, abi, case)
    pairs = [(l, r) for b in orc.hash_join(left, right, [], abi.JOIN_INNER) for l, r in zip(b[0], b[1])]
    assert len(pairs) == case["expect_rows"]
    if "expect_combinations" in case:
        got = {(case["left"][l][0], case["left"][l][1], case["right"][r][0], case["right"][r][1]) for l, r in pairs}
        assert got == {tuple(x) for x in case["expect_combinations"]}


def test_golden_inventory_says_what_pins_the_oracle():
    """How much of tests/golden/ is asserted by tests the reference itself holds (the pin), and how much was derived
    by reading the cited source lines (documentation of the restatement, no pin)."""
    held = derived = 0
    for doc in (TABLE, JOINS, AGGS):
        for c in doc["cases"]:
            if c.get("held_by_reference") is True or c.get("held_by_reference") == "partial":
                held += 1
            else:
                derived += 1
    held += len(STRINGS["cases"]) + len(STRING_SCANS["cases"]) + 1 + len(JOIN_FILTERS["cartesian"]) + len(MVCC["cases"]) + len(MVCC["count_cases"])
    print(f"golden cases held by the reference's own tests: {held}; derived from source lines: {derived}")
    assert held >= 99 and derived <= 10


def test_oracle_runs_distinct_accumulators_per_group(orc, abi):
    """GROUP BY hands every group's rows, in scan order, to the same accumulators an ungrouped query uses
    (llkv-executor/src/lib.rs:5100-5247) — the DISTINCT forms included: Int64 by value, Float64 by bit pattern, NULL cells
    skipped, f64 sums in order of first appearance.  Expectations derived by hand from those rules (not reference-held)."""
    import dataclasses
    t = orc.OracleTable(8).add(1, abi.DT_UTF8, ["a", "b", "a", "a", "b", "b", "a", "b"])
    t.add(2, abi.DT_INT64, np.array([1, 1, 2, 1, 5, 5, 3, 7]), [1, 1, 1, 1, 1, 0, 1, 1]).add(3, abi.DT_FLOAT64, np.array([.5, .5, .5, 1.5, 2., 2., .5, 4.]))
    A = abi.AggregateSpec
    D = lambda a: dataclasses.replace(a, distinct=True)
    res = orc.groupby(t, None, [1], [A.count_star(), D(A.count(2)), D(A.sum(2)), D(A.avg(2)), D(A.total(3)), D(A.sum(3)), D(A.max(2))], True)
    assert [[k.value for k in r.keys] + [v.value for v in r.values] for r in res] == [["a", 4, 3, 6, 2.0, 2.0, 2.0, 3], ["b", 4, 3, 13, 13 / 3, 6.5, 6.5, 7]]


def test_decimal_arithmetic_in_group_by_arguments_follows_the_planvalue_rules(orc, abi):
    """GROUP BY aggregate arguments with a Decimal operand: exact decimal arithmetic (llkv-executor/src/lib.rs:7229-7330 over
    llkv-compute/src/scalar/decimal.rs:128-234).  Hand-derived from those lines (held_by_reference: false — the reference has no
    test of this arm): the known quirks by value, then 300 random shapes against the Python-integer model of tests/decimal_model.py."""
    import random
    import decimal_model as dm
    A, S = abi.AggregateSpec, abi.ScalarExpr
    ops = {"+": abi.BIN_ADD, "-": abi.BIN_SUB, "*": abi.BIN_MUL, "/": abi.BIN_DIV}

    def run(c1, s1, c2, s2, ints, expr):
        n = len(c1)
        t = orc.OracleTable(n)
        t.add(1, abi.DT_DECIMAL128, c1, precision=38, scale=s1).add(2, abi.DT_DECIMAL128, c2, precision=38, scale=s2)
        t.add(3, abi.DT_INT64, np.array(ints, dtype=np.int64)).add(4, abi.DT_INT64, np.zeros(n, dtype=np.int64))
        return orc.groupby(t, None, [4], [A.sum(expr), A.min(expr), A.max(expr), A.avg(expr), A.count(expr)], True)[0].values

    # Q1's arguments under DECIMAL(15,2): price * (1 - discount) at scale 4, * (1 + tax) at scale 6; precision = digits of the first value
    v = run([10000, 20050], 2, [5, 10], 2, [0, 0], abi.col(1) * (1 - abi.col(2)))
    assert (v[0].value, v[0].precision, v[0].scale) == (10000 * 95 + 20050 * 90, 6, 4) and v[4].value == 2
    # 1.00 / 3 = 0.34 (half of an odd denominator is truncated), -0.01 / 3 = +0.01 (a truncated quotient of 0 counts as positive)
    v = run([100, -1], 2, [0, 0], 0, [3, 3], abi.col(1) / abi.col(3))
    assert (v[1].value, v[2].value, v[0].scale) == (1, 34, 2)
    # x / 0 is NULL, and a group whose every value is NULL has an Int64 temp column (SUM comes back as an Int64 NULL)
    v = run([100, 7], 2, [0, 0], 1, [0, 0], abi.col(1) / abi.col(2))
    assert v[0].is_null and v[0].dtype == abi.DT_INT64 and v[4].value == 0
    # a first value with fewer digits than its scale: arrow refuses Decimal128(1, 4)
    with pytest.raises(abi.LlkvError) as e:
        run([10000, 20000], 2, [0, 5], 2, [0, 0], abi.col(1) * abi.col(2))
    assert e.value.kind == "InvalidArgumentError" and "precision" in e.value.message
    # Decimal with a Float operand, Decimal % x: errors
    for expr in (abi.col(1) * 0.5, abi.col(1) % abi.col(3)):
        with pytest.raises(abi.LlkvError) as e:
            run([100], 2, [1], 0, [3], expr)
        assert e.value.kind == "InvalidArgumentError"

    random.seed(20240607)
    seen = set()
    for _ in range(300):
        n = random.randint(1, 12)
        s1, s2 = random.randint(0, 6), random.randint(0, 6)
        mag = random.choice([10, 10**6, 10**15, 10**19, 10**30, 10**37])
        c1 = [random.randint(-mag, mag) for _ in range(n)]
        c2 = [random.randint(-mag // 10**random.randint(0, 6) - 1, mag) for _ in range(n)]
        if random.random() < 0.3:
            c2[random.randrange(n)] = 0
        ints = [random.randint(-50, 50) for _ in range(n)]
        op = random.choice("+-*/")
        shape = random.choice(["dd", "di", "id", "dl", "dld"])
        lit = random.randint(-5, 5)
        left = {"dd": lambda i: (c1[i], s1), "di": lambda i: (c1[i], s1), "id": lambda i: (ints[i], 0), "dl": lambda i: (c1[i], s1), "dld": lambda i: (c1[i], s1)}[shape]
        right = {"dd": lambda i: (c2[i], s2), "di": lambda i: (ints[i], 0), "id": lambda i: (c2[i], s2), "dl": lambda i: (lit, 0), "dld": lambda i: (12345, 3)}[shape]
        le = {"dd": abi.col(1), "di": abi.col(1), "id": abi.col(3), "dl": abi.col(1), "dld": abi.col(1)}[shape]
        re_ = {"dd": abi.col(2), "di": abi.col(3), "id": abi.col(2), "dl": S.literal(lit), "dld": S.literal(abi.Literal.decimal(12345, 3))}[shape]
        try:
            vals = [dm.binary(left(i), right(i), op) for i in range(n)]
            typ = dm.temp_column(vals)
            if typ is None:
                want = ("all NULL",)
            else:
                tot = 0
                for x in vals:
                    if x is not None:
                        tot += x[0]
                        if not -(1 << 127) <= tot < (1 << 127):
                            raise dm.DecimalError("sum")
                if dm.digits(tot) > 38:
                    raise dm.DecimalError("final")
                nn = [x[0] for x in vals if x is not None]
                want = (tot, min(nn), max(nn), len(nn), typ[0], typ[1])
        except dm.DecimalError:
            want = ("error",)
        try:
            v = run(c1, s1, c2, s2, ints, S.binary(le, ops[op], re_))
            got = ("all NULL",) if v[0].is_null and v[0].dtype == abi.DT_INT64 else (v[0].value, v[1].value, v[2].value, v[4].value, v[0].precision, v[0].scale)
        except abi.LlkvError as ex:
            assert ex.kind == "InvalidArgumentError"
            got = ("error",)
        assert got == want, (shape, op, s1, s2, c1, c2, ints)
        seen.add((op, want[0] if isinstance(want[0], str) else "value"))
    assert {("*", "error"), ("/", "value"), ("/", "all NULL"), ("+", "value")} <= seen
