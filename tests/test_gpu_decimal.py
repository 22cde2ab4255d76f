"""GPU (-m gpu): decimal arithmetic inside GROUP BY aggregate arguments — the PlanValue Decimal arm
(llkv-executor/src/lib.rs:7229-7330 over llkv-compute/src/scalar/decimal.rs:128-234) — through the C ABI against the oracle,
every cell bit-equal (raw i128, precision, scale, NULL-ness, dtype), and at SF10 against Python integers.

Under the reference's own TPC-H DDL (DECIMAL(15,2) money columns, llkv-tpch/src/lib.rs:154,1027-1091) Q1's three computed sums
take exactly this arm: `l_extendedprice * (1 - l_discount)` is a scale-4 product, `… * (1 + l_tax)` a scale-6 one."""
import numpy as np
import pytest

import decimal_model as dm

pytestmark = pytest.mark.gpu


def stage_lineitem(rt, orc, abi, tpch, d, columns, chunk_rows, with_oracle=True):
    n = sum(chunk_rows)
    ht = rt.HipTable(1, chunk_rows)
    ot = orc.OracleTable(n) if with_oracle else None
    for c in columns:
        fid, dt = tpch.LINEITEM_SCHEMA[c][0], tpch.lineitem_dtype(c, decimal=True)
        if dt == abi.DT_DECIMAL128:
            ht.append_decimal128_column(fid, tpch.DECIMAL_PRECISION, tpch.DECIMAL_SCALE, d[c])
            if ot:
                ot.add(fid, dt, d[c], precision=tpch.DECIMAL_PRECISION, scale=tpch.DECIMAL_SCALE)
        elif dt == abi.DT_UTF8:
            ht.append_utf8_column(fid, d[c])
            if ot:
                ot.add(fid, dt, d[c])
        else:
            ht.append_column(fid, dt, d[c])
            if ot:
                ot.add(fid, dt, d[c])
    return ht, ot


def same_groups(got, want, ctx=""):
    assert [[k.value for k in r.keys] for r in got] == [[k.value for k in r.keys] for r in want], ctx
    for g, w in zip(got, want):
        for i, (x, y) in enumerate(zip(g.values, w.values)):
            assert x == y, (ctx, [k.value for k in g.keys], i, x, y)  # dataclass equality: dtype, NULL-ness, raw value, precision, scale


@pytest.mark.parametrize("sf,chunk", [("sf0.01", 8192), ("sf0.01", 131072), ("sf1", 131072)])
def test_q1_decimal_columns_match_oracle(rt, orc, abi, tpch, sf, chunk):
    """TPC-H Q1 over DECIMAL(15,2) quantity / price / discount / tax: every cell of every group bit-equal to the oracle's —
    Decimal128(p, s) sums and averages out (p = digits of the group's first value for the computed arguments), counts, key order
    and first-appearance order."""
    n = tpch.LINEITEM_ROWS[sf]
    q = tpch.q1()
    d = tpch.lineitem_as_decimal(tpch.gen_lineitem(n, tpch.SCALE[sf], q.columns))
    ht, ot = stage_lineitem(rt, orc, abi, tpch, d, q.columns, tpch.chunk_rows(n, chunk))
    for ordered in (True, False):
        got = rt.groupby(ht, q.predicate, q.keys, q.aggs, ordered)
        want = orc.groupby(ot, q.predicate, q.keys, q.aggs, ordered)
        assert len(got) == 4
        same_groups(got, want, f"q1 decimal {sf} ordered={ordered}")
    # the sums are decimals: Decimal128(·, 2) for the bare columns, scale 4 / 6 for the products
    v = got[0].values
    assert [x.dtype for x in v[:7]] == [abi.DT_DECIMAL128] * 7 and [x.scale for x in v[:7]] == [2, 2, 4, 6, 2, 2, 2]
    assert (v[0].precision, v[1].precision) == (15, 15)
    # … and exactly the integers Python adds up
    flags = d["l_returnflag"].astype(np.int64) * 256 + d["l_linestatus"]
    sel = d["l_shipdate"] <= tpch.DATE_1998_09_02
    price, disc, tax = (d[c].astype(object) for c in ("l_extendedprice", "l_discount", "l_tax")) if n < 100000 else (None, None, None)
    if price is not None:
        for r in rt.groupby(ht, q.predicate, q.keys, q.aggs, True):
            m = sel & (flags == ord(r.keys[0].value) * 256 + ord(r.keys[1].value))
            assert r.values[2].value == int(np.sum(price[m] * (100 - disc[m])))
            assert r.values[3].value == int(np.sum(price[m] * (100 - disc[m]) * (100 + tax[m])))


def _py_q1_sums(d, sel):
    """Σ price·(100 − disc) and Σ price·(100 − disc)·(100 + tax) per (flag, status) in exact integer arithmetic: int64 products
    never leave 64 bits row by row (≤ 1.2e11), and partial sums of 2^20 rows stay below 2^63 before they meet as Python ints."""
    out = {}
    flags = d["l_returnflag"].astype(np.int64) * 256 + d["l_linestatus"]
    p, dc, tx = d["l_extendedprice"], d["l_discount"], d["l_tax"]
    disc_price = p * (100 - dc)
    charge = disc_price * (100 + tx)
    for f in np.unique(flags[sel]):
        m = np.flatnonzero(sel & (flags == f))
        tot = [0, 0, 0, 0, 0, 0]
        for lo in range(0, len(m), 1 << 20):
            i = m[lo:lo + (1 << 20)]
            for k, a in enumerate((d["l_quantity"], p, disc_price, charge, dc)):
                tot[k] += int(a[i].sum())
            tot[5] += len(i)
        out[(chr(int(f) >> 8), chr(int(f) & 255))] = tot
    return out


def test_q1_decimal_sf10_against_python_integers(rt, abi, tpch):
    """The full size of BASELINE.json's metric: Q1 over SF10 DECIMAL(15,2) columns on the GPU against exact integer arithmetic
    on the host (no oracle at this size): the four sums as raw decimals, the three averages rounded half away from zero, the counts."""
    n = tpch.LINEITEM_ROWS["sf10"]
    q = tpch.q1()
    d = tpch.lineitem_as_decimal(tpch.gen_lineitem(n, 10.0, q.columns))
    ht = rt.HipTable(1, tpch.chunk_rows(n))
    for c in q.columns:
        fid, dt = tpch.LINEITEM_SCHEMA[c][0], tpch.lineitem_dtype(c, decimal=True)
        if dt == abi.DT_DECIMAL128:
            ht.append_decimal128_column(fid, 15, 2, d[c])
        elif dt == abi.DT_UTF8:
            ht.append_utf8_column(fid, d[c])
        else:
            ht.append_column(fid, dt, d[c])
    got = rt.groupby(ht, q.predicate, q.keys, q.aggs, True)
    want = _py_q1_sums(d, d["l_shipdate"] <= tpch.DATE_1998_09_02)
    assert [(r.keys[0].value, r.keys[1].value) for r in got] == sorted(want)

    def avg(s, c):  # llkv-aggregate/src/lib.rs:1720-1760
        qv, rem = abs(s) // c, abs(s) % c
        return (qv + (1 if rem * 2 >= c else 0)) * (1 if s >= 0 else -1)
    for r in got:
        qty, base, disc_price, charge, disc, cnt = want[(r.keys[0].value, r.keys[1].value)]
        v = r.values
        assert [v[0].value, v[1].value, v[2].value, v[3].value, v[7].value] == [qty, base, disc_price, charge, cnt]
        assert [v[4].value, v[5].value, v[6].value] == [avg(qty, cnt), avg(base, cnt), avg(disc, cnt)]
        assert [x.scale for x in v[:7]] == [2, 2, 4, 6, 2, 2, 2]


ROUTES = [("dense", {}), ("image", {}), ("partitioned", {}), ("sort", {"LLKV_HIP_GROUP_NO_IMAGE": "1", "LLKV_HIP_GROUP_NO_PART": "1"})]


@pytest.mark.parametrize("route,env", ROUTES, ids=[r[0] for r in ROUTES])
@pytest.mark.parametrize("chunks", [[13], [4096, 4097, 3], [65536, 9000]])
def test_decimal_expressions_in_group_by_arguments_match_oracle(rt, orc, abi, chunks, route, env, monkeypatch):
    """Decimal ∘ Decimal, Decimal ∘ Int64 column, Decimal ∘ integer / decimal literal under + − × ÷ with mixed scales, NULL cells
    on both sides, zero divisors (→ NULL) and every aggregate kind, on each GROUP BY route (per-thread LDS columns, shared image,
    partitioned, sort-based): every cell equal to the oracle's."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(len(chunks) * 7 + len(route))
    n = sum(chunks)
    # magnitudes keep every group's first value at least as many digits as its scale (the case that fails the query has its own
    # test below, and `outcome` compares errors too)
    a = (rng.integers(1000, 10**7, size=n) * rng.choice([-1, 1], size=n)).astype(np.int64)   # DECIMAL(15,2)
    b = rng.integers(10**6, 10**8, size=n).astype(np.int64)         # DECIMAL(12,4), positive
    c = rng.integers(-3, 4, size=n).astype(np.int64)                # DECIMAL(5,1) with zeros: divisors
    i = rng.integers(-40, 40, size=n).astype(np.int64)              # Int64
    va, vc = rng.random(n) > 0.15, rng.random(n) > 0.1
    keyspace = {"dense": 5, "image": 900, "partitioned": 200_000, "sort": 700}[route]
    key = rng.integers(0, keyspace, size=n).astype(np.int64)
    if route == "sort":
        key = key * 1_000_003  # sparse keys: no dense group id
    ht = rt.HipTable(1, chunks)
    ot = orc.OracleTable(n)
    for fid, (p, s, vals, valid) in {1: (15, 2, a, va), 2: (12, 4, b, None), 3: (5, 1, c, vc)}.items():
        ht.append_decimal128_column(fid, p, s, vals, valid=valid)
        ot.add(fid, abi.DT_DECIMAL128, vals, None if valid is None else list(valid), precision=p, scale=s)
    ht.append_column(4, abi.DT_INT64, i)
    ht.append_column(5, abi.DT_INT64, key)
    ot.add(4, abi.DT_INT64, i).add(5, abi.DT_INT64, key)
    A, S, col, F, O = abi.AggregateSpec, abi.ScalarExpr, abi.col, abi.Filter, abi.Operator
    lit = S.literal(abi.Literal.decimal(1250, 3))  # 1.250
    exprs = [col(2) * (1 - col(2)), col(1) + col(2), col(1) - col(3) * col(4), col(2) * col(4) + 7, col(1) * lit, (col(2) + lit) * (col(2) - 2),
             col(1) / col(3), col(2) / col(4), (col(1) + col(3)) / col(3)]
    aggs = [A.count_star()]
    for e in exprs:
        aggs += [A.sum(e), A.avg(e), A.min(e), A.max(e), A.count(e)]
    aggs += [A.total(exprs[1]), A.count_nulls(exprs[6])]
    step = 6  # (five groups × a dozen lanes still fit the per-thread columns)

    def outcome(m, t, pred, part, ordered):
        try:
            return m.groupby(t, pred, [5], part, ordered)
        except abi.LlkvError as e:
            assert e.kind == "InvalidArgumentError", e
            return ("error", e.kind)
    values = errors = 0
    for pred in (None, [F(4, O.GreaterThan(-20))]):
        for lo in range(0, len(aggs), step):
            part = aggs[lo:lo + step]
            pq = rt.PreparedQuery(ht, pred, part, [5], False)
            note = pq.route_note
            pq.close()
            if n > 60000:  # (a handful of rows leaves a key range every route's smaller neighbour takes)
                want = {"dense": "GROUP BY with per-thread accumulator columns", "image": "shared-image", "partitioned": "partitioned", "sort": "sort-based"}[route]
                assert note.startswith(want), (route, note)
            for ordered in (False, True):
                got, exp = outcome(rt, ht, pred, part, ordered), outcome(orc, ot, pred, part, ordered)
                if isinstance(exp, tuple) or isinstance(got, tuple):
                    assert got == exp, (route, lo, ordered, got if isinstance(got, tuple) else "values", exp if isinstance(exp, tuple) else "values")
                    errors += 1
                else:
                    same_groups(got, exp, f"{route} {lo} ordered={ordered}")
                    values += 1
    assert values > errors


def test_first_value_of_a_group_types_its_decimal_temp_column(rt, orc, abi):
    """plan_values_to_arrow_array (llkv-executor/src/lib.rs:298-330): the temp column of a computed decimal argument is
    Decimal128(digits of the group's first non-NULL value, scale) — the precision of the finalized cell — and arrow refuses a
    positive scale above the precision: a group whose first product is below 10^(scale−1) fails the query, COUNT included; a
    group without any non-NULL value has an Int64 temp column."""
    price = np.array([10000, 20050, 99, 5, 123456, 777], dtype=np.int64)          # DECIMAL(15,2)
    disc = np.array([5, 10, 50, 1, 0, 2], dtype=np.int64)                          # DECIMAL(15,2)
    key = np.array([0, 0, 1, 1, 2, 2], dtype=np.int64)
    valid = np.array([1, 1, 1, 1, 0, 0], dtype=bool)
    ht, ot = rt.HipTable(1, [6]), orc.OracleTable(6)
    ht.append_decimal128_column(1, 15, 2, price)
    ht.append_decimal128_column(2, 15, 2, disc, valid=valid)
    ht.append_column(3, abi.DT_INT64, key)
    ot.add(1, abi.DT_DECIMAL128, price, precision=15, scale=2).add(2, abi.DT_DECIMAL128, disc, list(valid), precision=15, scale=2).add(3, abi.DT_INT64, key)
    A, col = abi.AggregateSpec, abi.col
    e = col(1) * col(2)
    aggs = [A.sum(e), A.avg(e), A.min(e), A.max(e), A.total(e), A.count(e), A.sum(col(1) * (1 - col(2)))]
    got, want = rt.groupby(ht, None, [3], aggs, True), orc.groupby(ot, None, [3], aggs, True)
    same_groups(got, want)
    assert (got[0].values[0].value, got[0].values[0].precision, got[0].values[0].scale) == (10000 * 5 + 20050 * 10, 5, 4)
    assert got[2].values[0].is_null and got[2].values[0].dtype == abi.DT_INT64 and got[2].values[4].value == 0.0  # all NULL: Int64 column, TOTAL 0.0
    # price * disc with a first product of three digits under scale 4: both sides fail, for SUM and for COUNT
    disc2 = np.array([5, 10, 1, 1, 1, 1], dtype=np.int64)
    price2 = np.array([10000, 20050, 99, 5, 123456, 777], dtype=np.int64)
    for agg in (A.sum(e), A.count(e)):
        for m, t in ((rt, rt.HipTable(1, [6])), (orc, orc.OracleTable(6))):
            if m is rt:
                t.append_decimal128_column(1, 15, 2, price2)
                t.append_decimal128_column(2, 15, 2, disc2)
                t.append_column(3, abi.DT_INT64, key)
            else:
                t.add(1, abi.DT_DECIMAL128, price2, precision=15, scale=2).add(2, abi.DT_DECIMAL128, disc2, precision=15, scale=2).add(3, abi.DT_INT64, key)
            with pytest.raises(abi.LlkvError) as err:
                m.groupby(t, None, [3], [agg], True)
            assert err.value.kind == "InvalidArgumentError" and "precision" in err.value.message


def test_decimal_arguments_the_gpu_path_hands_back(rt, abi):
    """What stays on the caller's route (LLKV_UNSUPPORTED): a Float operand or Modulo beside a Decimal (errors the reference raises
    on the first non-NULL row), intermediates the column statistics cannot keep inside 64 bits."""
    n = 8
    ht = rt.HipTable(1, [n])
    ht.append_decimal128_column(1, 18, 2, np.full(n, 10**17, dtype=np.int64))
    ht.append_column(2, abi.DT_FLOAT64, np.ones(n))
    ht.append_column(3, abi.DT_INT64, np.arange(n, dtype=np.int64) % 2)
    A, col = abi.AggregateSpec, abi.col
    for e in (col(1) * col(2), col(1) % col(3), col(1) * col(1), col(1) * 0.5):
        with pytest.raises(abi.LlkvError) as err:
            rt.groupby(ht, None, [3], [A.sum(e)], True)
        assert err.value.kind == "Unsupported", err.value
