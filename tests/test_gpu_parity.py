"""GPU (-m gpu): parity of the HIP path, called through the C ABI, with the CPU oracle.
Bit-exact for counts / integer sums / min / max / group keys; f64 sums and averages within 1e-9 relative
(the reference sums strictly left to right, llkv-aggregate/src/lib.rs:881-887; its own TPC-H tolerance is
1e-9, llkv-tpch/src/qualification.rs:39)."""
import math
import os

import numpy as np
import pytest

from conftest import DTYPES, build_aggs, build_filter, column_values, fval, golden, oracle_table, same_value

pytestmark = pytest.mark.gpu

REL = 1e-9
AGGS = golden("aggregates.json")


def stage_both(rt, orc, abi, columns, chunk_rows):
    """columns: [(field_id, dtype, numpy array | list[str][, valid mask])] → (HipTable, OracleTable)."""
    n = sum(chunk_rows)
    ht = rt.HipTable(1, chunk_rows)
    ot = orc.OracleTable(n)
    for col in columns:
        fid, dt, vals = col[:3]
        valid = col[3] if len(col) > 3 else None
        if dt == abi.DT_UTF8:
            if valid is None:
                ot.add(fid, dt, vals)
            else:
                ot.add(fid, dt, [v if ok else None for v, ok in zip(vals, valid)])
            ht.append_utf8_column(fid, vals, valid=valid)
        else:
            ot.add(fid, dt, vals, None if valid is None else list(valid))
            ht.append_column(fid, dt, vals, valid=valid)
    return ht, ot


def assert_values(got, want, ctx="", abs_floor=0.0):
    """``abs_floor``: sums whose terms cancel are compared absolutely below this (the 1e-9 of the contract is relative
    to the magnitude of what was added, not to a result that happens to be ≈ 0)."""
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g.dtype == w.dtype, (ctx, g, w)
        if w.value is None or isinstance(w.value, int):
            assert g.value == w.value, (ctx, g, w)
        else:
            ok = same_value(g.value, w.value, REL)
            if not ok and abs_floor and isinstance(g.value, float) and isinstance(w.value, float):
                ok = abs(g.value - w.value) <= abs_floor
            assert ok, (ctx, g, w)


def lineitem(tpch, abi, rows, scale, cols=None):
    d = tpch.gen_lineitem(rows, scale, cols)
    return d, [(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], d[c]) for c in d]


@pytest.mark.parametrize("sf,chunk", [("sf0.01", 8192), ("sf0.01", 131072), ("sf1", 131072)])
def test_tpch_queries_match_oracle(rt, orc, abi, tpch, sf, chunk):
    """configs[0..2] shapes: C1, Q6, Q1 on synthetic lineitem, ragged last chunk."""
    n = tpch.LINEITEM_ROWS[sf]
    d, cols = lineitem(tpch, abi, n, tpch.SCALE[sf])
    ht, ot = stage_both(rt, orc, abi, cols, tpch.chunk_rows(n, chunk))
    for name in ("c1", "q6"):
        q = tpch.QUERIES[name]()
        assert_values(rt.aggregate(ht, q.predicate, q.aggs), orc.aggregate(ot, q.predicate, q.aggs), name)
    q = tpch.q1()
    got = rt.groupby(ht, q.predicate, q.keys, q.aggs, True)
    want = orc.groupby(ot, q.predicate, q.keys, q.aggs, True)
    assert [[k.value for k in r.keys] for r in got] == [[k.value for k in r.keys] for r in want]
    for g, w in zip(got, want):
        assert_values(g.values, w.values, "q1")
    # without ORDER BY: first-appearance order (llkv-executor/src/lib.rs:5065-5089)
    got = rt.groupby(ht, q.predicate, q.keys, q.aggs, False)
    want = orc.groupby(ot, q.predicate, q.keys, q.aggs, False)
    assert [[k.value for k in r.keys] for r in got] == [[k.value for k in r.keys] for r in want]


@pytest.mark.parametrize("case", [c for c in AGGS["cases"]], ids=lambda c: c["name"])
def test_reference_aggregate_known_answers(rt, abi, case):
    """The reference's own known answers (tests/golden/aggregates.json) through the GPU path (run-time
    specialised kernels)."""
    n = len(column_values(case["columns"][0]))
    ht = rt.HipTable(1, [n] if n <= 65536 else [65536] * (n // 65536) + ([n % 65536] if n % 65536 else []))
    for c in case["columns"]:
        dt = DTYPES[c["dtype"]]
        vals = column_values(c)
        if dt == abi.DT_DECIMAL128:
            ht.append_decimal128_column(c["field_id"], c["precision"], c["scale"], vals)
        elif dt == abi.DT_UTF8:
            ht.append_utf8_column(c["field_id"], vals)
        elif any(v is None for v in vals):  # NULL cells: a row id absent from the column
            ht.append_column(c["field_id"], dt, np.array([0 if v is None else fval(v) for v in vals], dtype=abi.NUMPY_OF_DTYPE[dt]), valid=[v is not None for v in vals])
        else:
            ht.append_column(c["field_id"], dt, np.array([fval(v) for v in vals], dtype=abi.NUMPY_OF_DTYPE[dt]))
    pred = [build_filter(abi, case["filter"])] if "filter" in case else None
    aggs = build_aggs(abi, case["aggs"])
    if "group_by" in case:
        rows = rt.groupby(ht, pred, case["group_by"], aggs)  # (no way out: a plan shape the GPU path hands back would fail here)
        assert len(rows) == 1 and all(same_value(g.value, w) for g, w in zip(rows[0].values, case["expect"]))
        return
    if "expect_error" in case:
        with pytest.raises(abi.LlkvError) as e:
            rt.aggregate(ht, pred, aggs)
        assert e.value.kind == case["expect_error"], e.value
        return
    got = rt.aggregate(ht, pred, aggs)
    for g, w in zip(got, case["expect"]):
        assert same_value(g.value, w), (g, w)


def random_columns(rng, n):
    i64 = rng.integers(-10**9, 10**9, size=n).astype(np.int64)
    f64 = rng.normal(size=n) * 1e3
    f64[rng.random(n) < 0.01] = np.nan
    f64[rng.random(n) < 0.01] = -0.0
    i32 = rng.integers(-1000, 1000, size=n).astype(np.int32)
    big = rng.integers(-2**40, 2**40, size=n).astype(np.int64)  # rows·max|v| < 2^63: overflow provably impossible
    flags = rng.integers(0, 3, size=n)
    s = np.array([ord("x"), ord("y"), ord("z")], dtype=np.uint8)[flags]
    return i64, f64, i32, big, s


@pytest.mark.parametrize("chunks", [[1], [15, 16, 17], [512, 513, 1], [8192, 100, 8192, 7], [1000] * 37 + [123]])
@pytest.mark.parametrize("seed", [1, 2])
def test_random_plans_match_oracle(rt, orc, abi, chunks, seed):
    """Ragged chunk lists (tile tails, 16-row alignment padding), general predicate programs (And/Or/Not),
    every aggregate kind, NaN / -0.0 / large integers."""
    rng = np.random.default_rng(seed * 1000 + len(chunks))
    n = sum(chunks)
    i64, f64, i32, big, s = random_columns(rng, n)
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT64, i64), (2, abi.DT_FLOAT64, f64), (3, abi.DT_INT32, i32), (4, abi.DT_INT64, big),
                                       (5, abi.DT_UTF8, s)], chunks)
    F, O, B, E, A, col = abi.Filter, abi.Operator, abi.Bound, abi.Expr, abi.AggregateSpec, abi.col
    preds = [
        None,
        [F(1, O.LessThan(0))],
        [F(3, O.Range(B.Included(-500), B.Excluded(250))), F(2, O.GreaterThan(-100.5))],
        E.any_of([F(1, O.GreaterThanOrEquals(5 * 10**8)), E.all_of([F(3, O.In([1, 2, 3, 5, 8])), F(2, O.LessThanOrEquals(0))])]),
        E.not_(E.all_of([F(2, O.GreaterThan(0.0)), F(1, O.LessThan(10**8))])),  # NaN rows: NOT(x > 0) is true
        [F(1, O.Equals(123456789012))],  # matches nothing
        [F(5, O.Equals("y"))],
    ]
    aggs = [A.count_star(), A.sum(1), A.sum(2), A.avg(1), A.avg(2), A.min(1), A.max(1), A.min(2), A.max(2), A.total(1), A.total(2),
            A.count(2), A.sum(col(2) * (1 - col(2))), A.sum(col(1) + col(3)), A.sum(col(1) * 3 - 7), A.min(col(2) * 2.0), A.sum(4)]
    for p in preds:
        try:
            want = orc.aggregate(ot, p, aggs)
        except abi.LlkvError as e:
            with pytest.raises(abi.LlkvError) as g:
                rt.aggregate(ht, p, aggs)
            assert g.value.kind in (e.kind, "Unsupported"), (g.value, e)
            continue
        assert_values(rt.aggregate(ht, p, aggs), want, str(p))
    # grouped: dictionary key, PlanValue argument semantics (Int∘Int through f64)
    gaggs = [A.count_star(), A.sum(1), A.sum(col(1) * col(3)), A.avg(2), A.sum(col(2) * (1 - col(3))), A.max(1)]
    for p in (None, [F(3, O.GreaterThan(0))]):
        got, want = rt.groupby(ht, p, [5], gaggs, True), orc.groupby(ot, p, [5], gaggs, True)
        assert [r.keys[0].value for r in got] == [r.keys[0].value for r in want]
        for g, w in zip(got, want):
            assert_values(g.values, w.values, "grouped")


def test_empty_table_and_no_matches(rt, abi):
    A = abi.AggregateSpec
    ht = rt.HipTable(1, [])
    ht.append_column(1, abi.DT_INT64, [])
    ht.append_column(2, abi.DT_FLOAT64, [])
    got = rt.aggregate(ht, None, [A.count_star(), A.sum(1), A.sum(2), A.total(2), A.avg(1), A.min(2)])
    assert [g.value for g in got] == [0, None, None, 0.0, None, None]
    ht = rt.HipTable(1, [5])
    ht.append_column(1, abi.DT_INT64, np.arange(5, dtype=np.int64))
    ht.append_utf8_column(2, ["a", "b", "a", "c", "b"])
    assert rt.groupby(ht, [abi.Filter(1, abi.Operator.GreaterThan(100))], [2], [A.count_star()]) == []
    rows = rt.groupby(ht, None, [2], [A.count_star(), A.sum(1)])
    assert [(r.keys[0].value, r.values[0].value, r.values[1].value) for r in rows] == [("a", 2, 2), ("b", 2, 5), ("c", 1, 3)]


def test_integer_overflow_semantics(rt, abi):
    """SUM(Int64) overflow is an error like the reference's checked_add (llkv-aggregate/src/lib.rs:816-829);
    a total that fits but whose prefixes may not is handed back (UNSUPPORTED) instead of guessed."""
    A = abi.AggregateSpec
    big = 2**62
    ht = rt.HipTable(1, [4])
    ht.append_column(1, abi.DT_INT64, np.array([big, big, big, big], dtype=np.int64))
    with pytest.raises(abi.LlkvError) as e:
        rt.aggregate(ht, None, [A.sum(1)])
    assert e.value.kind == "InvalidArgumentError" and "integer overflow" in e.value.message
    # total fits, a prefix does not (big + big overflows before the negatives arrive): the order-dependent
    # error of the reference is reproduced by the exact in-order check
    ht = rt.HipTable(1, [4])
    ht.append_column(1, abi.DT_INT64, np.array([big, big, -big, -big], dtype=np.int64))
    with pytest.raises(abi.LlkvError) as e:
        rt.aggregate(ht, None, [A.sum(1)])
    assert e.value.kind == "InvalidArgumentError" and "integer overflow" in e.value.message
    with pytest.raises(abi.LlkvError) as e:
        rt.aggregate(ht, None, [A.avg(1)])
    assert e.value.kind == "InvalidArgumentError" and "AVG aggregate sum exceeds i64 range" in e.value.message
    # same multiset in an order whose prefixes stay in range: a value, not an error
    ht = rt.HipTable(1, [4])
    ht.append_column(1, abi.DT_INT64, np.array([big, -big, big, -big], dtype=np.int64))
    got = rt.aggregate(ht, None, [A.sum(1), A.avg(1), A.count_star()])
    assert [g.value for g in got] == [0, 0.0, 4]
    # with a filter: only the selected rows form the chain
    ht = rt.HipTable(1, [5])
    ht.append_column(1, abi.DT_INT64, np.array([big, big, 5, -big, -big], dtype=np.int64))
    ht.append_column(2, abi.DT_INT64, np.array([1, 0, 1, 1, 0], dtype=np.int64))
    got = rt.aggregate(ht, [abi.Filter(2, abi.Operator.Equals(1))], [A.sum(1)])
    assert got[0].value == 5
    # checked multiply in a computed projection (arrow numeric::mul, fast_numeric.rs:328-334)
    ht2 = rt.HipTable(1, [1])
    ht2.append_column(1, abi.DT_INT64, np.array([2**40], dtype=np.int64))
    with pytest.raises(abi.LlkvError) as e:
        rt.aggregate(ht2, None, [A.sum(abi.col(1) * 2**30)])
    assert e.value.kind == "Internal" and "overflow" in e.value.message.lower()


def test_results_are_bit_reproducible_and_gpu_count_invariant(rt, abi, tpch):
    """Same bits run to run, and the same bits whether the table is one shard or two shards whose exchange
    images are summed as integers (what the RCCL all-reduce does) — on ONE device."""
    n = 1_000_003
    chunks = tpch.chunk_rows(n, 32768)
    d = tpch.gen_lineitem(n, 1.0)
    q = tpch.q1()

    def stage(rank, world):
        ht = rt.HipTable(1, chunks, rank, world)
        lo = sum(chunks[:ht.first_chunk])
        for c in q.columns:
            fid, dt = tpch.LINEITEM_SCHEMA[c]
            part = d[c][lo:lo + ht.local_rows]
            if dt == abi.DT_UTF8:  # ranks must agree on the dictionary codes: table-wide sorted dictionary
                ht.append_utf8_column(fid, part, sorted({chr(int(v)) for v in np.unique(d[c])}))
            else:
                ht.append_column(fid, dt, part)
                # every rank must lower the same plan: table-wide statistics, as the binding installs them from
                # the column descriptor (dist.share_column_stats); a bound that misses staged values is refused
                if world > 1 and ht.local_column_stats(fid) is not None:
                    if ht.local_rows:
                        with pytest.raises(abi.LlkvError):
                            ht.set_column_stats(fid, int(part.min()) - 2, int(part.max()) - 1)
                    ht.set_column_stats(fid, int(d[c].min()), int(d[c].max()))
        return ht

    one = stage(0, 1)
    pq = rt.PreparedQuery(one, q.predicate, q.aggs, q.keys, True)
    r1 = pq.run()
    ex1 = pq.read_exchange()
    r1b = pq.run()
    assert np.array_equal(ex1, pq.read_exchange())
    flat = lambda rows: [(tuple(k.value for k in r.keys), tuple(np.float64(v.value).tobytes() if isinstance(v.value, float) else v.value for v in r.values)) for r in rows]
    assert flat(r1) == flat(r1b)
    # the reduction association belongs to the canonical tile list, not to the launch geometry: g workgroups stream
    # the tiles [b·n/g, (b+1)·n/g) (engine.cpp: pick_scan_grid follows the LOCAL tile count) and the bits do not move
    for wgs in ("1", "7", "33", "62", "100000"):
        os.environ["LLKV_HIP_SCAN_WGS"] = wgs
        try:
            pg = rt.PreparedQuery(one, q.predicate, q.aggs, q.keys, True)
            assert flat(pg.run()) == flat(r1), wgs
            assert np.array_equal(pg.read_exchange(), ex1), wgs
            pg.close()
        finally:
            del os.environ["LLKV_HIP_SCAN_WGS"]
    # register-resident states (ungrouped plans) too: a workgroup that streams a run of tiles still reduces and publishes
    # every tile on its own
    q6 = tpch.q6()
    p6 = rt.PreparedQuery(one, q6.predicate, q6.aggs, q6.keys, False)
    r6, e6 = flat(p6.run()), p6.read_exchange()
    for wgs in ("1", "7", "33", "100", "245"):
        os.environ["LLKV_HIP_SCAN_WGS"] = wgs
        try:
            pg = rt.PreparedQuery(one, q6.predicate, q6.aggs, q6.keys, False)
            assert flat(pg.run()) == r6 and np.array_equal(pg.read_exchange(), e6), wgs
            assert flat(pg.run()) == r6, wgs  # (with the fold of the previous execution riding in the launch)
            pg.close()
        finally:
            del os.environ["LLKV_HIP_SCAN_WGS"]
    for world in (2, 4, 8):
        total = np.zeros_like(ex1).view(np.int64)
        last = None
        for rank in range(world):
            pr = rt.PreparedQuery(stage(rank, world), q.predicate, q.aggs, q.keys, True)
            pr.launch()
            ex = pr.read_exchange()
            owned = [o for o in range(8) if o * world // 8 == rank]
            assert not ex[[o for o in range(8) if o not in owned]].any(), "non-owned octants must be zero"
            total += ex.view(np.int64)
            last = pr
        assert np.array_equal(total.view(np.uint64), ex1), f"world={world}"
        assert flat(last.finish_from_host(total.view(np.uint64))) == flat(r1)


@pytest.mark.parametrize("name", ["q6", "q1"])
def test_full_size_properties_sf10(rt, abi, tpch, name):
    """BASELINE.json sizes (SF10, 59 986 052 rows): size-independent checks — integer lanes exact against
    numpy, f64 sums within 1e-9 of a pairwise numpy sum, keys sorted, idempotent."""
    n = tpch.LINEITEM_ROWS["sf10"]
    q = tpch.QUERIES[name]()
    d = tpch.gen_lineitem(n, 10.0, q.columns)
    ht = rt.HipTable(1, tpch.chunk_rows(n))
    for c in q.columns:
        fid, dt = tpch.LINEITEM_SCHEMA[c]
        ht.append_utf8_column(fid, d[c]) if dt == abi.DT_UTF8 else ht.append_column(fid, dt, d[c])
    pq = rt.PreparedQuery(ht, q.predicate, q.aggs, q.keys, q.order_by_keys)
    rows = pq.run()
    again = pq.run()
    assert [[v.value for v in r.values] for r in rows] == [[v.value for v in r.values] for r in again]
    if name == "q6":
        m = (d["l_shipdate"] >= 8766) & (d["l_shipdate"] < 9131) & (d["l_discount"] >= 0.05) & (d["l_discount"] <= 0.07) & (d["l_quantity"] < 24)
        want = float((d["l_extendedprice"][m] * d["l_discount"][m]).sum())
        assert abs(rows[0].values[0].value - want) <= REL * want
    else:
        keys = [tuple(k.value for k in r.keys) for r in rows]
        assert keys == sorted(keys) and len(keys) == 4
        sel = d["l_shipdate"] <= 10471
        assert sum(r.values[7].value for r in rows) == int(sel.sum())
        for r in rows:
            g = sel & (d["l_returnflag"] == ord(r.keys[0].value)) & (d["l_linestatus"] == ord(r.keys[1].value))
            assert r.values[0].value == int(d["l_quantity"][g].sum())  # exact
            assert r.values[7].value == int(g.sum())
            p, disc, tax = d["l_extendedprice"][g], d["l_discount"][g], d["l_tax"][g]
            assert abs(r.values[1].value - p.sum()) <= REL * p.sum()
            assert abs(r.values[2].value - (p * (1 - disc)).sum()) <= REL * p.sum()
            assert abs(r.values[3].value - (p * (1 - disc) * (1 + tax)).sum()) <= REL * p.sum()
            assert abs(r.values[4].value - d["l_quantity"][g].sum() / g.sum()) <= 1e-12 * 50
            assert abs(r.values[6].value - disc.sum() / g.sum()) <= REL


def test_full_size_selection_join_and_q3_sf10(rt, abi, tpch):
    """BASELINE.json sizes for the other routes (SF10: 59 986 052 lineitems, 14 996 513 orders, 1 500 000 customers),
    checked through properties numpy can state: the row-id vector of Q6's predicate, the scan windows, the pair
    sequence of lineitem ⋈ orders (every lineitem has exactly one order), and Q3's top 10 (configs[4])."""
    n = tpch.LINEITEM_ROWS["sf10"]
    scale = tpch.SCALE["sf10"]
    li = tpch.gen_lineitem(n, scale, ["l_orderkey", "l_quantity", "l_shipdate", "l_extendedprice", "l_discount"])
    lt = rt.HipTable(1, tpch.chunk_rows(n))
    for c in li:
        lt.append_column(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], li[c])
    # ---- selection-vector route
    q6 = tpch.q6()
    m = (li["l_shipdate"] >= 8766) & (li["l_shipdate"] < 9131) & (li["l_discount"] >= 0.05) & (li["l_discount"] <= 0.07) & (li["l_quantity"] < 24)
    ids = rt.filter_row_ids(lt, q6.predicate)
    assert np.array_equal(ids, np.flatnonzero(m).astype(np.uint64))
    import ctypes as C
    seen = []

    def consume(b):
        k = int(b.num_rows)
        seen.append((k, int(b.row_ids[0]), int(b.row_ids[k - 1]), float(np.ctypeslib.as_array(C.cast(b.columns[0].values, C.POINTER(C.c_double)), shape=(k,)).sum())))

    rt.scan_stream(lt, [abi.col(tpch.L_EXTENDEDPRICE) * abi.col(tpch.L_DISCOUNT)], q6.predicate, include_row_ids=True, consume=consume)
    assert [k for k, *_ in seen] == [65536] * (len(ids) // 65536) + ([len(ids) % 65536] if len(ids) % 65536 else [])
    assert [a for _, a, _, _ in seen] == [int(ids[w]) for w in range(0, len(ids), 65536)]
    want = float((li["l_extendedprice"][m] * li["l_discount"][m]).sum())
    assert abs(sum(x for *_, x in seen) - want) <= REL * want
    # ---- join_stream: lineitem ⋈ orders
    n_ord = tpch.orders_for_lineitems(n)
    od = tpch.gen_orders(n_ord, scale)
    ot_ = rt.HipTable(2, tpch.chunk_rows(n_ord))
    for c, (fid, dt) in tpch.ORDERS_SCHEMA.items():
        ot_.append_column(fid, dt, od[c])
    assert np.all(np.diff(od["o_orderkey"]) > 0)
    order_of = np.searchsorted(od["o_orderkey"], li["l_orderkey"])
    state = {"pairs": 0, "ok": True}
    CB = abi.ON_JOIN_BATCH
    def on_pairs(pl, pr, k, _u):
        l = np.ctypeslib.as_array(pl, shape=(k,)); r = np.ctypeslib.as_array(pr, shape=(k,))
        p0 = state["pairs"]
        state["ok"] &= bool(l[0] == p0 and l[-1] == p0 + k - 1 and np.array_equal(r, order_of[p0:p0 + k].astype(np.uint64)))
        state["pairs"] += k
    ck = (abi.CJoinKey * 1)(); ck[0].left_field, ck[0].right_field, ck[0].null_equals_null = tpch.L_ORDERKEY, tpch.O_ORDERKEY, 0
    opts = abi.CJoinOptions(abi.JOIN_INNER, 65536, 0)
    cb = CB(on_pairs)
    rt.check(rt.lib().llkv_hip_join_stream(lt.handle, ot_.handle, ck, C.c_uint32(1), C.byref(opts), cb, None))
    assert state["pairs"] == n and state["ok"]
    # ---- Q3 (configs[4], single GPU)
    n_cust = tpch.customers_for_scale(scale)
    cu = tpch.gen_customer(n_cust, scale)
    ct = rt.HipTable(3, tpch.chunk_rows(n_cust))
    ct.append_column(tpch.C_CUSTKEY, abi.DT_INT64, cu["c_custkey"])
    ct.append_utf8_column(tpch.C_MKTSEGMENT, [tpch.SEGMENTS[c] for c in cu["c_mktsegment"]])
    F, O, col = abi.Filter, abi.Operator, abi.col
    D = tpch.DATE_1995_03_15
    got, total = rt.join_groupby_topk(lt, [F(tpch.L_SHIPDATE, O.GreaterThan(D))], tpch.L_ORDERKEY, ot_, [F(tpch.O_ORDERDATE, O.LessThan(D))], tpch.O_ORDERKEY,
                                      col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT)), payload_fields=[tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], limit=10,
                                      dim_fk=tpch.O_CUSTKEY, dim2=ct, dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))], dim2_key=tpch.C_CUSTKEY)
    building = np.zeros(int(cu["c_custkey"].max()) + 1, dtype=bool)
    building[cu["c_custkey"][cu["c_mktsegment"] == list(tpch.SEGMENTS).index("BUILDING")]] = True
    o_ok = (od["o_orderdate"] < D) & building[od["o_custkey"]]
    l_ok = (li["l_shipdate"] > D) & o_ok[order_of]
    rev = np.bincount(order_of[l_ok], weights=(li["l_extendedprice"][l_ok] * (1 - li["l_discount"][l_ok])), minlength=n_ord)
    cnt = np.bincount(order_of[l_ok], minlength=n_ord)
    assert total == int((cnt > 0).sum())
    top = sorted(np.flatnonzero(cnt > 0).tolist(), key=lambda i: (-rev[i], od["o_orderdate"][i]))[:10]
    assert [r[0] for r in got] == [int(od["o_orderkey"][i]) for i in top]
    for r, i in zip(got, top):
        assert r[2] == int(cnt[i]) and r[3] == int(od["o_orderdate"][i]) and abs(r[1] - rev[i]) <= REL * rev[i]


def test_configs3_q1_sf10_as_eight_shards_sums_to_the_single_gpu_bits(rt, abi, tpch):
    """BASELINE.json configs[3] at full size on ONE device: SF10 lineitem staged as 8 shards, one after another (what the
    8 ranks hold), Q1 over each, the exchange images summed as integers — what ncclAllReduce(int64, sum) does — and
    finished: the same bits as the single-table run, and every shard leaves the octants it does not own at zero."""
    n = tpch.LINEITEM_ROWS["sf10"]
    chunks = tpch.chunk_rows(n)
    q = tpch.q1()
    d = tpch.gen_lineitem(n, tpch.SCALE["sf10"], q.columns)
    dicts = {c: sorted({chr(int(v)) for v in np.unique(d[c])}) for c in q.columns if tpch.LINEITEM_SCHEMA[c][1] == abi.DT_UTF8}
    stats = {c: (int(d[c].min()), int(d[c].max())) for c in q.columns if tpch.LINEITEM_SCHEMA[c][1] in (abi.DT_INT64, abi.DT_DATE32)}

    def stage(rank, world):
        ht = rt.HipTable(1, chunks, rank, world)
        lo = sum(chunks[:ht.first_chunk])
        for c in q.columns:
            fid, dt = tpch.LINEITEM_SCHEMA[c]
            part = d[c][lo:lo + ht.local_rows]
            if dt == abi.DT_UTF8:
                ht.append_utf8_column(fid, part, dicts[c])
            else:
                ht.append_column(fid, dt, part)
                if world > 1 and c in stats and ht.local_column_stats(fid) is not None:
                    ht.set_column_stats(fid, *stats[c])
        return ht

    flat = lambda rows: [(tuple(k.value for k in r.keys), tuple(np.float64(v.value).tobytes() if isinstance(v.value, float) else v.value for v in r.values)) for r in rows]
    one = stage(0, 1)
    pq = rt.PreparedQuery(one, q.predicate, q.aggs, q.keys, True)
    want = flat(pq.run())
    ex1 = pq.read_exchange()
    pq.close(); one.close()
    total = np.zeros_like(ex1).view(np.int64)
    last = None
    for rank in range(8):
        if last is not None:
            last[0].close(); last[1].close()
        ht = stage(rank, 8)
        assert abs(ht.local_rows - n // 8) <= 2 * max(chunks)
        pr = rt.PreparedQuery(ht, q.predicate, q.aggs, q.keys, True)
        pr.launch()
        ex = pr.read_exchange()
        assert not ex[[o for o in range(8) if o != rank]].any()
        total += ex.view(np.int64)
        last = (pr, ht)
    assert np.array_equal(total.view(np.uint64), ex1)
    assert flat(last[0].finish_from_host(total.view(np.uint64))) == want


def test_configs4_q3_sf10_probe_side_in_eight_shards(rt, abi, tpch):
    """BASELINE.json configs[4] at full size on ONE device: orders and customer replicated, SF10 lineitem staged as the 8
    shards the ranks hold; the phased pipeline's collectives done by hand (general form: counts summed as the int64
    all-reduce does, straddler pairs, candidates; range form: boundary runs, candidates).  Keys, counts, group total and
    the revenue BITS of the top 10 equal the single-GPU answer in both forms; a clustered fact table exchanges only the
    orders cut by a shard boundary."""
    import torch
    n, scale = tpch.LINEITEM_ROWS["sf10"], tpch.SCALE["sf10"]
    D = tpch.DATE_1995_03_15
    cols = ["l_orderkey", "l_shipdate", "l_extendedprice", "l_discount"]
    li = tpch.gen_lineitem(n, scale, cols)
    n_ord = tpch.orders_for_lineitems(n); od = tpch.gen_orders(n_ord, scale)
    n_cust = tpch.customers_for_scale(scale); cu = tpch.gen_customer(n_cust, scale)
    ot_ = rt.HipTable(2, tpch.chunk_rows(n_ord))
    for c, (fid, dt) in tpch.ORDERS_SCHEMA.items():
        ot_.append_column(fid, dt, od[c])
    ct = rt.HipTable(3, tpch.chunk_rows(n_cust))
    ct.append_column(tpch.C_CUSTKEY, abi.DT_INT64, cu["c_custkey"])
    ct.append_utf8_column(tpch.C_MKTSEGMENT, [tpch.SEGMENTS[c] for c in cu["c_mktsegment"]])
    F, O, col = abi.Filter, abi.Operator, abi.col
    chunks = tpch.chunk_rows(n)

    def fact(rank, world):
        t = rt.HipTable(1, chunks, rank, world)
        lo = sum(chunks[:t.first_chunk])
        for c in cols:
            t.append_column(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], li[c][lo:lo + t.local_rows])
        return t

    args = lambda t: dict(fact=t, fact_filters=[F(tpch.L_SHIPDATE, O.GreaterThan(D))], fact_key=tpch.L_ORDERKEY, dim=ot_,
                          dim_filters=[F(tpch.O_ORDERDATE, O.LessThan(D))], dim_key=tpch.O_ORDERKEY, sum_expr=col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT)),
                          payload_fields=[tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], dim_fk=tpch.O_CUSTKEY, dim2=ct,
                          dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))], dim2_key=tpch.C_CUSTKEY)
    whole = fact(0, 1)
    want, want_total = rt.join_groupby_topk(limit=10, **args(whole))
    whole.close()
    bits = lambda rws: [(r[0], np.float64(r[1]).tobytes(), r[2], r[3], r[4]) for r in rws]
    assert len(want) == 10 and want_total > 100_000
    shards = [fact(r, 8) for r in range(8)]
    # general form
    joins = [rt.JoinAgg(**args(t)) for t in shards]
    bufs = [j.counts_buffer() for j in joins]
    assert len({k for _, k in bufs}) == 1
    tensors = [_device_i64(p, k) for p, k in bufs]
    total = sum(t.clone() for t in tensors)
    for t in tensors:
        t.copy_(total)
    torch.cuda.synchronize()
    strad = [j.straddlers() for j in joins]
    assert sum(len(g) for g, _ in strad) <= 8 * 7  # at most one order per shard boundary, at most 7 lines each
    folded = rt.fold_straddlers([g for g, _ in strad], [v for _, v in strad])
    parts = [j.candidates(folded, r, 10) for r, j in enumerate(joins)]
    assert bits(rt.merge_join_rows([row for rows_r, _ in parts for row in rows_r], 2, 10)) == bits(want)
    assert sum(k for _, k in parts) == want_total
    del joins, tensors, total
    # range form: every rank selects the orders of its own key range, ~1 KB of boundary runs per rank is all they exchange
    ranged = [rt.JoinAgg(ranged=True, **args(t)) for t in shards]
    blocks = [j.boundary() for j in ranged]
    parts = [j.finish_ranged(blocks, r, 10) for r, j in enumerate(ranged)]
    assert bits(rt.merge_join_rows([row for rows_r, _ in parts for row in rows_r], 2, 10)) == bits(want)
    assert sum(k for _, k in parts) == want_total
    assert sum(len(b) for b in blocks) == 8 * 1088


@pytest.mark.parametrize("key", ["l_partkey", "l_orderkey"])
def test_full_size_sort_based_group_by_sf10(rt, abi, tpch, key):
    """SF10, GROUP BY l_partkey (2 000 000 groups, unsorted input) / l_orderkey (14 996 513 groups, input already in
    key order: the no-sort shortcut): group count and key order against numpy, and a sample of 3 000 groups cell by
    cell — counts and integer sums exact, f64 sums within 1e-9."""
    import ctypes as C
    n = tpch.LINEITEM_ROWS["sf10"]
    S = tpch.LINEITEM_SCHEMA
    d = tpch.gen_lineitem(n, tpch.SCALE["sf10"], [key, "l_quantity", "l_extendedprice", "l_discount"])
    t = rt.HipTable(1, tpch.chunk_rows(n))
    for c in d:
        t.append_column(S[c][0], S[c][1], d[c])
    A, col = abi.AggregateSpec, abi.col
    q = rt.PreparedQuery(t, None, [A.count_star(), A.sum(S["l_quantity"][0]), A.sum(col(S["l_extendedprice"][0]) * (1 - col(S["l_discount"][0])))], [S[key][0]], True)
    q.launch(0)
    L = rt.lib()
    rt.check(L.llkv_hip_query_finish(q._h, None))
    uniq, inv, counts = np.unique(d[key], return_inverse=True, return_counts=True)
    assert L.llkv_hip_query_num_groups(q._h) == len(uniq)
    qty = np.bincount(inv, weights=None, minlength=len(uniq)) * 0 + np.bincount(inv, weights=d["l_quantity"].astype(np.float64), minlength=len(uniq))
    rev = np.bincount(inv, weights=d["l_extendedprice"] * (1 - d["l_discount"]), minlength=len(uniq))
    v = abi.CValue()
    for g in np.random.default_rng(3).integers(0, len(uniq), size=3000).tolist() + [0, len(uniq) - 1]:
        rt.check(L.llkv_hip_query_group_key(q._h, g, 0, C.byref(v)))
        assert abi.Value.from_c(v).value == int(uniq[g])
        got = []
        for a in range(3):
            rt.check(L.llkv_hip_query_value(q._h, g, a, C.byref(v)))
            got.append(abi.Value.from_c(v).value)
        assert got[0] == int(counts[g]) and got[1] == int(qty[g]) and abs(got[2] - rev[g]) <= REL * rev[g], (g, got)


TABLE = golden("table_scan.json")
NULLABLE_TABLES = {name for name, t in TABLE["tables"].items() if any(v is None for c in t["columns"] for v in c["values"])}


@pytest.mark.parametrize("case", [c for c in TABLE["cases"] if c["table"] not in NULLABLE_TABLES], ids=lambda c: c["name"])
def test_reference_table_scan_known_answers(rt, abi, case):
    """The reference's filtered-scan / And / Or / Not / IN / projection / computed-projection tests
    (tests/golden/table_scan.json: llkv-table/src/table.rs:1697-2906, llkv-executor/src/lib.rs:13880-13925) through
    llkv_hip_scan_stream.  (The include-nulls cases run in test_reference_include_nulls_known_answers.)"""
    from conftest import build_predicate, build_expr
    tdef = TABLE["tables"][case["table"]]
    ht = rt.HipTable(1, [tdef["rows"]])
    for c in tdef["columns"]:
        dt = DTYPES[c["dtype"]]
        ht.append_column(c["field_id"], dt, np.array(c["values"], dtype=abi.NUMPY_OF_DTYPE[dt]))
    projections = [p if isinstance(p, int) else build_expr(abi, p) for p in case["project"]]
    if "expect_error" in case:
        with pytest.raises(abi.LlkvError) as e:
            rt.scan_stream(ht, projections, build_predicate(abi, case["predicate"]))
        assert e.value.kind == case["expect_error"]
        return
    batches = rt.scan_stream(ht, projections, build_predicate(abi, case["predicate"]))
    cols = [[] for _ in projections]
    for bcols, _ in batches:
        assert len(bcols[0]) > 0
        for i, c in enumerate(bcols):
            cols[i].extend(c)
    if "expect_sorted" in case:
        assert sorted(cols[0]) == case["expect_sorted"]
        return
    assert cols == case["expect"]
    if "expect_sum" in case:
        assert sum(cols[0]) == case["expect_sum"]
    if "expect_min" in case:
        assert min(cols[0]) == case["expect_min"] and max(cols[0]) == case["expect_max"]
    if "expect_sqrt" in case:
        assert [float(np.sqrt(np.float64(v))) for v in cols[0]] == case["expect_sqrt"]


@pytest.mark.parametrize("chunks", [[7], [15, 16, 17], [8192, 100, 8192, 7], [131072, 131072, 50000]])
def test_scan_stream_and_row_ids_match_oracle(rt, orc, abi, chunks):
    """Selection vectors: ascending ids, 65 536-row windows, gather of every storage type, computed projections."""
    rng = np.random.default_rng(len(chunks))
    n = sum(chunks)
    i64, f64, i32, big, s = random_columns(rng, n)
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT64, i64), (2, abi.DT_FLOAT64, f64), (3, abi.DT_INT32, i32), (5, abi.DT_UTF8, s)], chunks)
    F, O, B, E, col = abi.Filter, abi.Operator, abi.Bound, abi.Expr, abi.col
    preds = [None, [F(1, O.LessThan(0))], E.any_of([F(3, O.In([1, 2, 3])), E.not_(F(2, O.GreaterThan(-500.0)))]), [F(1, O.Equals(2**40))]]
    for p in preds:
        want_ids = orc.filter_row_ids(ot, p)
        got_ids = rt.filter_row_ids(ht, p)
        assert np.array_equal(got_ids, want_ids)
        projs = [1, 2, 3, 5, col(2) * 2.0 + col(1), col(1) + col(3)]
        got = rt.scan_stream(ht, projs, p, include_row_ids=True)
        want = orc.scan_stream(ot, projs, p, include_nulls=True, include_row_ids=True)
        assert [len(b[1]) for b in got] == [len(b[1]) for b in want]
        assert all(len(b[1]) <= 65536 and len(b[1]) > 0 for b in got)
        for (gc, gr), (wc, wr) in zip(got, want):
            assert gr == wr
            for a, b in zip(gc, wc):
                assert len(a) == len(b)
                for x, y in zip(a, b):
                    assert (x == y) or (isinstance(x, float) and math.isnan(x) and math.isnan(y)), (x, y)


@pytest.mark.parametrize("chunks", [[11], [4096, 4097, 5]])
def test_compares_with_the_null_literal_and_without_a_column(rt, orc, abi, chunks):
    """Expr::Compare where one side IS the NULL literal (`x = NULL`: literal_to_array gives a NullArray, coerce_types casts it to the
    other side's type — the compare is NULL on every row: nothing matches, nothing is determined, NOT of it holds nowhere either; the
    other side is still evaluated and its checked arithmetic can fail the scan) and where no side names a field
    (evaluate_constant_compare, llkv-scan/src/predicate.rs:887-907: TRUE keeps every row of the table, FALSE none, a NULL side
    determines nothing) — alone, negated, and inside AND / OR trees with ordinary leaves."""
    rng = np.random.default_rng(41 + len(chunks))
    n = sum(chunks)
    a = rng.integers(-50, 50, size=n).astype(np.int64)
    b = rng.normal(size=n)
    va = rng.random(n) > 0.2
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT64, a, va), (2, abi.DT_FLOAT64, b)], chunks)
    E, F, O, col, A = abi.Expr, abi.Filter, abi.Operator, abi.col, abi.AggregateSpec
    L = abi.ScalarExpr.literal
    NULL = L(abi.Literal.of(None))
    leaf = E.pred(F(1, O.GreaterThan(0)))
    atoms = [E.compare(col(1), abi.CMP_EQ, NULL), E.compare(NULL, abi.CMP_LT, col(2)), E.compare(col(1) * 3 + 1, abi.CMP_NOT_EQ, NULL),
             E.compare(L(1) + 2, abi.CMP_LT, 4), E.compare(L(5), abi.CMP_LT, 4.5), E.compare(L(2.5) * 2, abi.CMP_EQ, 5), E.compare(NULL, abi.CMP_EQ, 1),
             E.compare(L(1) - 1, abi.CMP_GT_EQ, L(0) * 7),
             # … and the constant forms of IN (evaluate_constant_in_list :909-963) and IS NULL (:276-284)
             E.in_list(L(3), [1, L(1) + 2, 5]), E.in_list(L(3), [1, NULL, 5]), E.in_list(L(3), [1, NULL, 3.0]), E.in_list(NULL, [1, 2]), E.in_list(L(4), [1, 2], negated=True),
             E.in_list(L(4), [NULL], negated=True), E.is_null(NULL), E.is_null(L(1) + 2), E.is_null(L(7) / 0), E.is_null(L(1.5), negated=True)]
    preds = []
    for x in atoms:
        preds += [x, E.not_(x), E.all_of([leaf, x]), E.any_of([leaf, E.not_(x)]), E.not_(E.any_of([x, leaf]))]
    for p in preds:
        assert np.array_equal(rt.filter_row_ids(ht, p), orc.filter_row_ids(ot, p))
        assert_values(rt.aggregate(ht, p, [A.count_star(), A.sum(1), A.min(2)]), orc.aggregate(ot, p, [A.count_star(), A.sum(1), A.min(2)]))
    assert len(orc.filter_row_ids(ot, atoms[3])) == n and len(orc.filter_row_ids(ot, atoms[0])) == 0
    # the side beside the NULL literal is evaluated: its overflow fails the scan on both sides
    big = a.copy()
    big[n // 2] = 2**62
    hb, ob = stage_both(rt, orc, abi, [(1, abi.DT_INT64, big)], chunks)
    for m, t in ((rt, hb), (orc, ob)):
        for p in (E.compare(col(1) * 4, abi.CMP_GT, NULL), E.not_(E.compare(col(1) * 4, abi.CMP_GT, NULL))):  # (NOT subtracts the rows from the domain: they are computed all the same)
            with pytest.raises(abi.LlkvError) as e:
                m.filter_row_ids(t, p)
            assert e.value.kind == "Internal" and "overflow" in e.value.message.lower(), m
    # what stays refused: a NULL literal inside a side's arithmetic; constant sides that do not fold to a literal
    for bad in (E.compare(col(1) + NULL, abi.CMP_LT, 4), E.compare(L(1) % 0, abi.CMP_LT, 4)):
        for m, t in ((rt, ht), (orc, ot)):
            with pytest.raises(abi.LlkvError) as e:
                m.filter_row_ids(t, bad)
            assert e.value.kind == "Unsupported", (m, bad)


@pytest.mark.parametrize("chunks", [[11], [4096, 4097, 5]])
def test_int32_only_arithmetic_runs_on_the_checked_32_bit_kernels(rt, orc, abi, chunks):
    """An expression whose every leaf is an Int32 (UInt32) column has that ROOT type (get_common_type, llkv-compute/src/kernels.rs:
    179-242: same ⊕ same → same) and the fast path runs arrow's checked 32-bit kernels over it (fast_numeric.rs:333-355): an Int32 /
    UInt32 result column, an error where an intermediate leaves 32 bits.  One Int64 leaf or literal makes the root Int64: the
    columns are cast up before the first kernel and `(a + b) * w` cannot overflow at 32 bits.  Aggregates over such an expression
    are refused on both sides (the reference has no Int32 accumulator)."""
    rng = np.random.default_rng(7 + len(chunks))
    n = sum(chunks)
    a = rng.integers(-40_000, 40_000, size=n).astype(np.int32)
    b = rng.integers(-40_000, 40_000, size=n).astype(np.int32)
    u = rng.integers(0, 60_000, size=n).astype(np.uint32)
    v = rng.integers(0, 60_000, size=n).astype(np.uint32)
    w = rng.integers(-2**40, 2**40, size=n).astype(np.int64)
    vb = rng.random(n) > 0.2
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT32, a), (2, abi.DT_INT32, b, vb), (3, abi.DT_UINT32, u), (4, abi.DT_UINT32, v), (5, abi.DT_INT64, w)], chunks)
    E, col, A = abi.Expr, abi.col, abi.AggregateSpec
    projs = [col(1) + col(2), col(1) * col(2) - col(1), col(1) - col(2), (col(1) + col(2)) * col(5), col(3) + col(4), col(3) * col(4), col(1) + col(3)]
    preds = [None, E.compare(col(1) + col(2), abi.CMP_GT, col(1) * col(2)), E.compare(col(3) * col(4), abi.CMP_LT_EQ, col(5)), E.compare(col(1) - col(2), abi.CMP_LT, col(3) + col(4))]
    for p in preds:
        assert np.array_equal(rt.filter_row_ids(ht, p), orc.filter_row_ids(ot, p))
        got = rt.scan_stream(ht, projs, p, include_row_ids=True)
        want = orc.scan_stream(ot, projs, p, include_nulls=True, include_row_ids=True)
        assert [b_[1] for b_ in got] == [b_[1] for b_ in want] and [b_[0] for b_ in got] == [b_[0] for b_ in want]
    # the products stay below 2^31 in magnitude only because the operands are small: 40 000² = 1.6e9 < 2^31; widen one operand and both sides fail
    a2 = a.copy()
    a2[n // 2] = 2**31 - 1
    u2 = u.copy()
    u2[n // 3] = 0
    h2, o2 = stage_both(rt, orc, abi, [(1, abi.DT_INT32, a2), (2, abi.DT_INT32, np.ones(n, dtype=np.int32)), (3, abi.DT_UINT32, u2), (4, abi.DT_UINT32, np.ones(n, dtype=np.uint32)), (5, abi.DT_INT64, w)], chunks)
    for expr in (col(1) + col(2), col(3) - col(4)):  # i32::MAX + 1; 0 − 1 in UInt32
        for m, t in ((rt, h2), (orc, o2)):
            with pytest.raises(abi.LlkvError) as e:
                m.scan_stream(t, [expr], None)
            assert e.value.kind == "Internal" and "overflow" in e.value.message.lower(), (m, expr)
            with pytest.raises(abi.LlkvError) as e:
                m.filter_row_ids(t, E.compare(expr, abi.CMP_GT, col(5)))
            assert e.value.kind == "Internal" and "overflow" in e.value.message.lower(), (m, expr)
    # … while an Int64 leaf widens the whole expression: no error, the same values
    assert rt.scan_stream(h2, [(col(1) + col(2)) + col(5)], None) == orc.scan_stream(o2, [(col(1) + col(2)) + col(5)], None, include_nulls=True)
    for m, t in ((rt, ht), (orc, ot)):
        with pytest.raises(abi.LlkvError) as e:
            m.aggregate(t, None, [A.sum(col(1) + col(2))])
        assert e.value.kind == "Unsupported", m


@pytest.mark.parametrize("chunks", [[11], [4096, 4097, 5], [65536, 70000]])
def test_row_ids_with_gaps_are_reported_as_the_tables_ids(rt, orc, abi, chunks):
    """A table whose row ids are not 0 … n − 1 (rows removed before it was staged; the row-id shadow column of
    llkv-column-map/src/store/descriptor.rs, gathered over by store/gather.rs:764-884): positions order as the ids do, so
    everything but the REPORTED ids is what a dense table gives, and the reported ids are ids[position]."""
    rng = np.random.default_rng(sum(chunks) + 3)
    n = sum(chunks)
    i64, f64, i32, big, s = random_columns(rng, n)
    ids = np.cumsum(rng.integers(1, 5, size=n)).astype(np.uint64) + np.uint64(2**40 if len(chunks) == 1 else 0)
    cols = [(1, abi.DT_INT64, i64), (2, abi.DT_FLOAT64, f64), (3, abi.DT_INT32, i32), (5, abi.DT_UTF8, s)]
    ht, ot = stage_both(rt, orc, abi, cols, chunks)
    ht.set_row_ids(ids)
    F, O, E, col, A = abi.Filter, abi.Operator, abi.Expr, abi.col, abi.AggregateSpec
    preds = [None, [F(1, O.LessThan(0))], E.any_of([F(3, O.In([1, 2, 3])), E.not_(F(2, O.GreaterThan(-500.0)))]), [F(1, O.Equals(2**40))]]
    for p in preds:
        want_pos = orc.filter_row_ids(ot, p)
        got_ids = rt.filter_row_ids(ht, p)
        assert got_ids.dtype == np.uint64 and np.array_equal(got_ids, ids[want_pos])
        projs = [1, 5, col(2) * 2.0 + col(1)]
        got = rt.scan_stream(ht, projs, p, include_row_ids=True)
        want = orc.scan_stream(ot, projs, p, include_nulls=True, include_row_ids=True)
        assert [len(b[1]) for b in got] == [len(b[1]) for b in want]
        for (gc, gr), (wc, wr) in zip(got, want):
            assert np.array_equal(np.asarray(gr, dtype=np.uint64), ids[np.asarray(wr, dtype=np.int64)])
            for a, b in zip(gc, wc):
                assert len(a) == len(b)
                assert all((x == y) or (isinstance(x, float) and math.isnan(x) and math.isnan(y)) for x, y in zip(a, b))
        # what reports no ids is untouched
        assert_values(rt.aggregate(ht, p, [A.count_star(), A.sum(1), A.min(1)]), orc.aggregate(ot, p, [A.count_star(), A.sum(1), A.min(1)]))
    # GROUP BY on the sort-based route (sparse keys) orders its groups by the first POSITION of each — not by the table's ids,
    # which here exceed the row count the sort's key bits are sized from: first-appearance and key order, as over a dense table
    sparse = (rng.integers(0, 40, size=n) * 1_000_003).astype(np.int64)
    hs, os_ = stage_both(rt, orc, abi, [(1, abi.DT_INT64, i64), (2, abi.DT_FLOAT64, f64), (6, abi.DT_INT64, sparse)], chunks)
    hs.set_row_ids(ids)
    for ordered in (False, True):
        pq = rt.PreparedQuery(hs, None, [A.count_star(), A.sum(1), A.min(2)], [6], ordered)
        note = pq.route_note
        pq.close()
        assert note.startswith("sort-based"), note
        g, w = rt.groupby(hs, None, [6], [A.count_star(), A.sum(1), A.min(2)], ordered), orc.groupby(os_, None, [6], [A.count_star(), A.sum(1), A.min(2)], ordered)
        assert [[k.value for k in r.keys] for r in g] == [[k.value for k in r.keys] for r in w], ordered
        for a, b in zip(g, w):
            assert_values(a.values, b.values, "sort-based GROUP BY over row ids with gaps")
    # dense ids from 0 keep nothing; ids that do not ascend are refused
    hd = rt.HipTable(1, chunks)
    hd.append_column(1, abi.DT_INT64, i64)
    hd.set_row_ids(np.arange(n, dtype=np.uint64))
    assert np.array_equal(rt.filter_row_ids(hd, [F(1, O.LessThan(0))]), orc.filter_row_ids(ot, [F(1, O.LessThan(0))]))
    if n > 1:
        bad = ids.copy()
        bad[n // 2] = bad[n // 2 - 1]
        hb = rt.HipTable(1, chunks)
        hb.append_column(1, abi.DT_INT64, i64)
        with pytest.raises(abi.LlkvError) as e:
            hb.set_row_ids(bad)
        assert e.value.kind == "InvalidArgumentError" and "ascend" in e.value.message


def test_joins_over_tables_whose_row_ids_have_gaps(rt, orc, abi):
    """The joined RecordBatches carry no row ids: identical to the dense table's.  The index-pair form reports ids and
    cuts by position — handed back rather than mixed."""
    rng = np.random.default_rng(77)
    nl, nr = 30000, 9000
    lk = rng.integers(0, 12000, size=nl).astype(np.int64)
    rk = rng.permutation(12000)[:nr].astype(np.int64)
    lv = rng.integers(-1000, 1000, size=nl).astype(np.int64)
    rv = rng.normal(size=nr)
    hl, ol = stage_both(rt, orc, abi, [(1, abi.DT_INT64, lk), (2, abi.DT_INT64, lv)], [nl])
    hr, orr = stage_both(rt, orc, abi, [(1, abi.DT_INT64, rk), (2, abi.DT_FLOAT64, rv)], [nr])
    hl.set_row_ids(np.cumsum(rng.integers(1, 4, size=nl)).astype(np.uint64))
    hr.set_row_ids(np.cumsum(rng.integers(1, 4, size=nr)).astype(np.uint64))
    lc, rc = [(1, "k"), (2, "v")], [(1, "k"), (2, "w")]
    for jt in (abi.JOIN_INNER, abi.JOIN_LEFT):
        got = rt.join_stream_batches(hl, hr, [(1, 1)], lc, rc, join_type=jt, batch_size=4096)
        want = orc.hash_join_batches(ol, orr, [(1, 1)], lc, rc, join_type=jt, batch_size=4096)
        assert len(got) == len(want)
        for (gn, gc), (wn, wc) in zip(got, want):
            assert gn == wn and [len(c) for c in gc] == [len(c) for c in wc]
            assert gc == wc
    with pytest.raises(abi.LlkvError) as e:
        rt.join_stream(hl, hr, [(1, 1)])
    assert e.value.kind == "Unsupported" and "row ids" in e.value.message
    # the Cartesian product cuts both scans into 65 536-entry windows of the row-id LIST (llkv-scan/src/execute.rs via stream_row_ids:
    # by position, not by id value): 70 000 × 3 rows with ids far apart give the dense table's batches, in its order
    nl2, nr2 = 70000, 3
    a, b = rng.integers(0, 100, size=nl2).astype(np.int64), np.arange(nr2, dtype=np.int64)
    hl2, ol2 = stage_both(rt, orc, abi, [(1, abi.DT_INT64, a)], [nl2])
    hr2, or2 = stage_both(rt, orc, abi, [(1, abi.DT_INT64, b)], [nr2])
    hl2.set_row_ids((np.arange(nl2, dtype=np.uint64) * np.uint64(7)) + np.uint64(2**33))
    hr2.set_row_ids(np.array([5, 70000, 2**35], dtype=np.uint64))
    got = rt.join_stream_batches(hl2, hr2, [], [(1, "a")], [(1, "b")], join_type=abi.JOIN_INNER)
    want = orc.hash_join_batches(ol2, or2, [], [(1, "a")], [(1, "b")], join_type=abi.JOIN_INNER)
    assert [gn for gn, _ in got] == [wn for wn, _ in want] and [gc for _, gc in got] == [wc for _, wc in want]


@pytest.mark.parametrize("chunks", [[9], [4096, 4097, 5], [65536, 70000]])
def test_expression_compares_match_oracle(rt, orc, abi, chunks):
    """Expr::Compare fused into the scan (Cmp node): every common-type class of get_common_type, totalOrder on
    floats, column ⋈ literal through the leaf route, NOT over the determined rows, inside AND / OR, feeding
    row ids, aggregates and GROUP BY."""
    rng = np.random.default_rng(sum(chunks))
    n = sum(chunks)
    i64 = rng.integers(-50, 50, size=n).astype(np.int64)
    f64 = rng.integers(-50, 50, size=n).astype(np.float64)
    f64[rng.random(n) < 0.05] = np.nan
    f64[rng.random(n) < 0.05] = -0.0
    f64[rng.random(n) < 0.05] = 0.0
    f64b = f64.copy()
    rng.shuffle(f64b)
    i32 = rng.integers(-50, 50, size=n).astype(np.int32)
    u64 = rng.integers(0, 60, size=n).astype(np.uint64)
    u64[rng.random(n) < 0.05] = 2**64 - 1
    u64[rng.random(n) < 0.05] = 2**53 + 1
    u32 = rng.integers(0, 60, size=n).astype(np.uint32)
    wide = rng.integers(2**53 - 2, 2**53 + 3, size=n).astype(np.int64)
    keys = np.array([ord("a"), ord("b"), ord("c")], dtype=np.uint8)[rng.integers(0, 3, size=n)]
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT64, i64), (2, abi.DT_FLOAT64, f64), (3, abi.DT_INT32, i32), (4, abi.DT_UINT64, u64),
                                       (5, abi.DT_UINT32, u32), (6, abi.DT_FLOAT64, f64b), (7, abi.DT_INT64, wide), (8, abi.DT_UTF8, keys)], chunks)
    E, col, A, F, O = abi.Expr, abi.col, abi.AggregateSpec, abi.Filter, abi.Operator
    ops = [abi.CMP_EQ, abi.CMP_NOT_EQ, abi.CMP_LT, abi.CMP_LT_EQ, abi.CMP_GT, abi.CMP_GT_EQ]
    sides = [(col(2), col(6)),            # Float64 totalOrder: NaN, ±0.0
             (col(1), col(3)),            # Int64 ⋈ Int32
             (col(3), col(5)),            # Int32 ⋈ UInt32 → Int64
             (col(4), col(5)),            # UInt64 ⋈ UInt32 → UInt64
             (col(7), col(4)),            # Int64 ⋈ UInt64 → Float64 (2^53 ± 1 collide)
             (col(5), col(1)),            # UInt32 ⋈ Int64 → Float64
             (col(4) + col(3), 40),       # the reference's own shape
             (col(1) * 2 - col(3), col(2)),
             (col(1), col(2) * 0.5),
             (col(2), 0.0), (0.0, col(2)), (col(3), 7), (7, col(5))]  # column ⋈ literal: leaf route unless <>
    for l, r in sides:
        for op in ops:
            e = E.compare(l, op, r)
            want = orc.filter_row_ids(ot, e)
            assert np.array_equal(rt.filter_row_ids(ht, e), want), (l.tokens if hasattr(l, "tokens") else l, op)
            ne = E.not_(e)
            assert np.array_equal(rt.filter_row_ids(ht, ne), orc.filter_row_ids(ot, ne))
    mixed = E.any_of([E.all_of([E.compare(col(1) + col(3), abi.CMP_GT, col(2)), F(3, O.LessThan(10))]), E.not_(E.compare(col(4), abi.CMP_LT_EQ, col(5)))])
    aggs = [A.count_star(), A.sum(1), A.min(1), A.max(2), A.sum(col(1) * col(3))]
    assert_values(rt.aggregate(ht, mixed, aggs), orc.aggregate(ot, mixed, aggs), "compare/aggregate")
    got, want = rt.groupby(ht, mixed, [8], [A.count_star(), A.sum(1)], True), orc.groupby(ot, mixed, [8], [A.count_star(), A.sum(1)], True)
    assert [[k.value for k in r.keys] for r in got] == [[k.value for k in r.keys] for r in want]
    for g, w in zip(got, want):
        assert_values(g.values, w.values, "compare/groupby")
    batches_g = rt.scan_stream(ht, [1, col(1) + col(3)], mixed, include_row_ids=True)
    batches_o = orc.scan_stream(ot, [1, col(1) + col(3)], mixed, include_nulls=True, include_row_ids=True)
    assert [b[1] for b in batches_g] == [b[1] for b in batches_o] and [b[0] for b in batches_g] == [b[0] for b in batches_o]
    # checked arithmetic inside a compare side fails the scan even when another conjunct rejects the row
    boom = E.all_of([E.compare(col(7) * 2**12, abi.CMP_GT, col(1)), F(3, O.Equals(10**6))])
    for run in (lambda: rt.filter_row_ids(ht, boom), lambda: rt.aggregate(ht, boom, [A.count_star()]),
                lambda: rt.groupby(ht, boom, [8], [A.count_star()], True), lambda: rt.scan_stream(ht, [1], boom)):
        with pytest.raises(abi.LlkvError) as e:
            run()
        assert e.value.kind == "Internal" and "overflow" in e.value.message.lower()
    with pytest.raises(abi.LlkvError) as e:
        orc.filter_row_ids(ot, boom)
    assert e.value.kind == "Internal" and "overflow" in e.value.message.lower()


@pytest.mark.parametrize("case", [c for c in TABLE["cases"] if c["table"] in NULLABLE_TABLES], ids=lambda c: c["name"])
def test_reference_include_nulls_known_answers(rt, abi, case):
    """test_scan_stream_include_nulls_toggle (table.rs:2357-2486) and the gather NULL-policy tests of llkv-column-map
    (gather_rows_policy_tests.rs, gather_multi_tests.rs) through llkv_hip_scan_stream: a NULL cell is staged as a
    validity bit, DropNulls drops the rows whose projected columns are all NULL."""
    from conftest import build_predicate
    tdef = TABLE["tables"][case["table"]]
    ht = rt.HipTable(1, [tdef["rows"]])
    for c in tdef["columns"]:
        dt = DTYPES[c["dtype"]]
        valid = [v is not None for v in c["values"]]
        ht.append_column(c["field_id"], dt, np.array([0 if v is None else v for v in c["values"]], dtype=abi.NUMPY_OF_DTYPE[dt]),
                         valid=None if all(valid) else valid)
    batches = rt.scan_stream(ht, case["project"], build_predicate(abi, case["predicate"]), include_nulls=case.get("include_nulls", False))
    cols = [[] for _ in case["project"]]
    for bcols, _ in batches:
        for i, c in enumerate(bcols):
            cols[i].extend(c)
    assert cols == case["expect"]


@pytest.mark.parametrize("chunks", [[11], [4096, 4097, 5], [65536, 70000]])
def test_null_cells_match_oracle(rt, orc, abi, chunks):
    """NULL cells (validity masks in HBM): three-valued predicates with domain-relative NOT, IS [NOT] NULL,
    compares, NULL-skipping accumulators with their own row counts, GROUP BY arguments, DropNulls / IncludeNulls
    scans with validity bitmaps on the way out."""
    rng = np.random.default_rng(7 + len(chunks))
    n = sum(chunks)
    i64 = rng.integers(-100, 100, size=n).astype(np.int64)
    f64 = rng.integers(-100, 100, size=n).astype(np.float64) / 4
    f64[rng.random(n) < 0.02] = np.nan
    i32 = rng.integers(-100, 100, size=n).astype(np.int32)
    dense = rng.integers(-2**40, 2**40, size=n).astype(np.int64)
    keys = np.array([ord("a"), ord("b"), ord("c")], dtype=np.uint8)[rng.integers(0, 3, size=n)]
    tags = [("x", "y", "zz")[k] for k in rng.integers(0, 3, size=n)]
    v1, v2, v3, v6 = rng.random(n) > 0.2, rng.random(n) > 0.1, rng.random(n) > 0.5, rng.random(n) > 0.3
    v1[:3] = False  # leading NULLs: "first row" of MIN/MAX must be the first non-NULL one
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT64, i64, v1), (2, abi.DT_FLOAT64, f64, v2), (3, abi.DT_INT32, i32, v3),
                                       (4, abi.DT_INT64, dense), (5, abi.DT_UTF8, keys), (6, abi.DT_UTF8, tags, v6),
                                       (7, abi.DT_DATE32, (i32 % 5).astype(np.int32) - 2, v3)], chunks)
    E, F, O, B, A, col = abi.Expr, abi.Filter, abi.Operator, abi.Bound, abi.AggregateSpec, abi.col
    preds = [None,
             [F(1, O.LessThan(10))],
             E.pred(F(1, O.IsNull)), E.pred(F(2, O.IsNotNull)), E.not_(F(1, O.IsNull)), E.not_(F(6, O.IsNotNull)),
             E.not_(F(1, O.LessThan(10))),
             E.not_(E.all_of([F(1, O.LessThan(10)), F(2, O.GreaterThan(-3.0))])),
             E.not_(E.any_of([F(1, O.LessThan(10)), F(3, O.In([1, 2, 3, 4, 5]))])),
             E.not_(E.any_of([F(1, O.LessThan(10)), F(4, O.GreaterThan(0))])),
             E.any_of([E.all_of([F(1, O.GreaterThanOrEquals(0)), E.not_(F(2, O.Range(B.Included(-5.0), B.Excluded(5.0))))]), E.pred(F(3, O.IsNull))]),
             E.pred(F(1, O.Range(B.Unbounded, B.Unbounded))), E.not_(F(1, O.Range(B.Unbounded, B.Unbounded))),
             E.compare(col(1), abi.CMP_GT, col(2)), E.not_(E.compare(col(1) + col(3), abi.CMP_LT_EQ, col(4))),
             E.all_of([E.compare(col(2), abi.CMP_NOT_EQ, 0.0), E.pred(F(6, O.Equals("zz")))]),
             E.not_(E.not_(F(2, O.GreaterThan(0.0)))),
             # ordering predicates on a dictionary-coded Utf8 column: the set of qualifying codes (str::cmp order)
             [F(6, O.GreaterThan("x"))], E.not_(F(6, O.Range(B.Included("y"), B.Unbounded))), [F(6, O.LessThanOrEquals("y")), F(5, O.GreaterThanOrEquals("b"))],
             E.pred(F(6, O.Range(B.Excluded("x"), B.Excluded("zz")))), [F(6, O.LessThan("a"))],
             [F(6, O.StartsWith("z"))], E.not_(F(6, O.EndsWith("Z", False))), E.any_of([F(6, O.Contains("y")), F(5, O.Contains("B", False))])]
    aggs = [A.count_star(), A.count(1), A.count_nulls(2), A.sum(1), A.avg(1), A.min(1), A.max(1), A.min(2), A.max(2), A.total(2), A.avg(2),
            A.sum(col(1) * col(2)), A.sum(col(1) * 3 - col(3)), A.sum(4), A.count(col(1) + col(3)), A.count(6)]
    gaggs = [A.count_star(), A.count(1), A.sum(1), A.avg(2), A.min(1), A.max(2), A.sum(col(1) * col(2)), A.sum(4)]
    for i, p in enumerate(preds):
        want = orc.filter_row_ids(ot, p)
        assert np.array_equal(rt.filter_row_ids(ht, p), want), i
        assert_values(rt.aggregate(ht, p, aggs), orc.aggregate(ot, p, aggs), f"pred {i}")
        got, exp = rt.groupby(ht, p, [5], gaggs, True), orc.groupby(ot, p, [5], gaggs, True)
        assert [[k.value for k in r.keys] for r in got] == [[k.value for k in r.keys] for r in exp]
        for g, w in zip(got, exp):
            assert_values(g.values, w.values, f"groupby pred {i}")
    # NULL is a group of its own: Utf8 and integer keys with NULL cells, alone and combined
    for keyset in ([6], [7], [6, 5], [7, 6]):
        for order in (True, False):
            ga = gaggs[:4] if len(keyset) == 1 else gaggs[:2]  # the dense LDS image holds ≤ 79 lanes
            got, exp = rt.groupby(ht, preds[1], keyset, ga, order), orc.groupby(ot, preds[1], keyset, ga, order)
            assert [[k.value for k in r.keys] for r in got] == [[k.value for k in r.keys] for r in exp], (keyset, order)
            assert any(k.value is None for r in got for k in r.keys)
            for g, w in zip(got, exp):
                assert_values(g.values, w.values, f"null keys {keyset}")
    if n < 20000:
        for p in (preds[0], preds[1], preds[10]):
            for projs in ([1], [1, 2], [2, col(1) * 2 + col(3)], [6], [6, 3], [4, 1]):
                for inc in (False, True):
                    got = rt.scan_stream(ht, projs, p, include_nulls=inc, include_row_ids=True)
                    want = orc.scan_stream(ot, projs, p, include_nulls=inc, include_row_ids=True)
                    assert [b[1] for b in got] == [b[1] for b in want], (projs, inc)
                    for (gc, _), (wc, _) in zip(got, want):
                        for a, b in zip(gc, wc):
                            assert len(a) == len(b)
                            assert all((x == y) or (isinstance(x, float) and isinstance(y, float) and math.isnan(x) and math.isnan(y)) for x, y in zip(a, b)), (projs, inc)
    # an arithmetic overflow under a NULL is not an error (arrow's checked kernels skip NULL slots) …
    big = np.full(n, 2**62, dtype=np.int64)
    ok = np.zeros(n, dtype=bool)
    ht2, ot2 = stage_both(rt, orc, abi, [(1, abi.DT_INT64, big, ok), (2, abi.DT_INT64, np.ones(n, dtype=np.int64))], chunks)
    for t, m in ((ht2, rt), (ot2, orc)):
        r = m.aggregate(t, E.not_(E.compare(col(1) * 4, abi.CMP_GT, col(2))), [A.sum(col(1) * 4), A.count_star(), A.count(1)])
        assert [x.value for x in r] == [None, 0, 0]
        r = m.aggregate(t, None, [A.sum(col(1) * 4), A.count_star(), A.count(1)])
        assert [x.value for x in r] == [None, n, 0]
    # … but it is one as soon as a single such cell is present
    ok[n // 2] = True
    ht3, ot3 = stage_both(rt, orc, abi, [(1, abi.DT_INT64, big, ok), (2, abi.DT_INT64, np.ones(n, dtype=np.int64))], chunks)
    for t, m in ((ht3, rt), (ot3, orc)):
        with pytest.raises(abi.LlkvError) as e:
            m.aggregate(t, None, [A.sum(col(1) * 4)])
        assert e.value.kind == "Internal" and "overflow" in e.value.message.lower()
        with pytest.raises(abi.LlkvError) as e:
            m.filter_row_ids(t, E.compare(col(1) * 4, abi.CMP_GT, col(2)))
        assert e.value.kind == "Internal" and "overflow" in e.value.message.lower()


@pytest.mark.parametrize("chunks", [[5], [4096, 4097, 3], [65536, 40000]])
def test_decimal128_accumulators_match_oracle(rt, orc, abi, chunks):
    """Decimal128 columns (staged narrowed to 64 bits): exact SUM / TOTAL / AVG (half away from zero) / MIN /
    MAX / COUNT with and without NULL cells, filters on other columns, GROUP BY, projection back to 16-byte
    values; the reference's refusal to filter a Decimal128 column."""
    rng = np.random.default_rng(len(chunks))
    n = sum(chunks)
    money = [int(v) for v in rng.integers(-10**13, 10**13, size=n)]          # DECIMAL(15,2)
    big = [int(v) for v in rng.integers(-2**62, 2**62, size=n)]               # sums leave i64, stay far inside i128
    flag = rng.integers(0, 4, size=n).astype(np.int64)
    keys = np.array([ord("a"), ord("b"), ord("c")], dtype=np.uint8)[rng.integers(0, 3, size=n)]
    valid = rng.random(n) > 0.2
    ht = rt.HipTable(1, chunks)
    ht.append_decimal128_column(1, 15, 2, money)
    ht.append_decimal128_column(2, 38, 4, big, valid=valid)
    ht.append_column(3, abi.DT_INT64, flag)
    ht.append_utf8_column(4, keys)
    ot = orc.OracleTable(n)
    ot.add(1, abi.DT_DECIMAL128, money, precision=15, scale=2)
    ot.add(2, abi.DT_DECIMAL128, big, list(valid), precision=38, scale=4)
    ot.add(3, abi.DT_INT64, flag).add(4, abi.DT_UTF8, keys)
    A, F, O, E = abi.AggregateSpec, abi.Filter, abi.Operator, abi.Expr
    aggs = [A.sum(1), A.avg(1), A.min(1), A.max(1), A.total(1), A.count(1), A.sum(2), A.avg(2), A.min(2), A.max(2), A.total(2), A.count(2), A.count_nulls(2), A.count_star()]
    for pred in (None, [F(3, O.Equals(1))], [F(3, O.GreaterThan(7))], E.not_(F(3, O.In([0, 2])))):
        got, want = rt.aggregate(ht, pred, aggs), orc.aggregate(ot, pred, aggs)
        assert got == want, pred  # dataclass equality: dtype, NULL-ness, raw i128, precision and scale
        g, w = rt.groupby(ht, pred, [4], aggs, True), orc.groupby(ot, pred, [4], aggs, True)
        assert [[k.value for k in r.keys] for r in g] == [[k.value for k in r.keys] for r in w]
        for a, b in zip(g, w):
            for i, (x, y) in enumerate(zip(a.values, b.values)):
                assert x == y, (pred, a.keys[0].value, i, x, y)
    if n < 20000:
        for inc in (False, True):
            got = rt.scan_stream(ht, [2, 1, 3], [F(3, O.LessThan(2))], include_nulls=inc, include_row_ids=True)
            want = orc.scan_stream(ot, [2, 1, 3], [F(3, O.LessThan(2))], include_nulls=inc, include_row_ids=True)
            assert got == want
    for m, t in ((rt, ht), (orc, ot)):
        with pytest.raises(abi.LlkvError) as e:
            m.aggregate(t, [F(1, O.LessThan(5))], [A.count_star()])
        assert e.value.kind == "Internal" and e.value.message.endswith("Filtering on type Decimal128(15, 2) is not supported")
        with pytest.raises(abi.LlkvError) as e:
            m.groupby(t, None, [1], [A.count_star()], True)
        assert e.value.kind == "InvalidArgumentError"


@pytest.mark.parametrize("chunks", [[7], [4096, 4097, 3], [65536, 9000]])
def test_wide_decimal128_min_max_over_a_span_below_64_bits(rt, orc, abi, chunks):
    """MIN / MAX over Decimal128 values beyond 64 bits (llkv-aggregate/src/lib.rs:1332-1352,1400-1420: i128 min / max of the
    non-NULL rows): when the column's values span less than 2^64 — a staging statistic — one MAX_U64 lane over v − min(column)
    (MAX) or max(column) − v (MIN) carries them (`MaxWideDelta`) and the host adds the base back in i128.  Ungrouped and on every
    GROUP BY route, with NULL cells and predicates, beside the limb sums of the same column; raw i128 equality with the oracle.
    A column that spans more stays `Unsupported`."""
    rng = np.random.default_rng(77 + len(chunks))
    n = sum(chunks)
    up = [10**30 + int(d) for d in rng.integers(-2**62, 2**62, size=n)]
    down = [-(10**33) - int(d) for d in rng.integers(0, 2**63, size=n)]
    up[n // 3], down[n // 4] = 10**30 + 2**62 + 5, -(10**33) - 2**63 - 11  # the extremes sit in known rows (still a span below 2^64)
    valid = rng.random(n) > 0.25
    flag = rng.integers(0, 4, size=n).astype(np.int64)
    keys = np.array([ord("a"), ord("b"), ord("c")], dtype=np.uint8)[rng.integers(0, 3, size=n)]
    day = rng.integers(9000, 9400, size=n).astype(np.int32)
    sparse = (rng.integers(0, 300, size=n) * 1_000_003).astype(np.int64)
    ht = rt.HipTable(1, chunks)
    ht.append_decimal128_column(1, 38, 4, up, valid=valid)
    ht.append_decimal128_column(2, 38, 0, down)
    ht.append_column(3, abi.DT_INT64, flag)
    ht.append_utf8_column(4, keys)
    ht.append_column(5, abi.DT_DATE32, day)
    ht.append_column(6, abi.DT_INT64, sparse)
    ot = orc.OracleTable(n)
    ot.add(1, abi.DT_DECIMAL128, up, list(valid), precision=38, scale=4)
    ot.add(2, abi.DT_DECIMAL128, down, precision=38, scale=0)
    ot.add(3, abi.DT_INT64, flag).add(4, abi.DT_UTF8, keys).add(5, abi.DT_DATE32, day).add(6, abi.DT_INT64, sparse)
    A, F, O = abi.AggregateSpec, abi.Filter, abi.Operator
    aggs = [A.min(1), A.max(1), A.min(2), A.max(2), A.count(1), A.sum(1), A.avg(2), A.count_star()]
    for pred in (None, [F(3, O.Equals(1))], [F(3, O.GreaterThan(7))]):
        got, want = rt.aggregate(ht, pred, aggs), orc.aggregate(ot, pred, aggs)
        assert got == want, pred
        for key_fields, order in (([4], True), ([5], False), ([6], True), ([4, 5], False)):
            g, w = rt.groupby(ht, pred, key_fields, aggs, order), orc.groupby(ot, pred, key_fields, aggs, order)
            assert [[k.value for k in r.keys] for r in g] == [[k.value for k in r.keys] for r in w], (pred, key_fields)
            for a, b in zip(g, w):
                assert a.values == b.values, (pred, key_fields, [k.value for k in a.keys])
    got = rt.aggregate(ht, None, [A.max(1), A.min(2)])
    assert got[0].value == max(v for v, ok in zip(up, valid) if ok) and got[1].value == min(down)
    far = rt.HipTable(2, [3])
    far.append_decimal128_column(1, 38, 0, [10**30, 10**30 + 2**64, 7])
    for bad in ([A.min(1)], [A.max(1)]):
        with pytest.raises(abi.LlkvError) as e:
            rt.aggregate(far, None, bad)
        assert e.value.kind == "Unsupported", bad


@pytest.mark.parametrize("chunks", [[7], [4096, 4097, 3], [65536, 9000]])
def test_wide_decimal128_sums_match_oracle(rt, orc, abi, chunks):
    """Decimal128 columns with values beyond 64 bits (staged as low and high halves): SUM / TOTAL / AVG (i128, half away
    from zero) and the counts equal the oracle's raw i128 results exactly — ungrouped, grouped by a small key (per-thread
    accumulators), by a date (shared-image kernel) and by a sparse key (sort-based route), with NULL cells and
    predicates on other columns; a scan passes the column through as 16-byte values.  Everything else over such a column
    stays on the caller's route (`Unsupported`), and so does a sum whose prefixes could leave i128 (the reference's check is order dependent)."""
    rng = np.random.default_rng(50 + len(chunks))
    n = sum(chunks)
    wide = [int(a) * 2**41 + int(b) for a, b in zip(rng.integers(-2**62, 2**62, size=n), rng.integers(0, 2**41, size=n))]  # |v| < 2^103
    wide[0], wide[n // 2], wide[-1] = 10**30, -(10**30) - 7, 5
    valid = rng.random(n) > 0.2
    flag = rng.integers(0, 4, size=n).astype(np.int64)
    keys = np.array([ord("a"), ord("b"), ord("c")], dtype=np.uint8)[rng.integers(0, 3, size=n)]
    day = rng.integers(9000, 9400, size=n).astype(np.int32)
    sparse = (rng.integers(0, 500, size=n) * 1_000_003).astype(np.int64)
    ht = rt.HipTable(1, chunks)
    ht.append_decimal128_column(1, 38, 4, wide, valid=valid)
    ht.append_decimal128_column(2, 38, 0, wide)
    ht.append_column(3, abi.DT_INT64, flag)
    ht.append_utf8_column(4, keys)
    ht.append_column(5, abi.DT_DATE32, day)
    ht.append_column(6, abi.DT_INT64, sparse)
    ot = orc.OracleTable(n)
    ot.add(1, abi.DT_DECIMAL128, wide, list(valid), precision=38, scale=4)
    ot.add(2, abi.DT_DECIMAL128, wide, precision=38, scale=0)
    ot.add(3, abi.DT_INT64, flag).add(4, abi.DT_UTF8, keys).add(5, abi.DT_DATE32, day).add(6, abi.DT_INT64, sparse)
    A, F, O, E = abi.AggregateSpec, abi.Filter, abi.Operator, abi.Expr
    aggs = [A.sum(1), A.avg(1), A.total(1), A.count(1), A.count_nulls(1), A.sum(2), A.avg(2), A.total(2), A.count_star(), A.sum(3)]
    for pred in (None, [F(3, O.Equals(1))], [F(3, O.GreaterThan(7))], E.not_(F(3, O.In([0, 2])))):
        got, want = rt.aggregate(ht, pred, aggs), orc.aggregate(ot, pred, aggs)
        assert got == want, pred  # dataclass equality: dtype, NULL-ness, raw i128, precision and scale
        for key_fields, order in (([4], True), ([5], False), ([6], True), ([4, 5], False)):
            g, w = rt.groupby(ht, pred, key_fields, aggs, order), orc.groupby(ot, pred, key_fields, aggs, order)
            assert [[k.value for k in r.keys] for r in g] == [[k.value for k in r.keys] for r in w], (pred, key_fields)
            for a, b in zip(g, w):
                assert a.values == b.values, (pred, key_fields, [k.value for k in a.keys])
    # what does not read the values works as for any column; what would need them in another form is handed back
    for bad in ([A.min(1)], [A.max(2)], [A.sum(abi.col(2) * 2)]):
        with pytest.raises(abi.LlkvError) as e:
            rt.aggregate(ht, None, bad)
        assert e.value.kind == "Unsupported", bad
    if n < 20000:  # the column itself passes through a scan as the 16-byte values it was staged from
        for inc in (False, True):
            got = rt.scan_stream(ht, [2, 1, 3], [F(3, O.LessThan(2))], include_nulls=inc, include_row_ids=True)
            want = orc.scan_stream(ot, [2, 1, 3], [F(3, O.LessThan(2))], include_nulls=inc, include_row_ids=True)
            assert got == want
    with pytest.raises(abi.LlkvError) as e:
        rt.scan_stream(ht, [abi.col(2) + 1], None)
    assert e.value.kind == "Unsupported"
    with pytest.raises(abi.LlkvError) as e:
        rt.join_stream(ht, ht, [(2, 2, False)], 0, 4096)
    assert e.value.kind == "Unsupported"
    for m, t in ((rt, ht), (orc, ot)):  # the reference does not filter Decimal128 columns at all
        with pytest.raises(abi.LlkvError) as e:
            m.aggregate(t, [F(2, O.LessThan(5))], [A.count_star()])
        assert e.value.kind == "Internal" and e.value.message.endswith("Filtering on type Decimal128(38, 0) is not supported")
    # rows · max|v| beyond i128: a prefix of the reference's checked_add chain may overflow although the total fits
    near = rt.HipTable(2, [4])
    near.append_decimal128_column(1, 38, 0, [10**38 - 1, -(10**38 - 1), 10**38 - 1, 3])
    with pytest.raises(abi.LlkvError) as e:
        rt.aggregate(near, None, [A.sum(1)])
    assert e.value.kind == "Unsupported" and "order dependent" in e.value.message
    assert [v.value for v in rt.aggregate(near, None, [A.count(1), A.count_star()])] == [4, 4]


@pytest.mark.parametrize("route", ["auto", "sort"])
@pytest.mark.parametrize("chunks", [[13], [4096, 4097, 5], [65536, 70000, 65536]])
def test_sort_based_group_by_matches_oracle(rt, orc, abi, chunks, route, monkeypatch):
    """GROUP BY shapes the per-thread accumulator kernel cannot hold — thousands of groups, sparse or unbounded integer
    keys, several keys, wide aggregate states: same groups, same first-appearance / key order, exact integer results,
    f64 sums within 1e-9.  `auto`: the shapes with statistics-bounded keys take the shared-image kernel, the others
    the sort-based route; `sort`: all of them take the sort-based route."""
    if route == "sort":
        monkeypatch.setenv("LLKV_HIP_GROUP_NO_IMAGE", "1")
    rng = np.random.default_rng(3 + len(chunks))
    n = sum(chunks)
    k_sparse = (rng.integers(0, 3000, size=n) * 1_000_003 - 10**9).astype(np.int64)   # ≤ 3000 groups, huge range
    k_date = rng.integers(8000, 8400, size=n).astype(np.int32)                          # 400 groups: range > 256
    k_u32 = rng.integers(0, 7, size=n).astype(np.uint32)
    k_tag = [("x", "y", "zz", "")[k] for k in rng.integers(0, 4, size=n)]
    i64 = rng.integers(-1000, 1000, size=n).astype(np.int64)
    f64 = rng.integers(-4000, 4000, size=n).astype(np.float64) / 8
    f64[rng.random(n) < 0.01] = np.nan
    vk, va = rng.random(n) > 0.1, rng.random(n) > 0.2
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT64, k_sparse), (2, abi.DT_DATE32, k_date, vk), (3, abi.DT_UINT32, k_u32), (4, abi.DT_UTF8, k_tag, vk),
                                       (5, abi.DT_INT64, i64, va), (6, abi.DT_FLOAT64, f64)], chunks)
    A, F, O, E, col = abi.AggregateSpec, abi.Filter, abi.Operator, abi.Expr, abi.col
    narrow = [A.count_star(), A.sum(5), A.max(6)]
    wide = [A.count_star(), A.count(5), A.count_nulls(5), A.sum(5), A.avg(5), A.min(5), A.max(5), A.total(5), A.sum(6), A.avg(6), A.min(6), A.max(6),
            A.sum(col(5) * col(6)), A.sum(col(5) * 3 - col(5)), A.total(6)]
    cases = [([1], narrow), ([2], narrow), ([1, 3], narrow), ([2, 4], narrow), ([4, 2, 3], narrow), ([3], wide), ([4], wide), ([2, 3], wide)]
    preds = [None, [F(5, O.GreaterThan(0))], E.not_(E.any_of([F(6, O.LessThan(0.0)), F(3, O.Equals(2))]))]
    for keys, aggs in cases:
        for pred in preds if len(chunks) < 3 else preds[:2]:
            for order in (True, False):
                got, exp = rt.groupby(ht, pred, keys, aggs, order), orc.groupby(ot, pred, keys, aggs, order)
                assert [[k.value for k in r.keys] for r in got] == [[k.value for k in r.keys] for r in exp], (keys, order)
                for g, w in zip(got, exp):
                    assert_values(g.values, w.values, f"sorted group by {keys}")
    # float and decimal keys are refused like the reference; no match → no groups
    with pytest.raises(abi.LlkvError) as e:
        rt.groupby(ht, None, [6, 1], narrow, True)
    assert e.value.kind == "InvalidArgumentError"
    assert rt.groupby(ht, [F(5, O.GreaterThan(10**6))], [1], wide, True) == []
    # an Int64 sum that leaves the range fails the query on both sides (PlanValue arguments go through f64 first)
    kinds = []
    for m, t in ((rt, ht), (orc, ot)):
        with pytest.raises(abi.LlkvError) as e:
            m.groupby(t, None, [3], [A.sum(col(1) * 2**40 * 2**20)] + wide, True)
        kinds.append((e.value.kind, e.value.message))
    # (when only a prefix of the chain can overflow the GPU route hands the query back instead of guessing)
    assert kinds[1][0] == "InvalidArgumentError" and "integer overflow" in kinds[1][1]
    assert kinds[0][0] in ("InvalidArgumentError", "Unsupported")


@pytest.mark.parametrize("route", ["partitioned", "sort"])
@pytest.mark.parametrize("chunks", [[17], [33001, 4096, 5]])
def test_partitioned_group_by_matches_oracle(rt, orc, abi, chunks, route, monkeypatch):
    """More dense groups than LDS-sized slices cover (here up to 1.2 M group ids: statistics-bounded integer keys, one or
    two of them, one with NULL cells): the partitioned route — count pass, scan, scatter pass, one LDS image per
    partition — returns the groups of the oracle in first-appearance order, integers exactly, f64 sums within 1e-9;
    with LLKV_HIP_GROUP_NO_PART=1 the sort-based route answers the same."""
    if route == "sort":
        monkeypatch.setenv("LLKV_HIP_GROUP_NO_PART", "1")
    rng = np.random.default_rng(41 + len(chunks))
    n = sum(chunks)
    k_big = rng.integers(-100_000, 200_000, size=n).astype(np.int64)      # 300 000 possible groups
    k_mid = rng.integers(0, 2_000, size=n).astype(np.int64)
    k_day = rng.integers(9000, 9600, size=n).astype(np.int32)              # × 2 000 = 1.2 M group ids
    i64 = rng.integers(-1000, 1000, size=n).astype(np.int64)
    f64 = rng.integers(1, 400_000, size=n).astype(np.float64) / 100
    g64 = rng.standard_normal(n) * 1e3
    vk, va = rng.random(n) > 0.1, rng.random(n) > 0.2
    k_tag = [("x", "y", "zz", "")[k] for k in rng.integers(0, 4, size=n)]     # a dictionary-coded key with NULL cells: × 300 001 ids
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT64, k_big), (2, abi.DT_INT64, k_mid, vk), (3, abi.DT_DATE32, k_day), (4, abi.DT_UTF8, k_tag, va), (5, abi.DT_INT64, i64, va),
                                       (6, abi.DT_FLOAT64, f64), (7, abi.DT_FLOAT64, g64)], chunks)
    A, F, O, col = abi.AggregateSpec, abi.Filter, abi.Operator, abi.col
    narrow = [A.count_star(), A.sum(5), A.sum(6)]
    wide = [A.count_star(), A.count(5), A.sum(5), A.avg(5), A.min(5), A.max(5), A.total(5), A.sum(6), A.avg(6), A.min(7), A.max(7), A.sum(col(6) * (10000 - col(6))),
            A.sum(7)]
    short = [A.count_star(), A.sum(1), A.sum(6)]  # records of ≤ 4 words over ≤ 512 partitions: the scatter writes whole lines (LLKV_HIP_PART_NO_LINES=1: runs)
    for keys, aggs in (([1], narrow), ([1], wide), ([2, 3], narrow), ([3, 2], wide), ([4, 1], narrow), ([1], narrow[:1]), ([1], short), ([1], short[:2]), ([2, 3], short)):  # (narrow[:1]: records of one word)
        for pred in (None, [F(5, O.GreaterThan(-500))]):
            pq = rt.PreparedQuery(ht, pred, aggs, keys, False)
            note = pq.route_note
            pq.close()
            assert note.startswith(route), note
            got, exp = rt.groupby(ht, pred, keys, aggs, False), orc.groupby(ot, pred, keys, aggs, False)
            assert [[k.value for k in r.keys] for r in got] == [[k.value for k in r.keys] for r in exp], (keys, route)
            for g, w in zip(got, exp):
                assert_values(g.values, w.values, f"{route} group by {keys}")
            if route == "partitioned" and len(aggs) <= 3:  # both forms of the scatter: the same records, so the same bits
                monkeypatch.setenv("LLKV_HIP_PART_NO_LINES", "1")
                runs = rt.groupby(ht, pred, keys, aggs, False)
                monkeypatch.delenv("LLKV_HIP_PART_NO_LINES")
                assert [(r.keys, r.values) for r in runs] == [(r.keys, r.values) for r in got], (keys, "line form vs runs")
    assert rt.groupby(ht, [F(5, O.GreaterThan(10**6))], [1], wide, False) == []
    # ORDER BY the keys: ascending group ids are ascending keys for integer keys without NULL cells; with NULL cells
    # (NULLS FIRST) or dictionary-coded strings the groups are sorted by their ranked id
    for keys, want in (([1], route), ([2, 3], route), ([4, 1], route)):
        pq = rt.PreparedQuery(ht, None, narrow, keys, True)
        assert pq.route_note.startswith(want), pq.route_note
        pq.close()
        got, exp = rt.groupby(ht, None, keys, narrow, True), orc.groupby(ot, None, keys, narrow, True)
        assert [[k.value for k in r.keys] for r in got] == [[k.value for k in r.keys] for r in exp], (keys, route)
        for g, w in zip(got, exp):
            assert_values(g.values, w.values, f"{route} group by {keys} in key order")
        if route == "partitioned" and keys == [1]:  # one integer key in key order: the copy-out of a range of partitions runs beside the next range's reduction
            monkeypatch.setenv("LLKV_HIP_PART_NO_OVERLAP", "1")
            serial = rt.groupby(ht, None, keys, narrow, True)
            monkeypatch.delenv("LLKV_HIP_PART_NO_OVERLAP")
            assert [(r.keys, r.values) for r in serial] == [(r.keys, r.values) for r in got], "ranges vs one reduction + one copy-out"


@pytest.mark.parametrize("chunks", [[11], [4096, 4097, 5], [65536, 30000]])
def test_aggregates_over_utf8_and_boolean_inputs_coerce_like_the_reference(rt, orc, abi, chunks):
    """SQLite-style coercion of aggregate inputs (validate_aggregate_type llkv-executor/src/lib.rs:5946-5988 →
    array_value_to_numeric llkv-aggregate/src/lib.rs:400-449): SUM / AVG / TOTAL / MIN / MAX over a Utf8 column use
    Float64 accumulators over the parsed strings (non-numbers count as 0), over a Boolean column over 0 / 1; NULL cells
    are skipped.  On the GPU the dictionary is parsed once and the codes index its numeric image.  Ungrouped, grouped
    by a small key (per-thread accumulators) and by a date (shared-image kernel, exact sums); Date32 inputs are left to
    the caller (the reference fails at the first non-NULL row)."""
    rng = np.random.default_rng(17 + len(chunks))
    n = sum(chunks)
    words = ["12", " 3.5 ", "abc", "", "1e2", "0x10", "-7.25", ".5", "1.", "+4", "0.125", "1000000.5"]
    txt = [words[i] for i in rng.integers(0, len(words), size=n)]
    boo = rng.integers(0, 2, size=n).astype(np.uint8)
    key = np.array([ord("p"), ord("q"), ord("r")], dtype=np.uint8)[rng.integers(0, 3, size=n)]
    day = rng.integers(9000, 9700, size=n).astype(np.int32)
    vt, vb = rng.random(n) > 0.15, rng.random(n) > 0.1
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_UTF8, txt, vt), (2, abi.DT_BOOLEAN, boo, vb), (3, abi.DT_UTF8, key), (4, abi.DT_DATE32, day),
                                       (5, abi.DT_UTF8, txt)], chunks)
    A, F, O = abi.AggregateSpec, abi.Filter, abi.Operator
    aggs = [A.count_star(), A.sum(1), A.avg(1), A.total(1), A.min(1), A.max(1), A.count(1), A.sum(2), A.avg(2), A.total(2), A.min(2), A.max(2), A.sum(5), A.max(5)]
    for pred in (None, [F(4, O.GreaterThan(9300))]):
        assert_values(rt.aggregate(ht, pred, aggs), orc.aggregate(ot, pred, aggs), "coercion")
        for keys in ([3], [4]):
            pq = rt.PreparedQuery(ht, pred, aggs, keys, True)
            assert not pq.route_note.startswith("sort"), pq.route_note
            pq.close()
            got, want = rt.groupby(ht, pred, keys, aggs, True), orc.groupby(ot, pred, keys, aggs, True)
            assert [r.keys[0].value for r in got] == [r.keys[0].value for r in want]
            for x, y in zip(got, want):
                assert_values(x.values, y.values, f"coercion/groupby {keys}")
    with pytest.raises(abi.LlkvError) as e:
        rt.aggregate(ht, None, [A.sum(4)])
    assert e.value.kind == "Unsupported"


def test_scan_stream_fails_at_the_window_whose_projection_failed(rt, abi):
    """A computed projection that fails (`% 0`) ends the stream at the 65 536-row window that holds the offending row:
    the windows before it have been delivered, that one and the later ones are not (the reference's arrow kernel
    fails the batch it is evaluating, llkv-scan/src/row_stream.rs)."""
    n = 300_000
    a = np.arange(n, dtype=np.int64)
    b = np.ones(n, dtype=np.int64)
    b[150_000] = 0  # third window
    ht = rt.HipTable(1, [n])
    ht.append_column(1, abi.DT_INT64, a)
    ht.append_column(2, abi.DT_INT64, b)
    seen = []
    with pytest.raises(abi.LlkvError) as e:
        rt.scan_stream(ht, [1, abi.col(1) % abi.col(2)], None, consume=lambda bv: seen.append(int(bv.num_rows)))
    assert e.value.kind == "Internal" and "Divide by zero" in e.value.message
    assert seen == [65536, 65536]
    seen.clear()
    b[150_000] = 1
    ht2 = rt.HipTable(2, [n])
    ht2.append_column(1, abi.DT_INT64, a)
    ht2.append_column(2, abi.DT_INT64, b)
    rt.scan_stream(ht2, [1, abi.col(1) % abi.col(2)], None, consume=lambda bv: seen.append(int(bv.num_rows)))
    assert seen == [65536] * 4 + [n - 4 * 65536]


@pytest.mark.parametrize("chunks", [[1000], [65536, 65536, 30000], [300000, 300000, 300000, 123457]])
def test_exact_f64_sums_option_gives_the_correctly_rounded_sum(rt, orc, abi, chunks):
    """llkv_hip_set_exact_f64_sums: SUM / AVG / TOTAL over f64 arguments are the exact sum of the rows' values, rounded
    once — math.fsum of the same per-row doubles, bit for bit, ungrouped (register plans) and grouped (per-thread LDS
    columns), whatever the geometry; and within the contract's 1e-9 of the oracle's sequential chain."""
    rng = np.random.default_rng(len(chunks) * 7 + 1)
    n = sum(chunks)
    price = np.round(rng.uniform(900.0, 105000.0, size=n), 2)          # cents: not dyadic
    disc = rng.integers(0, 11, size=n).astype(np.float64) / 100.0
    tax = rng.integers(0, 9, size=n).astype(np.float64) / 100.0
    qty = rng.integers(1, 51, size=n).astype(np.int64)
    flag = np.array([ord("A"), ord("N"), ord("R")], dtype=np.uint8)[rng.integers(0, 3, size=n)]
    valid = rng.random(n) > 0.05
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_FLOAT64, price), (2, abi.DT_FLOAT64, disc), (3, abi.DT_FLOAT64, tax), (4, abi.DT_INT64, qty),
                                       (5, abi.DT_UTF8, flag), (6, abi.DT_FLOAT64, price, valid)], chunks)
    A, F, O, col = abi.AggregateSpec, abi.Filter, abi.Operator, abi.col
    dp = col(1) * (1 - col(2))
    ch = col(1) * (1 - col(2)) * (1 + col(3))
    aggs = [A.count_star(), A.sum(1), A.sum(dp), A.sum(ch), A.avg(2), A.total(1), A.sum(6), A.avg(6), A.sum(4)]
    v_dp, v_ch = price * (1 - disc), price * (1 - disc) * (1 + tax)
    pred = [F(4, O.LessThan(40))]
    keep = qty < 40

    def expect(m):
        k = int(m.sum())
        nv = int((m & valid).sum())
        return [k, math.fsum(price[m]), math.fsum(v_dp[m]), math.fsum(v_ch[m]), math.fsum(disc[m]) / k, math.fsum(price[m]), math.fsum(price[m & valid]),
                math.fsum(price[m & valid]) / nv, int(qty[m].sum())]

    rt.set_exact_f64_sums(True)
    try:
        got = rt.aggregate(ht, pred, aggs)
        assert [g.value for g in got] == expect(keep)
        assert_values(got, orc.aggregate(ot, pred, aggs), "exact vs chain")
        rows = rt.groupby(ht, pred, [5], aggs, True)
        want = orc.groupby(ot, pred, [5], aggs, True)
        assert [r.keys[0].value for r in rows] == ["A", "N", "R"][: len(rows)]
        for r, w, code in zip(rows, want, (ord("A"), ord("N"), ord("R"))):
            assert [v.value for v in r.values] == expect(keep & (flag == code)), chr(code)
            assert_values(r.values, w.values, "exact vs chain, grouped")
        q = rt.PreparedQuery(ht, pred, aggs, [5], True)
        assert "SumF64Q2<" in q.kernel_signature
        q.close()
        # an argument the statistics cannot bound away from zero has no exact grid: handed back while the option is on
        with pytest.raises(abi.LlkvError) as e:
            rt.aggregate(ht, None, [A.sum(col(1) - col(6))])
        assert e.value.kind == "Unsupported"
    finally:
        rt.set_exact_f64_sums(False)
    q = rt.PreparedQuery(ht, pred, aggs, [5], True)
    assert "SumF64Q2<" not in q.kernel_signature
    q.close()


@pytest.mark.parametrize("chunks", [[9], [4096, 4097, 5]])
def test_constant_subexpressions_fold_like_the_reference(rt, orc, abi, chunks):
    """ScalarEvaluator::simplify (llkv-compute/src/eval.rs:761-791) on the host, before the plan is typed: literal ⊕ literal
    folds through compute_binary's rules; a fold that errors is a plan neither side takes; x / −0.0 is IEEE."""
    rng = np.random.default_rng(5 + len(chunks))
    n = sum(chunks)
    a = rng.integers(-1000, 1000, size=n).astype(np.int64)
    f = rng.integers(-40, 40, size=n).astype(np.float64) / 4
    f[rng.random(n) < 0.05] = -0.0
    f[rng.random(n) < 0.05] = 0.0
    va = rng.random(n) > 0.1
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT64, a, va), (2, abi.DT_FLOAT64, f)], chunks)
    L, col, A, E = abi.ScalarExpr.literal, abi.col, abi.AggregateSpec, abi.Expr
    exprs = [col(1) * (L(2) + 3), col(1) + (L(7) / 2), col(1) * (L(1) / 4), col(2) * (L(1) / 4.0), (L(10) % 4) + col(1),
             (L(2) * 3 + 1) * col(2) - (L(1.5) - 0.25), (L(6) / 3) / col(1), col(1) - (L(-7) % 3), (L(1) + 1) * (L(2.5) * 2) + col(1),
             col(1) / col(2), (L(3) - 2) / col(2)]
    for part in (exprs[:6], exprs[6:]):  # (a scan takes up to 8 projections)
        got = rt.scan_stream(ht, [1] + part, None, include_nulls=True, include_row_ids=True)
        want = orc.scan_stream(ot, [1] + part, None, include_nulls=True, include_row_ids=True)
        assert [b[1] for b in got] == [b[1] for b in want]
        for (gc, _), (wc, _) in zip(got, want):
            for k, (x, y) in enumerate(zip(gc, wc)):
                assert len(x) == len(y)
                bad = [(p, q) for p, q in zip(x, y) if not ((p == q and type(p) is type(q)) or (isinstance(p, float) and isinstance(q, float) and math.isnan(p) and math.isnan(q)))]
                assert not bad, (k, bad[:3])
    aggs = [A.count_star()] + [A.sum(e) for e in exprs[:9]] + [A.count(exprs[9]), A.count(exprs[10]), A.min(exprs[10]), A.max(exprs[9])]
    for pred in (None, E.compare(col(1) * (L(2) + 3), abi.CMP_GT, L(10) * 10), E.compare((L(3) - 2) / col(2), abi.CMP_LT, 0.0)):
        assert np.array_equal(rt.filter_row_ids(ht, pred), orc.filter_row_ids(ot, pred))
        assert_values(rt.aggregate(ht, pred, aggs), orc.aggregate(ot, pred, aggs), "folded")
    for bad in (col(1) + (L(2**62) + 2**62), col(1) + (L(5) % 0), col(1) * (L(-2**63) / -1)):
        for m, t in ((rt, ht), (orc, ot)):
            with pytest.raises(abi.LlkvError) as e:
                m.aggregate(t, None, [A.sum(bad)])
            assert e.value.kind == "Unsupported"


@pytest.mark.parametrize("chunks", [[9], [4096, 4097, 5], [65536, 70000]])
def test_division_and_modulo_match_oracle(rt, orc, abi, chunks):
    """Divide leaves the fast numeric path: per-node typing, zeros of a divisor become NULLs, integer division
    truncates (compute_binary, llkv-compute/src/kernels.rs:99-177); `%` is arrow `rem` (zero → "Divide by zero");
    GROUP BY arguments follow the PlanValue rules (x / 0 and x % 0 → NULL).  Arithmetic errors are raised node by
    node, only where that node's operands are valid."""
    rng = np.random.default_rng(11 + len(chunks))
    n = sum(chunks)
    a = rng.integers(-1000, 1000, size=n).astype(np.int64)
    b = rng.integers(-3, 4, size=n).astype(np.int64)            # plenty of zeros
    nz = np.where(b == 0, 5, b).astype(np.int64)                # never zero
    f = rng.integers(-40, 40, size=n).astype(np.float64) / 4    # zeros, -0.0 below
    f[rng.random(n) < 0.02] = -0.0
    g = rng.integers(1, 9, size=n).astype(np.float64) / 2
    keys = np.array([ord("p"), ord("q")], dtype=np.uint8)[rng.integers(0, 2, size=n)]
    va = rng.random(n) > 0.15
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT64, a, va), (2, abi.DT_INT64, b), (3, abi.DT_INT64, nz), (4, abi.DT_FLOAT64, f),
                                       (5, abi.DT_FLOAT64, g), (6, abi.DT_UTF8, keys)], chunks)
    A, F, O, E, col = abi.AggregateSpec, abi.Filter, abi.Operator, abi.Expr, abi.col
    aggs = [A.count_star(), A.sum(col(1) / col(2)), A.count(col(1) / col(2)), A.count_nulls(col(1) / col(2)), A.avg(col(1) / col(2)), A.sum(col(1) % col(3)),
            A.sum((col(1) + col(3)) / col(4)), A.min(col(5) / col(4)), A.max((col(1) * 2) / (col(2) * col(3))), A.total(col(1) / 4.0), A.sum(col(4) % col(5)),
            A.sum(col(1) / col(3) * col(2))]
    for pred in (None, [F(3, O.GreaterThan(0))], E.compare(col(1) / col(2), abi.CMP_GT, col(3)), E.not_(E.compare(col(5) / col(4), abi.CMP_LT_EQ, 1.0))):
        assert np.array_equal(rt.filter_row_ids(ht, pred), orc.filter_row_ids(ot, pred))
        assert_values(rt.aggregate(ht, pred, aggs), orc.aggregate(ot, pred, aggs), "division")
    gaggs = [A.count_star(), A.sum(col(4) / col(5)), A.count(col(5) / col(4)), A.sum(col(1) % col(2)), A.count(col(1) % col(2)), A.avg(col(1) / 8.0), A.sum(col(4) % col(5))]
    got, want = rt.groupby(ht, None, [6], gaggs, True), orc.groupby(ot, None, [6], gaggs, True)
    assert [r.keys[0].value for r in got] == [r.keys[0].value for r in want]
    for x, y in zip(got, want):
        assert_values(x.values, y.values, "division/groupby")
    if n < 20000:
        projs = [1, col(1) / col(2), col(5) / col(4), col(1) % col(3), (col(1) + 1) / col(3)]
        for inc in (False, True):
            got = rt.scan_stream(ht, projs, [F(3, O.LessThan(3))], include_nulls=inc, include_row_ids=True)
            want = orc.scan_stream(ot, projs, [F(3, O.LessThan(3))], include_nulls=inc, include_row_ids=True)
            assert [b[1] for b in got] == [b[1] for b in want]
            for (gc, _), (wc, _) in zip(got, want):
                for x, y in zip(gc, wc):
                    assert all((p == q) or (isinstance(p, float) and isinstance(q, float) and math.isnan(p) and math.isnan(q)) for p, q in zip(x, y))
    # errors: `% 0`; i64::MIN / -1; a node's error counts wherever ITS operands are valid, whatever else is NULL
    for m, t in ((rt, ht), (orc, ot)):
        with pytest.raises(abi.LlkvError) as e:
            m.aggregate(t, None, [A.sum(col(3) % col(2))])
        assert e.value.kind == "Internal" and "Divide by zero" in e.value.message
    # Int / Int in a GROUP BY argument (PlanValue: truncating, x / 0 → NULL) — taken when the statistics exclude the one pair that
    # turns Float in the reference, i64::MIN / −1 (handed back otherwise: the table below)
    gi = [A.sum(col(1) / col(3)), A.count(col(1) / col(3)), A.min(col(1) / col(3)), A.sum(col(1) / 7), A.sum(col(3) / 0)]
    g, w = rt.groupby(ht, None, [6], gi, True), orc.groupby(ot, None, [6], gi, True)
    assert [[k.value for k in r.keys] for r in g] == [[k.value for k in r.keys] for r in w]
    for a, b in zip(g, w):
        assert_values(a.values, b.values, "Int / Int in GROUP BY arguments")
    mn = np.array([-2**63, 5, 7] + [1] * (n - 3), dtype=np.int64)
    m1 = np.array([-1, 0, 2] + [1] * (n - 3), dtype=np.int64)
    other = np.zeros(n, dtype=np.int64)
    ov = np.ones(n, dtype=bool)
    ov[0] = False
    ht2, ot2 = stage_both(rt, orc, abi, [(1, abi.DT_INT64, mn), (2, abi.DT_INT64, m1), (3, abi.DT_INT64, other, ov)], chunks)
    for m, t in ((rt, ht2), (orc, ot2)):
        with pytest.raises(abi.LlkvError) as e:
            m.aggregate(t, None, [A.sum(col(1) / col(2))])
        assert e.value.kind == "Internal" and "overflow" in e.value.message.lower()
        with pytest.raises(abi.LlkvError) as e:  # row 0: column 3 is NULL there, but the division node's operands are not
            m.aggregate(t, None, [A.sum(col(1) / col(2) + col(3))])
        assert e.value.kind == "Internal" and "overflow" in e.value.message.lower()
        r = m.aggregate(t, [F(2, O.GreaterThanOrEquals(0))], [A.sum(col(1) / col(2)), A.count(col(1) / col(2)), A.sum(col(1) % col(2) if False else col(1) / col(2) + col(3))])
        assert [x.value for x in r] == [3 + (n - 3), n - 2, 3 + (n - 3)]
    with pytest.raises(abi.LlkvError) as e:  # i64::MIN over a column that holds −1: the group's temp column could turn Float
        rt.groupby(ht2, None, [3], [A.sum(col(1) / col(2))], True)
    assert e.value.kind == "Unsupported"


@pytest.mark.parametrize("chunks", [[10], [4096, 4097, 5], [65536, 70000]])
def test_in_list_and_is_null_expressions_match_oracle(rt, orc, abi, chunks):
    """EvalOp::PushInList / PushIsNull over scalar expressions, fused into the scan like every other predicate."""
    rng = np.random.default_rng(21 + len(chunks))
    n = sum(chunks)
    a = rng.integers(-5, 6, size=n).astype(np.int64)
    b = rng.integers(-5, 6, size=n).astype(np.int64)
    f = rng.integers(-5, 6, size=n).astype(np.float64)
    f[rng.random(n) < 0.05] = np.nan
    f[rng.random(n) < 0.05] = -0.0
    u = rng.integers(0, 6, size=n).astype(np.uint64)
    va, vf = rng.random(n) > 0.2, rng.random(n) > 0.2
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT64, a, va), (2, abi.DT_INT64, b), (3, abi.DT_FLOAT64, f, vf), (4, abi.DT_UINT64, u)], chunks)
    E, F, O, A, col = abi.Expr, abi.Filter, abi.Operator, abi.AggregateSpec, abi.col
    preds = [E.in_list(col(1), [1, 2, 3]), E.in_list(col(1), [1, 2, 3], negated=True), E.not_(E.in_list(col(1), [col(2), 4])),
             E.in_list(col(2) * 2, [col(1), col(3), 4]), E.in_list(col(3), [float("nan"), 0.0, col(2)]), E.in_list(col(4), [col(2), 3], negated=True),
             E.in_list(col(2), [], negated=True), E.in_list(col(2), []),
             E.is_null(col(1) + col(3)), E.is_null(col(1) + col(3), negated=True), E.not_(E.is_null(col(1) * col(2))), E.is_null(col(2) / (col(2) - 1)),
             E.is_null(col(1) / col(2), negated=True), E.is_null(col(3)), E.any_of([E.is_null(col(1) - col(2)), E.all_of([E.in_list(col(2), [0, 1]), F(3, O.GreaterThan(0.0))])])]
    aggs = [A.count_star(), A.sum(2), A.count(1)]
    for i, p in enumerate(preds):
        assert np.array_equal(rt.filter_row_ids(ht, p), orc.filter_row_ids(ot, p)), i
        assert_values(rt.aggregate(ht, p, aggs), orc.aggregate(ot, p, aggs), f"pred {i}")


def _random_predicate(rng, abi, depth):
    """Random predicate tree over the columns of test_random_predicate_trees_match_oracle."""
    E, F, O, B, col = abi.Expr, abi.Filter, abi.Operator, abi.Bound, abi.col
    ints, flts = [1, 2, 5], [3, 4]   # field ids by type (1, 3 have NULL cells; 5 is Int32)

    def scalar(kind):
        base = col(int(rng.choice(ints if kind == "i" else flts)))
        r = rng.random()
        if r < 0.3:
            return base
        if r < 0.55:
            return base + (int(rng.integers(-3, 4)) if kind == "i" else float(rng.integers(-3, 4)))
        if r < 0.75:
            return base * col(int(rng.choice(ints)))
        if r < 0.9:
            return base - col(int(rng.choice(ints + flts)))
        return base / col(int(rng.choice(ints + flts)))

    def leaf():
        r = rng.random()
        if r < 0.35:
            fid = int(rng.choice(ints + flts))
            v = int(rng.integers(-4, 5)) if fid in ints else float(rng.integers(-4, 5))
            k = rng.integers(0, 7)
            if k == 0: return E.pred(F(fid, O.Equals(v)))
            if k == 1: return E.pred(F(fid, O.LessThan(v)))
            if k == 2: return E.pred(F(fid, O.GreaterThanOrEquals(v)))
            if k == 3: return E.pred(F(fid, O.Range(B.Included(v), B.Excluded(v + 3))))
            if k == 4: return E.pred(F(fid, O.In([v, v + 1, v + 5])))
            if k == 5: return E.pred(F(fid, O.IsNull))
            return E.pred(F(fid, O.IsNotNull))
        if r < 0.6:
            ops = [abi.CMP_EQ, abi.CMP_NOT_EQ, abi.CMP_LT, abi.CMP_LT_EQ, abi.CMP_GT, abi.CMP_GT_EQ]
            return E.compare(scalar(rng.choice(["i", "f"])), int(rng.choice(ops)), scalar(rng.choice(["i", "f"])))
        if r < 0.8:
            k = rng.choice(["i", "f"])
            items = [scalar(k) if rng.random() < 0.3 else (int(rng.integers(-4, 5)) if k == "i" else float(rng.integers(-4, 5))) for _ in range(int(rng.integers(0, 4)))]
            items = [i for i in items if not (hasattr(i, "tokens") and any(t[0] == "bin" and t[1] == abi.BIN_DIV for t in i.tokens))]
            tgt = scalar(k)
            if any(t[0] == "bin" and t[1] == abi.BIN_DIV for t in tgt.tokens):
                tgt = col(1)
            return E.in_list(tgt, items, negated=bool(rng.random() < 0.4))
        if r < 0.95:
            return E.is_null(scalar(rng.choice(["i", "f"])), negated=bool(rng.random() < 0.5))
        return E.literal(bool(rng.random() < 0.5))

    def tree(d):
        if d == 0 or rng.random() < 0.25:
            return leaf()
        r = rng.random()
        if r < 0.35:
            return E.all_of([tree(d - 1) for _ in range(int(rng.integers(1, 4)))])
        if r < 0.7:
            return E.any_of([tree(d - 1) for _ in range(int(rng.integers(1, 4)))])
        return E.not_(tree(d - 1))

    return tree(depth)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6, 7, 8, 114, 122, 128, 141])  # the last four: computed NaNs under totalOrder compares
def test_random_predicate_trees_match_oracle(rt, orc, abi, seed):
    """Seeded random predicate trees (leaves, compares, IN lists, IS NULL over expressions, divisions, nested
    AND / OR / NOT over columns with NULL cells): the selected row ids must equal the oracle's, which restates the
    reference's three-valued domain algebra row set by row set."""
    rng = np.random.default_rng(100 + seed)
    chunks = [4096, 4097, 1000]
    n = sum(chunks)
    i1 = rng.integers(-4, 5, size=n).astype(np.int64)
    i2 = rng.integers(-4, 5, size=n).astype(np.int64)
    f3 = rng.integers(-4, 5, size=n).astype(np.float64)
    f3[rng.random(n) < 0.03] = np.nan
    f3[rng.random(n) < 0.03] = -0.0
    f4 = rng.integers(-4, 5, size=n).astype(np.float64) / 2
    i5 = rng.integers(-4, 5, size=n).astype(np.int32)
    v1, v3 = rng.random(n) > 0.25, rng.random(n) > 0.25
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT64, i1, v1), (2, abi.DT_INT64, i2), (3, abi.DT_FLOAT64, f3, v3), (4, abi.DT_FLOAT64, f4), (5, abi.DT_INT32, i5)], chunks)
    checked = 0
    for k in range(20):
        p = _random_predicate(rng, abi, 3)
        try:
            want = orc.filter_row_ids(ot, p)
        except abi.LlkvError as e:  # shapes neither side restates (e.g. Int32-only arithmetic) or arithmetic errors
            with pytest.raises(abi.LlkvError) as g:
                rt.filter_row_ids(ht, p)
            assert g.value.kind == e.kind or g.value.kind == "Unsupported", (k, e, g.value)
            continue
        try:
            got = rt.filter_row_ids(ht, p)
        except abi.LlkvError as g:
            assert g.kind == "Unsupported", (k, g)  # a shape only the GPU lowering declines
            continue
        assert np.array_equal(got, want), (seed, k)
        checked += 1
    assert checked >= 10


@pytest.mark.parametrize("seed", [1, 2, 3, 120, 122, 158])  # 122, 158: COUNT over an argument whose arithmetic fails
def test_random_aggregate_lists_match_oracle(rt, orc, abi, seed):
    """Seeded random aggregate lists over bare columns and random expressions (NULL cells, + - * / %), ungrouped
    (fast / generic projection typing) and grouped (PlanValue typing; dense and sort-based routes)."""
    rng = np.random.default_rng(500 + seed)
    chunks = [4096, 4097, 777]
    n = sum(chunks)
    i1 = rng.integers(-50, 50, size=n).astype(np.int64)
    i2 = rng.integers(-3, 4, size=n).astype(np.int64)
    f3 = rng.integers(-40, 40, size=n).astype(np.float64) / 4
    f3[rng.random(n) < 0.02] = np.nan
    f4 = rng.integers(1, 9, size=n).astype(np.float64) / 2
    key = np.array([ord("a"), ord("b"), ord("c")], dtype=np.uint8)[rng.integers(0, 3, size=n)]
    key2 = rng.integers(0, 700, size=n).astype(np.int64) * 1000  # sparse: the sort-based route
    v1, v3 = rng.random(n) > 0.2, rng.random(n) > 0.2
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT64, i1, v1), (2, abi.DT_INT64, i2), (3, abi.DT_FLOAT64, f3, v3), (4, abi.DT_FLOAT64, f4),
                                       (5, abi.DT_UTF8, key), (6, abi.DT_INT64, key2)], chunks)
    A, F, O, col = abi.AggregateSpec, abi.Filter, abi.Operator, abi.col

    def expr(grouped):
        e = col(int(rng.choice([1, 2, 3, 4])))
        for _ in range(int(rng.integers(1, 4))):
            other = col(int(rng.choice([1, 2, 3, 4]))) if rng.random() < 0.6 else (int(rng.integers(1, 5)) if rng.random() < 0.5 else float(rng.integers(1, 5)) / 2)
            op = rng.choice(["+", "-", "*", "/", "%"], p=[0.3, 0.25, 0.25, 0.1, 0.1])
            e = {"+": lambda a, b: a + b, "-": lambda a, b: a - b, "*": lambda a, b: a * b, "/": lambda a, b: a / b, "%": lambda a, b: a % b}[op](e, other)
        return e

    def agg_list(grouped):
        out = [A.count_star()]
        for _ in range(int(rng.integers(2, 6))):
            arg = int(rng.choice([1, 3, 4])) if rng.random() < 0.4 else expr(grouped)
            kind = rng.choice(["sum", "avg", "min", "max", "count", "total", "count_nulls"])
            out.append(getattr(A, kind)(arg))
        return out

    checked = 0
    for k in range(6):
        pred = [None, [F(2, O.GreaterThanOrEquals(0))], [F(1, O.LessThan(20))]][k % 3]
        for grouped, keys in ((False, None), (True, [5]), (True, [6])):
            aggs = agg_list(grouped)
            run_o = (lambda: orc.groupby(ot, pred, keys, aggs, True)) if grouped else (lambda: orc.aggregate(ot, pred, aggs))
            run_g = (lambda: rt.groupby(ht, pred, keys, aggs, True)) if grouped else (lambda: rt.aggregate(ht, pred, aggs))
            try:
                want = run_o()
            except abi.LlkvError as e:
                with pytest.raises(abi.LlkvError) as g:
                    run_g()
                assert g.value.kind in (e.kind, "Unsupported"), (k, grouped, e, g.value)
                continue
            try:
                got = run_g()
            except abi.LlkvError as g:
                assert g.kind == "Unsupported", (k, grouped, g)
                continue
            if grouped:
                assert [[x.value for x in r.keys] for r in got] == [[x.value for x in r.keys] for r in want]
                for a, b in zip(got, want):
                    assert_values(a.values, b.values, f"seed {seed} case {k} grouped", abs_floor=1e-9)
            else:
                assert_values(got, want, f"seed {seed} case {k}", abs_floor=1e-9)
            checked += 1
    assert checked >= 8


@pytest.mark.parametrize("case", golden("string_predicates.json")["cases"], ids=lambda c: c["name"])
def test_reference_string_predicate_known_answers(rt, abi, case):
    """typed_predicate.rs:538-600 through the GPU path: string predicates run on dictionary codes, the host having
    evaluated the predicate once per dictionary string."""
    from conftest import build_string_operator
    ht = rt.HipTable(1, [len(case["values"])])
    ht.append_utf8_column(1, case["values"])
    ids = set(rt.filter_row_ids(ht, [abi.Filter(1, build_string_operator(abi, case["op"]))]).tolist())
    assert [i in ids for i in range(len(case["values"]))] == case["expect"]


def test_concurrent_callers_get_sequential_answers(rt, abi, tpch):
    """The reference's storage traits are Send + Sync and queries arrive from several threads (SURVEY §8b
    "Threading"): four host threads share one table image and run aggregates, GROUP BYs (dense and sort-based),
    selections, scans and joins at the same time; every answer must equal the one computed alone."""
    import threading
    n = 300_000
    d = tpch.gen_lineitem(n, 0.05)
    ht = rt.HipTable(1, tpch.chunk_rows(n, 32768))
    for c, (fid, dt) in tpch.LINEITEM_SCHEMA.items():
        if dt == abi.DT_UTF8:
            ht.append_utf8_column(fid, d[c])
        else:
            ht.append_column(fid, dt, d[c])
    dim = rt.HipTable(2, [5000])
    dim.append_column(1, abi.DT_INT64, np.arange(1, 5001, dtype=np.int64) * 7)
    S, A, F, O, col = tpch.LINEITEM_SCHEMA, abi.AggregateSpec, abi.Filter, abi.Operator, abi.col
    q1, q6 = tpch.q1(), tpch.q6()
    flat = lambda rows: [(tuple(k.value for k in r.keys), tuple(np.float64(v.value).tobytes() if isinstance(v.value, float) else v.value for v in r.values)) for r in rows]
    jobs = {
        "q6": lambda: [np.float64(v.value).tobytes() for v in rt.aggregate(ht, q6.predicate, q6.aggs)],
        "q1": lambda: flat(rt.groupby(ht, q1.predicate, q1.keys, q1.aggs, True)),
        "by_partkey": lambda: flat(rt.groupby(ht, [F(S["l_quantity"][0], O.LessThan(10))], [S["l_partkey"][0]], [A.count_star(), A.sum(S["l_quantity"][0])], True))[:500],
        "row_ids": lambda: rt.filter_row_ids(ht, [F(S["l_discount"][0], O.GreaterThan(0.08))]).tolist(),
        "scan": lambda: [b[1][:50] for b in rt.scan_stream(ht, [S["l_orderkey"][0], col(S["l_extendedprice"][0]) * 2.0], [F(S["l_quantity"][0], O.Equals(7))], include_row_ids=True)],
        "join": lambda: [x for b in rt.join_stream(ht, dim, [(S["l_partkey"][0], 1)], JT["semi"], 8192) for x in b[0]][:2000],
        "join_topk": lambda: [(r[0], np.float64(r[1]).tobytes(), r[2]) for r in rt.join_groupby_topk(ht, [F(S["l_quantity"][0], O.LessThan(30))], S["l_partkey"][0], dim, [], 1,
                                                                                                   col(S["l_extendedprice"][0]) * (1 - col(S["l_discount"][0])), limit=7)[0]],
    }
    want = {k: f() for k, f in jobs.items()}
    errors, names = [], list(jobs)

    def worker(tid):
        try:
            for it in range(6):
                name = names[(tid + it) % len(names)]
                if jobs[name]() != want[name]:
                    errors.append((tid, it, name, "mismatch"))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, "exception", repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a worker hung"
    assert not errors, errors[:5]


@pytest.mark.parametrize("chunks", [[7], [4096, 4097, 5], [65536, 70000]])
def test_distinct_aggregates_match_oracle(rt, orc, abi, chunks):
    """COUNT / SUM / TOTAL / AVG (DISTINCT x): the accumulators add a value the first time they see it
    (llkv-aggregate/src/lib.rs:787-799,831-867,889-924), i.e. the distinct values in order of first appearance —
    Int keys by value, Float keys by bit pattern (NaN payloads and -0.0 are their own keys)."""
    rng = np.random.default_rng(31 + len(chunks))
    n = sum(chunks)
    few = rng.integers(-20, 20, size=n).astype(np.int64)
    many = rng.integers(-10**6, 10**6, size=n).astype(np.int64)
    f = rng.integers(-30, 30, size=n).astype(np.float64) / 4
    f[rng.random(n) < 0.03] = np.nan
    f[rng.random(n) < 0.03] = -0.0
    sel = rng.integers(0, 5, size=n).astype(np.int64)
    vf = rng.random(n) > 0.2
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT64, few), (2, abi.DT_INT64, many), (3, abi.DT_FLOAT64, f, vf), (4, abi.DT_INT64, sel)], chunks)
    A, F, O, col = abi.AggregateSpec, abi.Filter, abi.Operator, abi.col

    def D(kind, e):
        s = getattr(A, kind)(e)
        s.distinct = True
        return s

    aggs = [D("count", 1), D("sum", 1), D("avg", 1), D("total", 1), D("count", 2), D("sum", 2), D("count", 3), D("sum", 3), D("avg", 3), D("total", 3),
            D("sum", col(1) * col(4)), D("count", col(3) * 2.0), D("min", 1), D("max", 3), A.sum(1), A.count_star()]
    for pred in (None, [F(4, O.Equals(2))], [F(4, O.GreaterThan(9))]):
        got, want = rt.aggregate(ht, pred, aggs), orc.aggregate(ot, pred, aggs)
        assert_values(got, want, f"distinct {pred}")
        for i in (0, 1, 4, 5, 6, 10, 11):  # counts and integer sums are exact
            assert got[i].value == want[i].value
    # checked i64 adds over the distinct values, in first-appearance order
    big = np.array([2**62, 2**62, 5, 2**62 + 1] + [5] * (n - 4), dtype=np.int64)[:n] if n >= 4 else np.array([2**62] * n, dtype=np.int64)
    ht2, ot2 = stage_both(rt, orc, abi, [(1, abi.DT_INT64, big)], chunks)
    if n >= 4:
        for m, t in ((rt, ht2), (orc, ot2)):
            with pytest.raises(abi.LlkvError) as e:
                m.aggregate(t, None, [D("sum", 1)])
            assert e.value.kind == "InvalidArgumentError" and "integer overflow" in e.value.message
            with pytest.raises(abi.LlkvError) as e:
                m.aggregate(t, None, [D("avg", 1)])
            assert e.value.kind == "InvalidArgumentError" and "AVG(DISTINCT) aggregate sum exceeds i64 range" in e.value.message
            assert m.aggregate(t, None, [D("count", 1), D("total", 1)])[0].value == 3


@pytest.mark.parametrize("chunks", [[9], [4096, 4097, 5], [65536, 70000]])
def test_distinct_aggregates_over_string_boolean_date_and_decimal_keys(rt, orc, abi, chunks):
    """DistinctKey (llkv-aggregate/src/lib.rs:252-331) beyond Int / Float: strings by value (the staged dictionary holds each
    once: the code IS the key), booleans, dates, raw decimals; SUM / TOTAL / AVG add a new key's numeric image in order of
    first appearance (:889-924) or run in i128 over decimals (:943-967,1762-1800)."""
    rng = np.random.default_rng(41 + len(chunks))
    n = sum(chunks)
    words = ["12", " 3.5 ", "abc", "", "1e2", "12.0", "-7.25", ".5", "1.", "+4", "0.125", "1000000.5", "1", "1.0"]
    txt = [words[i] for i in rng.integers(0, len(words), size=n)]
    boo = rng.integers(0, 2, size=n).astype(np.uint8)
    day = rng.integers(9000, 9040, size=n).astype(np.int32)
    dec = [int(v) for v in rng.integers(-500, 500, size=n)]
    sel = rng.integers(0, 5, size=n).astype(np.int64)
    vt, vb, vd = rng.random(n) > 0.15, rng.random(n) > 0.1, rng.random(n) > 0.1
    ht = rt.HipTable(1, chunks)
    ht.append_utf8_column(1, txt, valid=vt)
    ht.append_column(2, abi.DT_BOOLEAN, boo, valid=vb)
    ht.append_column(3, abi.DT_DATE32, day)
    ht.append_decimal128_column(4, 12, 2, dec, valid=vd)
    ht.append_column(5, abi.DT_INT64, sel)
    ot = orc.OracleTable(n)
    ot.add(1, abi.DT_UTF8, [t if ok else None for t, ok in zip(txt, vt)]).add(2, abi.DT_BOOLEAN, boo, list(vb)).add(3, abi.DT_DATE32, day)
    ot.add(4, abi.DT_DECIMAL128, dec, list(vd), precision=12, scale=2).add(5, abi.DT_INT64, sel)
    A, F, O = abi.AggregateSpec, abi.Filter, abi.Operator

    def D(kind, e):
        s = getattr(A, kind)(e)
        s.distinct = True
        return s

    aggs = [D(k, f) for f in (1, 2, 3, 4) for k in ("count", "sum", "total", "avg")] + [A.count_star(), A.sum(5)]
    for pred in (None, [F(5, O.Equals(2))], [F(5, O.GreaterThan(9))]):
        got, want = rt.aggregate(ht, pred, aggs), orc.aggregate(ot, pred, aggs)
        assert got == want, pred  # the same keys in the same order: the same f64 chain, bit for bit; decimals as raw i128


@pytest.mark.parametrize("chunks", [[9], [4096, 4097, 5], [65536, 70000]])
def test_ordered_scans_match_oracle(rt, orc, abi, chunks):
    """ScanStreamOptions.order (sort_row_ids_with_order, llkv-scan/src/ordering.rs:16-140): the selected rows are
    sorted by one column — Int64 / Int32 / Utf8 (string order), ascending or descending, NULLs first or last —
    before they are cut into 65 536-row windows; ties keep row-id order."""
    rng = np.random.default_rng(41 + len(chunks))
    n = sum(chunks)
    i64 = rng.integers(-50, 50, size=n).astype(np.int64)
    i32 = rng.integers(-1000, 1000, size=n).astype(np.int32)
    tags = [("pear", "Apple", "fig", "zebra", "apple", "")[k] for k in rng.integers(0, 6, size=n)]
    f64 = rng.normal(size=n)
    v1, v3 = rng.random(n) > 0.2, rng.random(n) > 0.3
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT64, i64, v1), (2, abi.DT_INT32, i32), (3, abi.DT_UTF8, tags, v3), (4, abi.DT_FLOAT64, f64)], chunks)
    F, O = abi.Filter, abi.Operator
    specs = [(1, False, True, abi.ORDER_IDENTITY_INT64), (1, True, False, abi.ORDER_IDENTITY_INT64), (2, False, False, abi.ORDER_IDENTITY_INT32),
             (2, True, True, abi.ORDER_IDENTITY_INT32), (3, False, False, abi.ORDER_IDENTITY_UTF8), (3, True, True, abi.ORDER_IDENTITY_UTF8)]
    for order in specs if n < 20000 else specs[:1] + specs[5:]:  # (the Python oracle binding walks every cell: keep the large case short)
        for pred in (None, [F(4, O.GreaterThan(0.0))]) if n < 20000 else ([F(4, O.GreaterThan(0.0))],):
            got = rt.scan_stream(ht, [order[0], 4], pred, include_nulls=True, include_row_ids=True, order=order)
            want = orc.scan_stream(ot, [order[0], 4], pred, include_nulls=True, include_row_ids=True, order=order)
            assert [b[1] for b in got] == [b[1] for b in want], order
            assert [b[0][0] for b in got] == [b[0][0] for b in want], order
    for m, t in ((rt, ht), (orc, ot)):
        with pytest.raises(abi.LlkvError) as e:
            m.scan_stream(t, [1], None, order=(2, False, False, abi.ORDER_IDENTITY_INT64))
        assert e.value.kind == "InvalidArgumentError" and "IdentityInt64" in e.value.message


JOINS = golden("joins.json")
JT = {"inner": 0, "left": 1, "semi": 4, "anti": 5}


def _join_side(rt, abi, rows):
    t = rt.HipTable(1, [len(rows)])
    t.append_column(1, abi.DT_INT32, np.array([r[0] for r in rows], dtype=np.int32))
    t.append_utf8_column(2, [r[1] for r in rows])
    return t


@pytest.mark.parametrize("case", JOINS["cases"], ids=lambda c: c["name"])
def test_reference_join_known_answers(rt, abi, case):
    """llkv-join/tests/join_tests.rs through llkv_hip_join_stream."""
    left, right = _join_side(rt, abi, case["left"]), _join_side(rt, abi, case["right"])
    if "expect_error" in case:
        with pytest.raises(abi.LlkvError) as e:
            rt.join_stream(left, right, [] if case.get("cross") else [(1, 1)], JT[case["type"]], case.get("batch_size", 8192))
        assert e.value.kind == case["expect_error"]
        return
    batches = rt.join_stream(left, right, [] if case.get("cross") else [(1, 1)], JT[case["type"]], case.get("batch_size", 8192))
    ls = [x for b in batches for x in b[0]]
    assert len(ls) == case["expect_rows"]
    if "expect_pairs" in case:
        rs = [x for b in batches for x in b[1]]
        assert [[l, None if r == 2**64 - 1 else r] for l, r in zip(ls, rs)] == case["expect_pairs"]
    if "expect_left" in case:
        assert ls == case["expect_left"] and all(b[1] is None for b in batches)


JOIN_COLS = [(1, "user_id"), (2, "name")]  # create_test_table join_tests.rs:18-53


@pytest.mark.parametrize("case", JOINS["cases"], ids=lambda c: c["name"])
def test_reference_join_record_batches(rt, abi, case):
    """llkv-join/tests/join_tests.rs through llkv_hip_join_stream_batches — what the reference's `on_batch` receives:
    `expect_columns`, the output names (left, right, `_1`) and every cell of the joined rows, gathered on the device."""
    from test_oracle_golden import check_join_batches
    left, right = _join_side(rt, abi, case["left"]), _join_side(rt, abi, case["right"])
    keys = [] if case.get("cross") else [(1, 1)]
    if "expect_error" in case:
        with pytest.raises(abi.LlkvError) as e:
            rt.join_stream_batches(left, right, keys, JOIN_COLS, JOIN_COLS, JT[case["type"]], case.get("batch_size", 8192))
        assert e.value.kind == case["expect_error"]
        return
    check_join_batches(case, rt.join_stream_batches(left, right, keys, JOIN_COLS, JOIN_COLS, JT[case["type"]], case.get("batch_size", 8192)))


def test_reference_join_record_batches_expression_filters(rt, abi):
    """join_tests.rs:470-558 on the joined batches themselves (columns 0, 2, 5, 6 as the reference reads them)."""
    from test_oracle_golden import check_expression_filter_batches
    c = golden("join_filters.json")["expression_filters"]
    def mk(tid, cols):
        t = rt.HipTable(tid, [len(cols[0]["values"])])
        for col in cols:
            if col["dtype"] == "Utf8":
                t.append_utf8_column(col["field_id"], col["values"])
            else:
                t.append_column(col["field_id"], DTYPES[col["dtype"]], np.array(col["values"], dtype=abi.NUMPY_OF_DTYPE[DTYPES[col["dtype"]]]))
        return t
    left, right = mk(41, c["left"]["columns"]), mk(84, c["right"]["columns"])
    lcols = [(col["field_id"], nm) for col, nm in zip(c["left"]["columns"], ["customer_id", "segment", "annual_revenue", "loyalty_score"])]
    rcols = [(col["field_id"], nm) for col, nm in zip(c["right"]["columns"], ["order_id", "customer_id", "avg_order_value", "trailing_spend"])]
    check_expression_filter_batches(c, rt.join_stream_batches(left, right, [tuple(k) for k in c["join_keys"]], lcols, rcols, JT["inner"]))


def _wide_join_tables(rt, orc, abi, n_left, n_right, keyspace, chunks_left=None, seed=0, key_dtype=None, all_nullable=False):
    """Two tables with a key and a mix of payload types (Int64 / Float64 / Int32 / Date32 / Utf8 / Decimal128 / nullable
    columns — seven user columns a side, so every gather runs in two launches), staged on both engines."""
    rng = np.random.default_rng(seed + n_left + 3 * n_right)
    key_dtype = key_dtype or abi.DT_INT64
    npk = np.dtype(abi.NUMPY_OF_DTYPE[key_dtype])
    words = ["ash", "birch", "cedar", "<NULL>", "elm", ""]
    def side(tid, n, chunks, lo, hi, base):
        ht, ot = rt.HipTable(tid, chunks or [n]), orc.OracleTable(n)
        kv = rng.random(n) > 0.05
        cols = [(base + 1, key_dtype, rng.integers(lo, hi, size=n).astype(npk), kv if all_nullable or tid == 1 else None),
                (base + 2, abi.DT_FLOAT64, rng.normal(size=n), (rng.random(n) > 0.2) if all_nullable else None),
                (base + 3, abi.DT_INT32, rng.integers(-5, 5, size=n).astype(np.int32), rng.random(n) > 0.3),
                (base + 4, abi.DT_DATE32, rng.integers(8000, 9000, size=n).astype(np.int32), (rng.random(n) > 0.2) if all_nullable else None),
                (base + 6, abi.DT_INT64, rng.integers(-2**62, 2**62, size=n), rng.random(n) > 0.5)]
        for f, dt, vals, valid in cols:
            ht.append_column(f, dt, vals, valid=valid)
            ot.add(f, dt, vals, None if valid is None else list(valid))
        strs = [words[i] for i in rng.integers(0, len(words), size=n)]
        sv = rng.random(n) > 0.25
        ht.append_utf8_column(base + 5, strs, valid=sv)
        ot.add(base + 5, abi.DT_UTF8, [s if ok else None for s, ok in zip(strs, sv)])
        dec = [int(v) for v in rng.integers(-10**12, 10**12, size=n)]
        dv = rng.random(n) > 0.1
        ht.append_decimal128_column(base + 7, 15, 2, dec, valid=dv)
        ot.add(base + 7, abi.DT_DECIMAL128, dec, list(dv), precision=15, scale=2)
        names = ["key", "f", "i32", "d", "big", "s", "dec"]
        return ht, ot, [(base + 1 + i if i < 4 else base + {4: 6, 5: 5, 6: 7}[i], names[i]) for i in range(7)]
    lt, ol, lcols = side(1, n_left, chunks_left, -keyspace, keyspace, 0)
    rtab, orr, rcols = side(2, n_right, None, -keyspace // 2, 2 * keyspace, 10)
    return lt, rtab, ol, orr, lcols, rcols


def _same_batches(got, want, what=""):
    assert len(got) == len(want), (what, len(got), len(want))
    assert [names for names, _ in got] == [names for names, _ in want], what
    assert [len(cols[0]) for _, cols in got] == [len(cols[0]) for _, cols in want], what  # the reference's batch cuts
    for (_, g), (_, w) in zip(got, want):
        for ci, (gc, wc) in enumerate(zip(g, w)):
            if gc != wc:  # NaN-free columns: plain equality of the cells (None = NULL)
                bad = next(i for i, (a, b) in enumerate(zip(gc, wc)) if a != b)
                raise AssertionError(f"{what}: column {ci} row {bad}: {gc[bad]!r} != {wc[bad]!r}")


@pytest.mark.parametrize("n_left,n_right,keyspace,batch,chunks", [(1000, 300, 50, 8192, None), (150_000, 40_000, 30_000, 8192, [65536, 70000, 14464]),
                                                                   (70_000, 10, 3, 1000, [4096, 60000, 5904]), (5, 20_000, 1000, 7, None),
                                                                   (9000, 700, 200, 1, [1000, 8000])])
@pytest.mark.parametrize("jt", ["inner", "left", "semi", "anti"])
def test_random_join_record_batches_match_oracle(rt, orc, abi, n_left, n_right, keyspace, batch, chunks, jt):
    """Whole joined batches against an oracle that gathers the way the reference does: names, batch cuts, every cell
    (NULL keys, NULL payload cells, NULL padding of LEFT joins, Utf8 / Decimal128 / Date32 columns, ragged chunks so that
    device steps end inside a reference batch, many-to-many keys, batch_size 1)."""
    lt, rtab, ol, orr, lcols, rcols = _wide_join_tables(rt, orc, abi, n_left, n_right, keyspace, chunks)
    got = rt.join_stream_batches(lt, rtab, [(1, 11)], lcols, rcols, JT[jt], batch)
    want = orc.hash_join_batches(ol, orr, [(1, 11)], lcols, rcols, JT[jt], batch)
    _same_batches(got, want, jt)
    assert len(want) > 0


def test_join_record_batches_generic_path_and_executor_rules(rt, orc, abi):
    """Composite / mixed keys (the generic typed-key path: slices of batch_size probe rows) and the executor's rules
    (one batch per device step here, ONE batch in the reference: compared as the concatenation)."""
    lt, rtab, ol, orr, lcols, rcols = _wide_join_tables(rt, orc, abi, 40_000, 9000, 300, [30_000, 10_000], seed=7)
    small = _wide_join_tables(rt, orc, abi, 3000, 150, 300, [1000, 2000], seed=8)  # low-cardinality keys: many-to-many
    for keys, tabs in (([(1, 11), (3, 13)], None), ([(1, 11, True), (3, 13, True)], None), ([(5, 15)], small), ([(5, 15, True)], small), ([(3, 13)], small)):
        a, b, oa, ob, lc, rc_ = tabs or (lt, rtab, ol, orr, lcols, rcols)
        for jt in ("inner", "left", "semi", "anti"):
            for batch in (8192, 100):
                _same_batches(rt.join_stream_batches(a, b, keys, lc, rc_, JT[jt], batch), orc.hash_join_batches(oa, ob, keys, lc, rc_, JT[jt], batch), (keys, jt, batch))
    for jt in ("inner", "left"):
        got = rt.join_stream_batches(lt, rtab, [(1, 11), (3, 13)], lcols, rcols, JT[jt], key_rules=1)
        want = orc.hash_join_batches(ol, orr, [(1, 11), (3, 13)], lcols, rcols, JT[jt], key_rules=1)
        assert len(want) == 1 and all(names == want[0][0] for names, _ in got)
        assert want[0][0][7:] == [nm for _, nm in rcols]  # the executor keeps the names as given (no `_1`)
        cat = [[v for _, cols in got for v in cols[ci]] for ci in range(len(want[0][1]))]
        assert cat == want[0][1]


def test_join_record_batches_drop_rows_null_in_every_column(rt, orc, abi):
    """The reference reads both sides with scan_stream's DropNulls gather: a row that is NULL in EVERY user column never
    reaches the join — not built, not probed, not padded by a LEFT join, not listed by an ANTI join."""
    lt, rtab, ol, orr, lcols, rcols = _wide_join_tables(rt, orc, abi, 70_000, 3000, 500, [40_000, 30_000], seed=3, all_nullable=True)
    # two columns a side, both nullable: a fifth of the rows is NULL in both
    lsub, rsub = [lcols[0], lcols[2]], [rcols[0], rcols[2]]
    for jt in ("inner", "left", "semi", "anti"):
        got = rt.join_stream_batches(lt, rtab, [(1, 11, True)], lsub, rsub, JT[jt], 500)
        want = orc.hash_join_batches(ol, orr, [(1, 11, True)], lsub, rsub, JT[jt], 500)
        _same_batches(got, want, jt)
    rows = sum(len(c[0]) for _, c in orc.hash_join_batches(ol, orr, [(1, 11)], lsub, rsub, JT["left"], 500))
    assert rows < sum(len(b[0]) for b in orc.hash_join(ol, orr, [(1, 11)], JT["left"], 500))  # the index-pair delivery lists every row
    # the generic path counts its slices in surviving rows: not on the GPU path (nor restated)
    for m, a, b in ((rt, lt, rtab), (orc, ol, orr)):
        with pytest.raises(abi.LlkvError) as e:
            m.join_stream_batches(a, b, [(1, 11), (3, 13)], lsub, rsub, JT["inner"]) if m is rt else m.hash_join_batches(a, b, [(1, 11), (3, 13)], lsub, rsub, JT["inner"])
        assert e.value.kind == "Unsupported"


def test_a_join_side_whose_every_row_is_null_arrives_as_one_synthetic_batch(rt, orc, abi):
    """A table none of whose rows survives the DropNulls gather of its user columns comes out of the reference's scan as ONE batch
    of total_rows NULL rows (llkv-scan/src/execute.rs:355-372, llkv-compute/src/projection.rs:36-66) instead of no batch at all:
    as a probe side it is one window of 70 000 rows (a LEFT join pads every one of them, cut by the batch size alone), as a build
    side its NULL keys meet only NULL keys that are allowed to (null_equals_null), in a cross product it is one window."""
    n = 70_000  # two scan windows of a side that has rows
    rng = np.random.default_rng(5)
    none = np.zeros(n, dtype=bool)
    nulls_h, nulls_o = rt.HipTable(1, [65536, n - 65536]), orc.OracleTable(n)
    for f, dt, v in ((1, abi.DT_INT64, rng.integers(0, 50, size=n)), (2, abi.DT_FLOAT64, rng.normal(size=n))):
        nulls_h.append_column(f, dt, v.astype(abi.NUMPY_OF_DTYPE[dt]), valid=none); nulls_o.add(f, dt, v.astype(abi.NUMPY_OF_DTYPE[dt]), list(none))
    m = 3000
    rk, rvalid = rng.integers(0, 50, size=m).astype(np.int64), rng.random(m) > 0.3
    w12 = rng.normal(size=m)
    live_h = rt.HipTable(2, [m])
    live_h.append_column(11, abi.DT_INT64, rk, valid=rvalid); live_h.append_column(12, abi.DT_FLOAT64, w12)
    live_o = orc.OracleTable(m).add(11, abi.DT_INT64, rk, list(rvalid)).add(12, abi.DT_FLOAT64, w12)
    lcols, rcols = [(1, "k"), (2, "v")], [(11, "rk"), (12, "w")]
    # probe side all NULL
    for jt, keys in (("inner", [(1, 11)]), ("left", [(1, 11)]), ("anti", [(1, 11)]), ("semi", [(1, 11, True)]), ("left", [(1, 11, True)])):
        got = rt.join_stream_batches(nulls_h, live_h, keys, lcols, rcols, JT[jt], 5000)  # (5 000 does not divide a 65 536-row window)
        want = orc.hash_join_batches(nulls_o, live_o, keys, lcols, rcols, JT[jt], 5000)
        _same_batches(got, want, jt)
    assert [len(c[0]) for _, c in orc.hash_join_batches(nulls_o, live_o, [(1, 11)], lcols, rcols, JT["left"], 5000)] == [5000] * 14
    # build side all NULL
    for jt, keys in (("inner", [(11, 1)]), ("left", [(11, 1)]), ("inner", [(11, 1, True)]), ("anti", [(11, 1)])):
        got = rt.join_stream_batches(live_h, nulls_h, keys, rcols, lcols, JT[jt], 512)
        want = orc.hash_join_batches(live_o, nulls_o, keys, rcols, lcols, JT[jt], 512)
        _same_batches(got, want, jt)
    # cross products: the all-NULL side is ONE window whichever side it is on
    small_h, small_o = rt.HipTable(6, [3]), orc.OracleTable(3)
    small_h.append_utf8_column(7, ["x", "y", "x"]); small_o.add(7, abi.DT_UTF8, ["x", "y", "x"])
    got = rt.join_stream_batches(nulls_h, small_h, [], [(1, "k")], [(7, "s")], JT["inner"])
    want = orc.hash_join_batches(nulls_o, small_o, [], [(1, "k")], [(7, "s")], JT["inner"])
    _same_batches(got, want, "cross, left side all NULL")
    assert [len(c[0]) for _, c in want] == [3 * n]
    got = rt.join_stream_batches(small_h, nulls_h, [], [(7, "s")], [(2, "v")], JT["inner"])
    want = orc.hash_join_batches(small_o, nulls_o, [], [(7, "s")], [(2, "v")], JT["inner"])
    _same_batches(got, want, "cross, right side all NULL")


def test_join_record_batches_edge_cases(rt, orc, abi):
    """Empty sides, no columns on a side, the reference's LEFT-join-without-a-build-batch behaviour, cross products with
    NULL padding, SEMI / ANTI cross products."""
    lt, rtab, ol, orr, lcols, rcols = _wide_join_tables(rt, orc, abi, 3000, 500, 100, seed=11)
    e_h = rt.HipTable(3, [0]); e_o = orc.OracleTable(0)
    for f, dt in ((11, abi.DT_INT64), (13, abi.DT_INT32)):
        e_h.append_column(f, dt, np.zeros(0, dtype=abi.NUMPY_OF_DTYPE[dt])); e_o.add(f, dt, np.zeros(0, dtype=abi.NUMPY_OF_DTYPE[dt]))
    ecols = [(11, "key"), (13, "i32")]
    def both(keys, lc, rc_, jt, right=(rtab, orr), left=(lt, ol), **kw):
        out = []
        for m, a, b in ((rt, left[0], right[0]), (orc, left[1], right[1])):
            f = m.join_stream_batches if m is rt else m.hash_join_batches
            try:
                out.append(f(a, b, keys, lc, rc_, JT[jt], **kw))
            except abi.LlkvError as e:
                out.append((e.kind, e.message))
        return out
    # empty build side: INNER / SEMI nothing, ANTI every probe row
    for jt in ("inner", "semi", "anti"):
        g, w = both([(1, 11)], lcols, ecols, jt, right=(e_h, e_o))
        _same_batches(g, w, jt)
        assert (len(w) > 0) == (jt == "anti")
    # LEFT join, no build batch: the fast path drops every probe batch (the failed RecordBatch::try_new is logged and
    # swallowed), the generic path returns that error
    g, w = both([(1, 11)], lcols, ecols, "left", right=(e_h, e_o))
    assert g == w == []
    g, w = both([(1, 11), (3, 13)], lcols, ecols, "left", right=(e_h, e_o))
    assert g == w and g[0] == "Internal" and "number of columns(7) must match number of fields(9)" in g[1]
    # no right columns: nothing is built; no left columns: nothing is probed
    for jt in ("inner", "anti"):
        g, w = both([(1, 11)], lcols, [], jt)
        _same_batches(g, w, jt)
    assert both([(1, 11)], [], rcols, "inner") == [[], []]
    # empty probe side
    assert both([(11, 11)], ecols, rcols, "left", left=(e_h, e_o)) == [[], []]
    # cross products: two left windows × one right window; LEFT with an empty right side pads; SEMI / ANTI fail on the
    # first pair of batches and deliver nothing against an empty side
    big_h, big_o = rt.HipTable(5, [65536, 4464]), orc.OracleTable(70_000)
    v = np.arange(70_000, dtype=np.int64); ok = (v % 3) != 0
    big_h.append_column(1, abi.DT_INT64, v, valid=ok); big_o.add(1, abi.DT_INT64, v, list(ok))
    small_h, small_o = rt.HipTable(6, [3]), orc.OracleTable(3)
    small_h.append_utf8_column(7, ["x", "y", "x"]); small_o.add(7, abi.DT_UTF8, ["x", "y", "x"])
    g, w = both([], [(1, "v")], [(7, "s")], "inner", left=(big_h, big_o), right=(small_h, small_o))
    _same_batches(g, w, "cross")
    assert [len(c[0]) for _, c in w] == [3 * int(ok[:65536].sum()), 3 * int(ok[65536:].sum())]  # rows NULL in every column are dropped before the product
    g, w = both([], [(1, "v")], ecols, "left", left=(big_h, big_o), right=(e_h, e_o))
    _same_batches(g, w, "cross left")
    assert w[0][0] == ["v", "key", "i32"] and set(w[0][1][1]) == {None}
    g, w = both([], [(1, "v")], [(7, "s")], "semi", left=(big_h, big_o), right=(small_h, small_o))
    assert g == w and g[0] == "Internal"
    assert both([], [(1, "v")], ecols, "anti", left=(big_h, big_o), right=(e_h, e_o)) == [[], []]
    assert rt.join_output_names(lcols[:2], [(11, "key"), (12, "key_1"), (13, "f")]) == ["key", "f", "key_1", "key_1_1", "f_1"]  # hash_join.rs:913-939: the set of taken names includes the renamed ones
    assert rt.join_output_names(lcols[:2], rcols[:2], JT["semi"]) == ["key", "f"]


def test_join_record_batches_sharded_probe_side_concatenates(rt, orc, abi):
    """Probe side sharded by chunk over 4 ranks (emulated on one device), build side replicated: the ranks' batches in
    rank order are the single-device rows (batch cuts follow each rank's own scan windows, as its reference would)."""
    n_left, n_right = 300_000, 20_000
    rng = np.random.default_rng(9)
    lk, lv = rng.integers(0, 30_000, size=n_left), rng.normal(size=n_left)
    rk, rv = rng.integers(0, 40_000, size=n_right), rng.integers(0, 99, size=n_right).astype(np.int32)
    chunks = [65536] * 4 + [n_left - 4 * 65536]
    rtab = rt.HipTable(2, [n_right]); rtab.append_column(11, abi.DT_INT64, rk); rtab.append_column(12, abi.DT_INT32, rv)
    orr = orc.OracleTable(n_right).add(11, abi.DT_INT64, rk).add(12, abi.DT_INT32, rv)
    ol = orc.OracleTable(n_left).add(1, abi.DT_INT64, lk).add(2, abi.DT_FLOAT64, lv)
    want = orc.hash_join_batches(ol, orr, [(1, 11)], [(1, "k"), (2, "v")], [(11, "k"), (12, "w")], JT["left"])
    rows = []
    for rank in range(4):
        lt = rt.HipTable(1, chunks, rank=rank, world=4)
        lo = sum(chunks[:lt.first_chunk])
        lt.append_column(1, abi.DT_INT64, lk[lo:lo + lt.local_rows]); lt.append_column(2, abi.DT_FLOAT64, lv[lo:lo + lt.local_rows])
        for names, cols in rt.join_stream_batches(lt, rtab, [(1, 11)], [(1, "k"), (2, "v")], [(11, "k"), (12, "w")], JT["left"]):
            assert names == ["k", "v", "k_1", "w"]
            rows.extend(zip(*cols))
    assert rows == [r for _, cols in want for r in zip(*cols)]


def test_cross_products_match_oracle(rt, orc, abi):
    """Empty join keys = Cartesian product (llkv-join/src/hash_join.rs:1500-1599): window by window, left-major;
    LEFT with an empty right side pads with NULLs; SEMI / ANTI fail the reference's schema check."""
    nl, nr = 70_000, 300  # two left scan windows
    lt = rt.HipTable(1, [nl]); lt.append_column(1, abi.DT_INT64, np.arange(nl, dtype=np.int64))
    rtab = rt.HipTable(2, [nr]); rtab.append_column(7, abi.DT_INT64, np.arange(nr, dtype=np.int64))
    empty = rt.HipTable(3, [0]); empty.append_column(7, abi.DT_INT64, np.zeros(0, dtype=np.int64))
    ol, orr, oe = orc.OracleTable(nl).add(1, abi.DT_INT64, np.arange(nl, dtype=np.int64)), orc.OracleTable(nr).add(7, abi.DT_INT64, np.arange(nr, dtype=np.int64)), orc.OracleTable(0).add(7, abi.DT_INT64, np.zeros(0, dtype=np.int64))
    got, want = rt.join_stream(lt, rtab, [], JT["inner"]), orc.hash_join(ol, orr, [], JT["inner"])
    assert [len(b[0]) for b in got] == [len(b[0]) for b in want] == [65536 * nr, (nl - 65536) * nr]
    for (gl, gr_), (wl, wr) in zip(got, want):
        assert gl[:1000] == wl[:1000] and gr_[:1000] == wr[:1000] and gl[-1000:] == wl[-1000:] and gr_[-1000:] == wr[-1000:]
    got, want = rt.join_stream(lt, empty, [], JT["left"]), orc.hash_join(ol, oe, [], JT["left"])
    assert [(b[0][0], b[0][-1], b[1][0], len(b[0])) for b in got] == [(b[0][0], b[0][-1], b[1][0], len(b[0])) for b in want]
    assert rt.join_stream(lt, empty, [], JT["inner"]) == []
    for m, (a, b) in ((rt, (lt, rtab)), (orc, (ol, orr))):
        with pytest.raises(abi.LlkvError) as e:
            (m.join_stream if m is rt else m.hash_join)(a, b, [], JT["semi"])
        assert e.value.kind == "Internal"


def test_right_and_full_joins_are_rejected_like_the_reference(rt, abi):
    left, right = _join_side(rt, abi, [[1, "a"]]), _join_side(rt, abi, [[1, "b"]])
    for jt in (abi.JOIN_RIGHT, abi.JOIN_FULL):
        with pytest.raises(abi.LlkvError) as e:
            rt.join_stream(left, right, [(1, 1)], jt)
        assert e.value.kind == "InvalidArgumentError"


@pytest.mark.parametrize("n_left,n_right,keyspace,batch", [(1000, 300, 50, 8192), (200_000, 50_000, 40_000, 8192), (70_000, 10, 3, 1000), (5, 100_000, 1000, 7)])
@pytest.mark.parametrize("jt", ["inner", "left", "semi", "anti"])
def test_random_joins_match_oracle(rt, orc, abi, n_left, n_right, keyspace, batch, jt):
    """Many-to-many duplicates, skew, misses: pair SEQUENCE identical to the reference order (probe order ×
    build insertion order); batch boundaries follow the flush rule within each 65 536-row probe window."""
    rng = np.random.default_rng(n_left + n_right)
    lk = rng.integers(-keyspace, keyspace, size=n_left).astype(np.int64)
    rk = rng.integers(-keyspace // 2, keyspace * 2, size=n_right).astype(np.int64)
    lt = rt.HipTable(1, [n_left]); lt.append_column(1, abi.DT_INT64, lk)
    rtab = rt.HipTable(2, [n_right]); rtab.append_column(7, abi.DT_INT64, rk)
    ol = orc.OracleTable(n_left).add(1, abi.DT_INT64, lk)
    orr = orc.OracleTable(n_right).add(7, abi.DT_INT64, rk)
    got = rt.join_stream(lt, rtab, [(1, 7)], JT[jt], batch)
    want = orc.hash_join(ol, orr, [(1, 7)], JT[jt], batch)
    gl = [x for b in got for x in b[0]]; wl = [x for b in want for x in b[0]]
    assert gl == wl
    if jt in ("inner", "left"):
        assert [x for b in got for x in b[1]] == [x for b in want for x in b[1]]
    else:
        assert all(b[1] is None for b in got)
    assert [len(b[0]) for b in got] == [len(b[0]) for b in want]  # flushes: ≥ batch_size pairs, and every 65 536 probe rows


def test_join_build_side_with_a_hot_key(rt, orc, abi):
    """One key holds half of the build rows (200 000 duplicates): the build's run lengths come from a gallop + bisection at
    each run head, not from a walk of the run; pairs and batches as the reference's."""
    n_right, n_left = 400_000, 60
    rng = np.random.default_rng(21)
    rk = rng.integers(1000, 2_000_000, size=n_right).astype(np.int64)
    rk[rng.permutation(n_right)[:n_right // 2]] = 7
    lk = rng.integers(1000, 2_000_000, size=n_left).astype(np.int64)
    lk[[3, 41]] = 7
    lt = rt.HipTable(1, [n_left]); lt.append_column(1, abi.DT_INT64, lk)
    rtab = rt.HipTable(2, [131072, 131072, 137856]); rtab.append_column(7, abi.DT_INT64, rk)
    ol, orr = orc.OracleTable(n_left).add(1, abi.DT_INT64, lk), orc.OracleTable(n_right).add(7, abi.DT_INT64, rk)
    for jt in ("inner", "semi", "anti"):
        got, want = rt.join_stream(lt, rtab, [(1, 7)], JT[jt], 8192), orc.hash_join(ol, orr, [(1, 7)], JT[jt], 8192)
        assert [len(b[0]) for b in got] == [len(b[0]) for b in want], jt
        for (gl, gr_), (wl, wr) in zip(got, want):
            assert gl == wl and gr_ == wr, jt
    n_pairs = sum(len(b[0]) for b in orc.hash_join(ol, orr, [(1, 7)], JT["inner"], 8192))
    assert n_pairs >= 2 * (n_right // 2)
    got = rt.join_stream_batches(lt, rtab, [(1, 7)], [(1, "k")], [(7, "k")], JT["inner"], 8192)
    assert sum(len(c[0]) for _, c in got) == n_pairs and all(c[0] == c[1] for _, c in got)


@pytest.mark.parametrize("dt", ["DT_INT64", "DT_INT32", "DT_UINT32", "DT_UINT64"])
@pytest.mark.parametrize("null_eq", [False, True])
def test_joins_with_null_keys_match_oracle(rt, orc, abi, dt, null_eq):
    """A NULL key matches nothing (hash_join.rs:1116-1123,1172-1177); with null_equals_null the reference
    substitutes a per-type sentinel, so a real key of that value joins with the NULLs (:1429-1465)."""
    dtype = getattr(abi, dt)
    npdt = np.dtype(abi.NUMPY_OF_DTYPE[dtype])
    sentinel = {abi.DT_INT64: -2**63, abi.DT_INT32: -2**31, abi.DT_UINT32: 2**32 - 1, abi.DT_UINT64: 2**64 - 1}[dtype]
    rng = np.random.default_rng(5)
    n_left, n_right = 9000, 3000
    lk = rng.integers(0, 400, size=n_left).astype(npdt)
    rk = rng.integers(100, 600, size=n_right).astype(npdt)
    lk[rng.random(n_left) < 0.01] = sentinel
    rk[rng.random(n_right) < 0.01] = sentinel
    lv, rv = rng.random(n_left) > 0.1, rng.random(n_right) > 0.1
    lt = rt.HipTable(1, [4096, n_left - 4096]); lt.append_column(1, dtype, lk, valid=lv)
    rtab = rt.HipTable(2, [n_right]); rtab.append_column(7, dtype, rk, valid=rv)
    ol = orc.OracleTable(n_left).add(1, dtype, lk, list(lv))
    orr = orc.OracleTable(n_right).add(7, dtype, rk, list(rv))
    for jt in ("inner", "left", "semi", "anti"):
        got = rt.join_stream(lt, rtab, [(1, 7, null_eq)], JT[jt], 8192)
        want = orc.hash_join(ol, orr, [(1, 7, null_eq)], JT[jt], 8192)
        assert [x for b in got for x in b[0]] == [x for b in want for x in b[0]], jt
        if jt in ("inner", "left"):
            assert [x for b in got for x in b[1]] == [x for b in want for x in b[1]], jt
        assert [len(b[0]) for b in got] == [len(b[0]) for b in want]
    # two different key types leave the integer fast path (hash_join.rs:174-198): the generic path compares typed
    # values, which are then never equal
    if dtype != abi.DT_INT64:
        other = rt.HipTable(3, [4]); other.append_column(9, abi.DT_INT64, np.arange(4, dtype=np.int64))
        assert rt.join_stream(lt, other, [(1, 9)], JT["inner"], 8192) == []


def _keyed_tables(rt, orc, abi, cols_left, cols_right, chunks_left, n_right):
    """cols: [(field, dtype, values, valid|None)] staged on both engines."""
    lt, rtab = rt.HipTable(1, chunks_left), rt.HipTable(2, [n_right])
    ol, orr = orc.OracleTable(sum(chunks_left)), orc.OracleTable(n_right)
    for ht, ot, cols in ((lt, ol, cols_left), (rtab, orr, cols_right)):
        for f, dt, vals, valid in cols:
            if dt == abi.DT_UTF8:
                ht.append_utf8_column(f, vals)
                ot.add(f, dt, vals)
            else:
                ht.append_column(f, dt, vals, valid=valid)
                ot.add(f, dt, vals, None if valid is None else list(valid))
    return lt, rtab, ol, orr


def _same_join(rt, orc, tables, keys, batch=8192, jts=("inner", "left", "semi", "anti")):
    lt, rtab, ol, orr = tables
    for jt in jts:
        got, want = rt.join_stream(lt, rtab, keys, JT[jt], batch), orc.hash_join(ol, orr, keys, JT[jt], batch)
        assert [x for b in got for x in b[0]] == [x for b in want for x in b[0]], (keys, jt)
        if jt in ("inner", "left"):
            assert [x for b in got for x in b[1]] == [x for b in want for x in b[1]], (keys, jt)
        assert [len(b[0]) for b in got] == [len(b[0]) for b in want], (keys, jt, batch)
    return got


@pytest.mark.parametrize("null_eq", [(False, False), (True, False), (True, True)])
def test_composite_key_joins_match_oracle(rt, orc, abi, null_eq):
    """Key lists that are not one fast integer pair take the reference's generic typed-key path
    (llkv-join/src/hash_join.rs:200-335,377-505): all parts equal, NULL handling per part, probe cut into slices
    of batch_size rows.  Ragged chunks on the probe side: reference batches that span two device windows."""
    rng = np.random.default_rng(77)
    chunks, n_right = [30_000, 50_001, 7, 20_000], 40_000
    n_left = sum(chunks)
    la, lb = rng.integers(0, 300, size=n_left).astype(np.int64), rng.integers(0, 40, size=n_left).astype(np.int32)
    ra, rb = rng.integers(0, 300, size=n_right).astype(np.int64), rng.integers(0, 40, size=n_right).astype(np.int32)
    lva, lvb, rva, rvb = rng.random(n_left) > 0.05, rng.random(n_left) > 0.05, rng.random(n_right) > 0.05, rng.random(n_right) > 0.05
    tabs = _keyed_tables(rt, orc, abi, [(1, abi.DT_INT64, la, lva), (2, abi.DT_INT32, lb, lvb)], [(7, abi.DT_INT64, ra, rva), (8, abi.DT_INT32, rb, rvb)], chunks, n_right)
    keys = [(1, 7, null_eq[0]), (2, 8, null_eq[1])]
    got = _same_join(rt, orc, tabs, keys, 8192)
    assert sum(len(b[0]) for b in got) > 0
    _same_join(rt, orc, tabs, keys, 1000, jts=("inner", "anti"))
    _same_join(rt, orc, tabs, keys, 100_000, jts=("left",))
    # the same columns through the single-key fast path keep its batching (no slices) on the ragged probe side
    _same_join(rt, orc, tabs, [(1, 7, null_eq[0])], 5000, jts=("inner", "semi"))


@pytest.mark.parametrize("chunks", [[11], [4096, 4097, 5], [65536, 40000]])
def test_float_min_max_over_columns_without_nan_or_negative_zero(rt, orc, abi, chunks, monkeypatch):
    """MinFloat64 / MaxFloat64 fold by partial_cmp in row order (a leading NaN sticks, ±0 ties keep the earlier row:
    llkv-aggregate/src/lib.rs:1309-1331,1377-1399).  A bare column whose staging statistics show neither NaN / ±∞ nor −0.0 has no such
    case and takes ONE order-key lane (`MinF64P` / `MaxF64P`); a column with a −0.0 or a NaN keeps the three row-order lanes.  Both
    against the oracle — ungrouped, per-thread columns, shared image, sort-based — with NULL cells and +0.0 values."""
    rng = np.random.default_rng(3 + len(chunks))
    n = sum(chunks)
    clean = rng.integers(-50, 50, size=n).astype(np.float64) / 4      # plenty of +0.0, no −0.0, no NaN
    dirty = clean.copy()
    dirty[rng.random(n) < 0.05] = -0.0
    dirty[rng.random(n) < 0.02] = np.nan
    dirty[3], dirty[7] = -0.0, np.nan  # (eleven rows may draw neither)
    valid = rng.random(n) > 0.2
    small = rng.integers(0, 4, size=n).astype(np.int64)
    day = rng.integers(9000, 9700, size=n).astype(np.int32)
    sparse = (rng.integers(0, 90, size=n) * 1_000_003).astype(np.int64)
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_FLOAT64, clean, valid), (2, abi.DT_FLOAT64, dirty), (3, abi.DT_INT64, small), (4, abi.DT_DATE32, day), (5, abi.DT_INT64, sparse),
                                       (6, abi.DT_FLOAT64, clean)], chunks)
    A, F, O = abi.AggregateSpec, abi.Filter, abi.Operator
    aggs = [A.min(1), A.max(1), A.min(6), A.max(6), A.min(2), A.max(2), A.count(1)]
    pq = rt.PreparedQuery(ht, None, aggs, [3], True)
    sig = pq.kernel_signature
    pq.close()
    assert "MinF64P<" in sig and "MaxF64P<" in sig and "MinF64<Col<" in sig  # clean columns: one lane; the dirty one: three
    for order_env in (None, "1"):
        if order_env:
            monkeypatch.setenv("LLKV_HIP_MINMAX_ROW_ORDER", order_env)  # the row-order lanes everywhere: the same answers
        for pred in (None, [F(3, O.GreaterThan(0))], [F(6, O.GreaterThan(1000.0))]):
            assert_values(rt.aggregate(ht, pred, aggs), orc.aggregate(ot, pred, aggs), "ungrouped")
            for keys in ([3], [4], [5]):
                g, w = rt.groupby(ht, pred, keys, aggs, True), orc.groupby(ot, pred, keys, aggs, True)
                assert [[k.value for k in r.keys] for r in g] == [[k.value for k in r.keys] for r in w]
                for a, b in zip(g, w):
                    for x, y in zip(a.values, b.values):
                        assert x.is_null == y.is_null and (x.is_null or same_value(x.value, y.value) or (x.value == 0.0 and y.value == 0.0 and math.copysign(1, x.value) == math.copysign(1, y.value))), (keys, x, y)
                        if not x.is_null and isinstance(y.value, float) and y.value == 0.0:
                            assert math.copysign(1, x.value) == math.copysign(1, y.value), (keys, x, y)  # the sign of a zero result too


def test_join_key_lists_of_up_to_eight_pairs(rt, orc, abi):
    """The reference's generic path takes any number of key pairs (hash_join.rs:200-335); the kernels' key tuple holds eight (r03: four):
    six and eight pairs of mixed types, some with NULL cells, against the oracle; a ninth pair is handed back."""
    rng = np.random.default_rng(5)
    chunks, n_right = [20_000, 10_001], 9_000
    n_left = sum(chunks)
    def cols(n, first_fid):
        out = []
        for k in range(8):
            dt = (abi.DT_INT64, abi.DT_INT32, abi.DT_UINT32, abi.DT_INT64)[k % 4]
            vals = rng.integers(0, 3, size=n).astype({abi.DT_INT64: np.int64, abi.DT_INT32: np.int32, abi.DT_UINT32: np.uint32}[dt])
            out.append((first_fid + k, dt, vals, (rng.random(n) > 0.03) if k % 3 == 0 else None))
        return out
    tabs = _keyed_tables(rt, orc, abi, cols(n_left, 1), cols(n_right, 11), chunks, n_right)
    for n_pairs in (6, 8):
        keys = [(1 + k, 11 + k, k % 2 == 0) for k in range(n_pairs)]
        got = _same_join(rt, orc, tabs, keys, 4096, jts=("inner", "left", "semi", "anti"))
        assert sum(len(b[0]) for b in got) > 0
    with pytest.raises(abi.LlkvError) as e:
        rt.join_stream(tabs[0], tabs[1], [(1 + k % 8, 11 + k % 8, False) for k in range(9)], abi.JOIN_INNER, 4096)
    assert e.value.kind == "Unsupported"


def test_generic_key_types_match_oracle(rt, orc, abi):
    """Float64 keys by bit pattern, Utf8 keys through the two tables' dictionaries (with the "<NULL>" marker
    string), mismatched types (never equal), Date32 (key extraction fails: the row matches nothing)."""
    rng = np.random.default_rng(78)
    n_left, n_right = 70_001, 5_000
    words_l = ["a", "bb", "<NULL>", "ccc", "", None, "only-left"]
    words_r = ["bb", "ccc", None, "", "only-right", "a"]
    ls = [words_l[k] for k in rng.integers(0, len(words_l), size=n_left)]
    rs = [words_r[k] for k in rng.integers(0, len(words_r), size=n_right)]
    fl = rng.choice(np.array([0.0, -0.0, np.nan, 1.5, 2.5, -7.25]), size=n_left)
    fr = rng.choice(np.array([0.0, np.nan, 1.5, 3.5, -7.25]), size=n_right)
    il, ir = rng.integers(0, 50, size=n_left).astype(np.int32), rng.integers(0, 50, size=n_right).astype(np.int64)
    dl, dr = rng.integers(0, 5, size=n_left).astype(np.int32), rng.integers(0, 5, size=n_right).astype(np.int32)
    vl, vr = rng.random(n_left) > 0.1, rng.random(n_right) > 0.02
    tabs = _keyed_tables(rt, orc, abi,
                         [(1, abi.DT_UTF8, ls, None), (2, abi.DT_FLOAT64, fl, vl), (3, abi.DT_INT32, il, vl), (4, abi.DT_DATE32, dl, vl), (5, abi.DT_INT64, il.astype(np.int64), None)],
                         [(1, abi.DT_UTF8, rs, None), (2, abi.DT_FLOAT64, fr, vr), (3, abi.DT_INT64, ir, vr), (4, abi.DT_DATE32, dr, vr), (5, abi.DT_INT64, ir, None)],
                         [n_left], n_right)
    for null_eq in (False, True):
        _same_join(rt, orc, tabs, [(1, 1, null_eq)], 50_000, jts=("semi", "anti"))        # Utf8 ⋈ Utf8
        _same_join(rt, orc, tabs, [(2, 2, null_eq)], 50_000, jts=("semi", "anti"))        # Float64 bits
        _same_join(rt, orc, tabs, [(3, 3, null_eq)], 8192, jts=("inner", "left"))          # Int32 against Int64
        _same_join(rt, orc, tabs, [(4, 4, null_eq)], 8192, jts=("inner", "anti"))          # Date32
        _same_join(rt, orc, tabs, [(1, 3, null_eq)], 8192, jts=("inner",))                 # Utf8 against Int64 (only the marker meets NULLs)
        _same_join(rt, orc, tabs, [(3, 1, null_eq)], 8192, jts=("semi", "anti"))           # Int32 against Utf8
        _same_join(rt, orc, tabs, [(5, 5), (1, 1, null_eq)], 20_000, jts=("semi", "left"))  # Int64 + Utf8 composite
    with pytest.raises(abi.LlkvError) as e:
        rt.join_stream(tabs[0], tabs[1], [(5, 5)] * 9, JT["inner"])  # (up to eight pairs run since r04)
    assert e.value.kind == "Unsupported"


def test_sharded_probe_side_and_scans_concatenate(rt, abi):
    """SURVEY §8e for the selection-vector and join routes: a rank scans / probes the rows of its own chunks (row
    ids stay global), the build side of a join is replicated; the ranks' outputs concatenated in rank order are
    the single-GPU output."""
    rng = np.random.default_rng(17)
    chunks = [5000, 70_000, 300, 9000, 4096, 80_000, 123, 6000]
    n = sum(chunks)
    key = rng.integers(0, 3000, size=n).astype(np.int64)
    val = rng.normal(size=n)
    valid = rng.random(n) > 0.1
    n_right = 2000
    rkey = rng.integers(0, 4000, size=n_right).astype(np.int64)
    rtab = rt.HipTable(2, [n_right]); rtab.append_column(7, abi.DT_INT64, rkey)
    F, O = abi.Filter, abi.Operator
    pred = [F(2, O.GreaterThan(0.25))]

    def shard(rank, world):
        t = rt.HipTable(1, chunks, rank, world)
        lo = sum(chunks[:t.first_chunk])
        t.append_column(1, abi.DT_INT64, key[lo:lo + t.local_rows], valid=valid[lo:lo + t.local_rows])
        t.append_column(2, abi.DT_FLOAT64, val[lo:lo + t.local_rows])
        return t

    def outputs(t):
        pairs = rt.join_stream(t, rtab, [(1, 7)], JT["left"], 4096)
        ids = rt.filter_row_ids(t, pred).tolist()
        scan = rt.scan_stream(t, [1, 2], pred, include_nulls=True, include_row_ids=True)
        return ([x for b in pairs for x in b[0]], [x for b in pairs for x in b[1]], ids,
                [x for b in scan for x in b[1]], [x for b in scan for x in b[0][0]])

    want = outputs(shard(0, 1))
    assert len(want[0]) > n and len(want[2]) > 1000
    for world in (2, 4, 8):
        parts = [outputs(shard(r, world)) for r in range(world)]
        for k in range(5):
            assert [x for p in parts for x in p[k]] == want[k], (world, k)
    with pytest.raises(abi.LlkvError) as e:  # a sharded build side would join against a fraction of the table
        rt.join_stream(shard(0, 1), shard(1, 2), [(1, 1)], JT["inner"])
    assert e.value.kind == "InvalidArgumentError"


def test_columns_staged_from_arrow_arrays(rt, orc, abi):
    """llkv_hip_table_append_arrow_column: pyarrow arrays (sliced: non-zero offsets; NULLs; strings; decimals;
    booleans) staged through the Arrow C Data Interface give the same answers as the plain staging calls."""
    pa = pytest.importorskip("pyarrow")
    from decimal import Decimal
    rng = np.random.default_rng(29)
    chunks = [5000, 3, 4093]
    n = sum(chunks)
    i64 = rng.integers(-50, 50, size=n)
    f64 = rng.normal(size=n)
    d32 = rng.integers(8000, 8100, size=n).astype(np.int32)
    flag = rng.random(n) > 0.5
    tags = [("x", "yy", "", "zzz")[k] for k in rng.integers(0, 4, size=n)]
    dec = [int(v) for v in rng.integers(-10**9, 10**9, size=n)]
    v1, v5 = rng.random(n) > 0.2, rng.random(n) > 0.3
    pad = 3  # every chunk is a slice of a longer array: offsets 3, 3 + …
    def sliced(build):
        out, lo = [], 0
        for r in chunks:
            out.append(build(lo - pad if lo >= pad else None, lo, lo + r))
            lo += r
        return out
    def arr(values, typ, mask=None):
        def build(_, lo, hi):
            head = [values[lo]] * pad if hi > lo else [values[0]] * pad
            m = None if mask is None else np.concatenate([np.zeros(pad, bool), ~np.asarray(mask[lo:hi])])
            return pa.array(head + list(values[lo:hi]), type=typ, mask=m).slice(pad, hi - lo)
        return sliced(build)
    ht = rt.HipTable(1, chunks)
    ht.append_arrow_column(1, arr(i64.tolist(), pa.int64(), v1))
    ht.append_arrow_column(2, arr(f64.tolist(), pa.float64()))
    ht.append_arrow_column(3, arr(d32.tolist(), pa.date32()))
    ht.append_arrow_column(4, arr(tags, pa.string(), v5))
    ht.append_arrow_column(5, arr([Decimal(x).scaleb(-2) for x in dec], pa.decimal128(20, 2)))
    ht.append_arrow_column(6, arr(flag.tolist(), pa.bool_()))
    ot = orc.OracleTable(n).add(1, abi.DT_INT64, i64, list(v1)).add(2, abi.DT_FLOAT64, f64).add(3, abi.DT_DATE32, d32) \
                          .add(4, abi.DT_UTF8, [t if ok else None for t, ok in zip(tags, v5)]).add(5, abi.DT_DECIMAL128, dec, precision=20, scale=2)
    F, O, A, col = abi.Filter, abi.Operator, abi.AggregateSpec, abi.col
    pred = [F(3, O.GreaterThanOrEquals(8050))]
    aggs = [A.count_star(), A.count(1), A.sum(1), A.sum(col(2) * 2.0), A.sum(5), A.min(1)]
    assert [same_value(g.value, w.value, REL) for g, w in zip(rt.aggregate(ht, pred, aggs), orc.aggregate(ot, pred, aggs))] == [True] * len(aggs)
    got = rt.scan_stream(ht, [1, 4, 6], pred, include_nulls=True, include_row_ids=True)
    want = orc.scan_stream(ot, [1, 4], pred, include_nulls=True, include_row_ids=True)
    assert [x for b in got for x in b[1]] == [x for b in want for x in b[1]]
    assert [x for b in got for x in b[0][0]] == [x for b in want for x in b[0][0]]
    assert [x for b in got for x in b[0][1]] == [x for b in want for x in b[0][1]]
    ids = [x for b in got for x in b[1]]
    assert [bool(x) for b in got for x in b[0][2]] == [bool(flag[i]) for i in ids]


def test_scan_batches_as_arrow_record_batches(rt, abi):
    """scan_stream → llkv_hip_batch_export_arrow → pyarrow: the batches outlive the callback and carry the same cells
    (values, NULLs, strings, row ids) as the raw views."""
    pa = pytest.importorskip("pyarrow")
    rng = np.random.default_rng(23)
    n = 150_000
    i64 = rng.integers(-1000, 1000, size=n).astype(np.int64)
    f64 = rng.normal(size=n)
    tags = [("x", "yy", "", "zzz")[k] for k in rng.integers(0, 4, size=n)]
    v1 = rng.random(n) > 0.2
    t = rt.HipTable(1, [n])
    t.append_column(1, abi.DT_INT64, i64, valid=v1)
    t.append_column(2, abi.DT_FLOAT64, f64)
    t.append_utf8_column(3, tags)
    pred = [abi.Filter(2, abi.Operator.GreaterThan(0.0))]
    kept = []
    rt.scan_stream(t, [1, 2, 3, abi.col(2) * 2.0], pred, include_nulls=True, include_row_ids=True, consume=lambda b: kept.append(rt.batch_to_arrow(b, ["a", "b", "s", "b2"])))
    raw = rt.scan_stream(t, [1, 2, 3, abi.col(2) * 2.0], pred, include_nulls=True, include_row_ids=True)
    assert [rb.num_rows for rb in kept] == [len(b[1]) for b in raw] and len(kept) >= 2
    tab = pa.Table.from_batches(kept)
    assert tab.column("a").to_pylist() == [x for b in raw for x in b[0][0]]
    assert tab.column("b").to_pylist() == [x for b in raw for x in b[0][1]]
    assert tab.column("s").to_pylist() == [x for b in raw for x in b[0][2]]
    assert tab.column("b2").to_pylist() == [x for b in raw for x in b[0][3]]
    assert tab.column("rowid").to_pylist() == [x for b in raw for x in b[1]]
    assert tab.column("a").null_count == int((~v1 & (f64 > 0.0)).sum())


def test_join_and_scan_edge_sizes(rt, orc, abi):
    """Empty sides, and selections that end exactly on / just past the 16-window device buffers."""
    X = abi.JOIN_KEYS_EXECUTOR
    k3 = np.array([1, 2, 2], dtype=np.int64)
    none = np.zeros(0, dtype=np.int64)
    for lv, rv in ((none, k3), (k3, none), (none, none)):
        lt = rt.HipTable(1, [len(lv)]); lt.append_column(1, abi.DT_INT64, lv)
        rtab = rt.HipTable(2, [len(rv)]); rtab.append_column(7, abi.DT_INT64, rv)
        ol, orr = orc.OracleTable(len(lv)).add(1, abi.DT_INT64, lv), orc.OracleTable(len(rv)).add(7, abi.DT_INT64, rv)
        for jt in ("inner", "left", "semi", "anti"):
            assert rt.join_stream(lt, rtab, [(1, 7)], JT[jt]) == orc.hash_join(ol, orr, [(1, 7)], JT[jt]), (len(lv), len(rv), jt)
            assert rt.join_stream(lt, rtab, [(1, 7), (1, 7)], JT[jt]) == orc.hash_join(ol, orr, [(1, 7), (1, 7)], JT[jt])
        for jt in ("inner", "left"):
            assert rt.join_stream(lt, rtab, [(1, 7)], JT[jt], key_rules=X) == orc.hash_join(ol, orr, [(1, 7)], JT[jt], key_rules=X)
    for n in (16 * 65536, 16 * 65536 + 1, 65536, 65537):
        v = np.arange(n, dtype=np.int64)
        t = rt.HipTable(1, [n]); t.append_column(1, abi.DT_INT64, v)
        seen = []
        rt.scan_stream(t, [1], None, include_row_ids=True, consume=lambda b: seen.append((int(b.num_rows), int(b.row_ids[0]), int(b.row_ids[int(b.num_rows) - 1]))))
        want = [(min(65536, n - w0), w0, min(n, w0 + 65536) - 1) for w0 in range(0, n, 65536)]
        assert seen == want, n
        assert rt.filter_row_ids(t, [abi.Filter(1, abi.Operator.GreaterThanOrEquals(5))], count_only=True) == n - 5
    # join → GROUP BY → top-k: no qualifying dim row, no matching fact row, LIMIT 0, LIMIT beyond the groups
    dim = rt.HipTable(2, [4]); dim.append_column(1, abi.DT_INT64, np.array([10, 20, 30, 40])); dim.append_column(2, abi.DT_INT64, np.array([1, 2, 3, 4]))
    fact = rt.HipTable(1, [5]); fact.append_column(7, abi.DT_INT64, np.array([20, 20, 40, 50, 20])); fact.append_column(8, abi.DT_FLOAT64, np.array([1.5, 2.5, 4.0, 9.0, 1.0]))
    F, O, col = abi.Filter, abi.Operator, abi.col
    assert rt.join_groupby_topk(fact, [], 7, dim, [F(2, O.GreaterThan(100))], 1, col(8) * 1.0, payload_fields=[2], limit=3) == ([], 0)
    assert rt.join_groupby_topk(fact, [F(8, O.GreaterThan(100.0))], 7, dim, [], 1, col(8) * 1.0, payload_fields=[2], limit=3) == ([], 0)
    assert rt.join_groupby_topk(fact, [], 7, dim, [], 1, col(8) * 1.0, payload_fields=[2], limit=0) == ([], 2)
    assert rt.join_groupby_topk(fact, [], 7, dim, [], 1, col(8) * 1.0, payload_fields=[2], limit=9) == ([(20, 5.0, 3, 2), (40, 4.0, 1, 4)], 2)


def test_repeated_statements_do_not_leak_device_memory(rt, abi, tpch):
    """Statement after statement (prepare → run → close, scans, joins, the Q3 pipeline, a sort-based GROUP BY): the
    pools settle, after which the free device memory no longer moves."""
    import torch
    n = 200_000
    d = tpch.gen_lineitem(n, 0.05)
    t = rt.HipTable(1, tpch.chunk_rows(n, 65536))
    for c, (fid, dt) in tpch.LINEITEM_SCHEMA.items():
        t.append_utf8_column(fid, d[c]) if dt == abi.DT_UTF8 else t.append_column(fid, dt, d[c])
    keys = rt.HipTable(2, [5000]); keys.append_column(1, abi.DT_INT64, np.unique(d["l_orderkey"])[:5000])
    q1, q6 = tpch.q1(), tpch.q6()
    A, col = abi.AggregateSpec, abi.col

    def round_trip():
        for q in (q1, q6):
            pq = rt.PreparedQuery(t, q.predicate, q.aggs, q.keys, q.order_by_keys)
            pq.run()
            pq.close()
        rt.filter_row_ids(t, q6.predicate)
        rt.scan_stream(t, [tpch.L_QUANTITY], q6.predicate, consume=lambda b: None)
        rt.join_stream(t, keys, [(tpch.L_ORDERKEY, 1)], JT["semi"], consume=lambda k: None)
        rt.join_groupby_topk(t, [], tpch.L_ORDERKEY, keys, [], 1, col(tpch.L_EXTENDEDPRICE) * 1.0, limit=5)
        pq = rt.PreparedQuery(t, None, [A.count_star(), A.sum(tpch.L_QUANTITY)], [tpch.L_PARTKEY], True)
        pq.launch(0)
        rt.check(rt.lib().llkv_hip_query_finish(pq._h, None))
        pq.close()

    for _ in range(5):
        round_trip()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    pinned0 = rt.pinned_stats()
    for _ in range(30):
        round_trip()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] >= free0 - (8 << 20)  # nothing grows with the statement count
    # … nor does the page-locked host memory: what the statements borrowed rests in the cache again, nothing is still handed out
    cached, outstanding = rt.pinned_stats()
    assert outstanding == pinned0[1] and cached <= pinned0[0] + (8 << 20), (pinned0, (cached, outstanding))


def test_large_id_vectors_come_in_recycled_pinned_blocks(rt, abi):
    """An id vector above the 256 MB block classes (here 40 M rows = 320 MB): the block is page-locked once, rests in the library's
    cache after llkv_hip_free and serves the next call — no hipHostMalloc / hipHostFree of hundreds of MB per statement."""
    n = 40_000_000
    t = rt.HipTable(1, [n])
    t.append_column(1, abi.DT_INT64, np.zeros(n, dtype=np.int64))
    c0, o0 = rt.pinned_stats()
    ids = rt.filter_row_ids(t, None)
    assert len(ids) == n and int(ids[0]) == 0 and int(ids[-1]) == n - 1
    del ids
    c1, o1 = rt.pinned_stats()
    assert o1 == o0 and c1 >= c0 + n * 8 - (4 << 20) or c1 >= n * 8, (c0, o0, c1, o1)  # the block went back into the cache
    import time
    t0 = time.perf_counter()
    assert rt.filter_row_ids(t, None, count_only=True) == n  # (the ids reach host memory; the binding does not copy them on)
    dt = time.perf_counter() - t0
    c2, o2 = rt.pinned_stats()
    assert (c2, o2) == (c1, o1)   # the same block again
    assert dt < 0.2, dt           # (320 MB at the link's rate is ~7 ms and pinning them anew ~25 ms more: boxes differ too much for a tighter bound — the statistics above are the assertion)
    t.close()


def test_join_with_exploding_match_counts_shrinks_its_steps(rt, abi):
    """Every probe row matches 1 000 build rows: 3 × 10⁸ pairs.  A 2 M-row device step would need 32 GB of pair
    buffers; the steps shrink until their pairs fit 1 GiB, and the batches still follow the reference's rule
    (≥ batch_size pairs after a probe row, and the end of every 65 536-row scan batch)."""
    nl, nr, batch = 300_000, 1000, 65536
    lt = rt.HipTable(1, [nl]); lt.append_column(1, abi.DT_INT64, np.ones(nl, dtype=np.int64))
    rtab = rt.HipTable(2, [nr]); rtab.append_column(7, abi.DT_INT64, np.ones(nr, dtype=np.int64))
    sizes = []
    rt.join_stream(lt, rtab, [(1, 7)], JT["inner"], batch, consume=sizes.append)
    want, acc = [], 0
    for row in range(nl):
        acc += nr
        if acc >= batch or (row + 1) % 65536 == 0 or row + 1 == nl:
            want.append(acc)
            acc = 0
    assert sum(sizes) == nl * nr and sizes == want


def test_executor_rule_joins_match_oracle(rt, orc, abi):
    """llkv_join_options.key_rules = EXECUTOR: the SQL joins of the executor (normalised keys, arrow-row equality,
    INNER / LEFT, no batch structure) — llkv-executor/src/lib.rs:12218-12581."""
    rng = np.random.default_rng(91)
    n_left, n_right = 150_000, 30_000
    X = abi.JOIN_KEYS_EXECUTOR
    cols_l = [(1, abi.DT_INT32, rng.integers(-50, 500, size=n_left).astype(np.int32), rng.random(n_left) > 0.05),
              (2, abi.DT_UINT64, rng.choice(np.array([3, 7, 2**63, 2**64 - 1, 100], dtype=np.uint64), size=n_left), None),
              (3, abi.DT_FLOAT32, rng.choice(np.array([0.5, 0.1, -0.0, 0.0, 7.25], dtype=np.float32), size=n_left), None),
              (4, abi.DT_DATE32, rng.integers(0, 40, size=n_left).astype(np.int32), None),
              (5, abi.DT_UTF8, [("x", "yy", "zzz", None)[k] for k in rng.integers(0, 4, size=n_left)], None)]
    cols_r = [(1, abi.DT_INT64, rng.integers(-50, 500, size=n_right).astype(np.int64), rng.random(n_right) > 0.05),
              (2, abi.DT_INT64, rng.choice(np.array([3, 7, -1, 100, -2**63], dtype=np.int64), size=n_right), None),
              (3, abi.DT_FLOAT64, rng.choice(np.array([0.5, 0.1, float(np.float32(0.1)), 0.0, 7.25]), size=n_right), None),
              (4, abi.DT_DATE32, rng.integers(0, 40, size=n_right).astype(np.int32), None),
              (5, abi.DT_UTF8, [("yy", "x", "w", None)[k] for k in rng.integers(0, 4, size=n_right)], None)]
    lt, rtab, ol, orr = _keyed_tables(rt, orc, abi, cols_l, cols_r, [70_000, 80_000], n_right)

    def same(keys, jt):
        got = rt.join_stream(lt, rtab, keys, JT[jt], 8192, key_rules=X)
        want = orc.hash_join(ol, orr, keys, JT[jt], 8192, key_rules=X)
        assert [x for b in got for x in b[0]] == [x for b in want for x in b[0]], (keys, jt)
        assert [x for b in got for x in b[1]] == [x for b in want for x in b[1]], (keys, jt)
        return sum(len(b[0]) for b in got)

    assert same([(1, 1)], "inner") > 0                 # Int32 meets Int64
    same([(1, 1)], "left")
    assert same([(1, 1), (4, 4)], "inner") > 0         # + Date32
    assert same([(1, 1), (5, 5)], "left") > n_left     # + Utf8 through the two dictionaries
    assert same([(1, 1), (3, 3)], "inner") > 0         # + Float32 against Float64
    assert same([(1, 1), (2, 2)], "inner") > 0         # + UInt64 against Int64: 2^63 and 2^64-1 are NULL after the cast
    assert same([(4, 1)], "inner") == 0                # Date32 never meets Int64
    for jt in ("semi", "anti"):
        for m, args in ((rt.join_stream, (lt, rtab)), (orc.hash_join, (ol, orr))):
            with pytest.raises(abi.LlkvError) as e:
                m(*args, [(1, 1)], JT[jt], 8192, key_rules=X)
            assert e.value.kind == "Internal"


@pytest.mark.parametrize("key_images", [False, True])
@pytest.mark.parametrize("rows,scale", [(60175, 0.01), (600_000, 0.1)])
def test_q3_join_groupby_topk_matches_oracle(rt, orc, abi, tpch, rows, scale, key_images, monkeypatch):
    """TPC-H Q3 shape (BASELINE.json configs[4], single GPU): customer(segment) ⋉ orders(date) ⋈ lineitem(shipdate),
    GROUP BY l_orderkey, o_orderdate, o_shippriority, SUM(price*(1-disc)), ORDER BY revenue DESC, o_orderdate LIMIT 10.
    The oracle side composes the restated operators in the executor's order (join → mask → group-by → sort → limit);
    sums add an order's lineitems in scan order on both sides, so revenue is compared bit for bit."""
    if key_images:  # the scans read 4-byte images of the Int64 key columns (KeyImage, csrc/engine.hpp): at SF10 they do by themselves
        monkeypatch.setenv("LLKV_HIP_KEY_IMAGE_MIN_ROWS", "1")
    D = tpch.DATE_1995_03_15
    li = tpch.gen_lineitem(rows, scale)
    n_ord = tpch.orders_for_lineitems(rows)
    od = tpch.gen_orders(n_ord, scale)
    n_cust = tpch.customers_for_scale(scale)
    cu = tpch.gen_customer(n_cust, scale)
    seg = [tpch.SEGMENTS[c] for c in cu["c_mktsegment"]]

    # ---- GPU
    lt = rt.HipTable(1, tpch.chunk_rows(rows, 65536))
    for c in ("l_orderkey", "l_shipdate", "l_extendedprice", "l_discount"):
        lt.append_column(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], li[c])
    ot_ = rt.HipTable(2, tpch.chunk_rows(n_ord, 65536))
    for c, (fid, dt) in tpch.ORDERS_SCHEMA.items():
        ot_.append_column(fid, dt, od[c])
    ct = rt.HipTable(3, tpch.chunk_rows(n_cust, 65536))
    ct.append_column(tpch.C_CUSTKEY, abi.DT_INT64, cu["c_custkey"])
    ct.append_utf8_column(tpch.C_MKTSEGMENT, seg)
    F, O, col = abi.Filter, abi.Operator, abi.col
    revenue = col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT))
    got, total = rt.join_groupby_topk(
        lt, [F(tpch.L_SHIPDATE, O.GreaterThan(D))], tpch.L_ORDERKEY,
        ot_, [F(tpch.O_ORDERDATE, O.LessThan(D))], tpch.O_ORDERKEY, revenue,
        payload_fields=[tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], limit=10,
        dim_fk=tpch.O_CUSTKEY, dim2=ct, dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))], dim2_key=tpch.C_CUSTKEY)
    assert (lt.key_images()[0], ot_.key_images()[0]) == ((1, 2) if key_images else (0, 0))

    # ---- oracle composition
    cust_t = orc.OracleTable(n_cust).add(tpch.C_CUSTKEY, abi.DT_INT64, cu["c_custkey"]).add(tpch.C_MKTSEGMENT, abi.DT_UTF8, seg)
    c_rows = orc.filter_row_ids(cust_t, [F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))])
    custkeys = cu["c_custkey"][c_rows.astype(np.int64)]
    # customer ⋈ orders (inner, key custkey): the orders whose customer qualifies, in order-row order
    o_tab = orc.OracleTable(n_ord).add(tpch.O_CUSTKEY, abi.DT_INT64, od["o_custkey"])
    c_tab = orc.OracleTable(len(custkeys)).add(tpch.C_CUSTKEY, abi.DT_INT64, custkeys)
    o_sel = np.array([l for b in orc.hash_join(o_tab, c_tab, [(tpch.O_CUSTKEY, tpch.C_CUSTKEY)], abi.JOIN_SEMI) for l in b[0]], dtype=np.int64)
    # ⋈ lineitem on orderkey
    l_tab = orc.OracleTable(rows).add(tpch.L_ORDERKEY, abi.DT_INT64, li["l_orderkey"])
    oj = orc.OracleTable(len(o_sel)).add(tpch.O_ORDERKEY, abi.DT_INT64, od["o_orderkey"][o_sel])
    pairs = orc.hash_join(l_tab, oj, [(tpch.L_ORDERKEY, tpch.O_ORDERKEY)], abi.JOIN_INNER)
    pl = np.array([x for b in pairs for x in b[0]], dtype=np.int64)
    po = o_sel[np.array([x for b in pairs for x in b[1]], dtype=np.int64)]
    # remaining WHERE conjuncts as a mask over the joined rows (llkv-executor/src/lib.rs:1629-1646)
    m = (od["o_orderdate"][po] < D) & (li["l_shipdate"][pl] > D)
    pl, po = pl[m], po[m]
    j = orc.OracleTable(len(pl))
    j.add(1, abi.DT_INT64, li["l_orderkey"][pl]).add(2, abi.DT_DATE32, od["o_orderdate"][po]).add(3, abi.DT_INT64, od["o_shippriority"][po])
    j.add(4, abi.DT_FLOAT64, li["l_extendedprice"][pl]).add(5, abi.DT_FLOAT64, li["l_discount"][pl])
    groups = orc.groupby(j, None, [1, 2, 3], [abi.AggregateSpec.sum(col(4) * (1 - col(5))), abi.AggregateSpec.count_star()])
    want = sorted(((g.keys[0].value, g.values[0].value, g.values[1].value, g.keys[1].value, g.keys[2].value) for g in groups),
                  key=lambda r: (-r[1], r[3]))[:10]
    assert total == len(groups)
    assert len(got) == len(want) == 10
    for g, w in zip(got, want):
        assert g[0] == w[0] and g[2] == w[2] and g[3] == w[3] and g[4] == w[4], (g, w)
        assert np.float64(g[1]).tobytes() == np.float64(w[1]).tobytes(), (g, w)  # bit-exact revenue


def test_late_materialisation_and_its_eager_forms_give_the_same_bits(rt, abi, tpch, monkeypatch):
    """Argument-only columns read late (Plan::EARLY, ProbePlan EARLY / KEYBIT) and the top-k selection's slice winners taken from
    the run sums are the SAME computation as their eager forms — same rows, same order of additions: every switch leaves
    every bit of Q6, of a selective two-aggregate scan and of the Q3 pipeline where it was."""
    rows, scale = 300_000, 0.05
    li = tpch.gen_lineitem(rows, scale)
    n_ord = tpch.orders_for_lineitems(rows)
    od = tpch.gen_orders(n_ord, scale)
    n_cust = tpch.customers_for_scale(scale)
    cu = tpch.gen_customer(n_cust, scale)
    lt = rt.HipTable(1, tpch.chunk_rows(rows, 65536))
    for c in ("l_orderkey", "l_shipdate", "l_quantity", "l_extendedprice", "l_discount"):
        lt.append_column(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], li[c])
    ot_ = rt.HipTable(2, tpch.chunk_rows(n_ord, 65536))
    for c, (fid, dt) in tpch.ORDERS_SCHEMA.items():
        ot_.append_column(fid, dt, od[c])
    ct = rt.HipTable(3, tpch.chunk_rows(n_cust, 65536))
    ct.append_column(tpch.C_CUSTKEY, abi.DT_INT64, cu["c_custkey"])
    ct.append_utf8_column(tpch.C_MKTSEGMENT, [tpch.SEGMENTS[c] for c in cu["c_mktsegment"]])
    A, F, O, B, col = abi.AggregateSpec, abi.Filter, abi.Operator, abi.Bound, abi.col
    D = tpch.DATE_1995_03_15
    price, disc, qty = tpch.L_EXTENDEDPRICE, tpch.L_DISCOUNT, tpch.L_QUANTITY
    q6 = tpch.q6()
    scans = [(q6.predicate, q6.aggs), ([F(qty, O.LessThan(3))], [A.sum(price), A.sum(col(price) * col(disc)), A.min(price), A.count_star()]),
             ([F(qty, O.LessThan(60))], [A.sum(price), A.avg(disc)])]

    def run_all():
        out, sigs = [], []
        for pred, aggs in scans:
            q = rt.PreparedQuery(lt, pred, aggs)
            sigs.append(q.kernel_signature)
            out.append([(v.dtype, v.is_null, np.float64(v.value).tobytes() if isinstance(v.value, float) else v.value) for v in q.run()[0].values])
            q.close()
        top, total = rt.join_groupby_topk(lt, [F(tpch.L_SHIPDATE, O.GreaterThan(D))], tpch.L_ORDERKEY, ot_, [F(tpch.O_ORDERDATE, O.LessThan(D))], tpch.O_ORDERKEY,
                                          col(price) * (1 - col(disc)), payload_fields=[tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], limit=10, dim_fk=tpch.O_CUSTKEY, dim2=ct,
                                          dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))], dim2_key=tpch.C_CUSTKEY)
        out.append([(g[0], np.float64(g[1]).tobytes(), g[2], g[3], g[4]) for g in top] + [total])
        return out, sigs

    base, sigs = run_all()
    assert all(s.endswith(",0,1,%d>" % n) for s, n in zip(sigs, (3, 1, 1))), sigs  # the late form is what runs by default
    assert base[3][-1] > 0 and len(base[3]) == 11
    assert lt.key_images()[0] == ot_.key_images()[0] == 0  # (tables of this size read their key columns as they are)
    for switch in ("LLKV_HIP_SCAN_NO_LATE", "LLKV_HIP_JOIN_NO_LATE", "LLKV_HIP_JOIN_PROBE_RANKS", "LLKV_HIP_TOPK_TWO_LAUNCHES", "LLKV_HIP_KEY_IMAGE_MIN_ROWS"):
        monkeypatch.setenv(switch, "1")
        if switch == "LLKV_HIP_SCAN_NO_LATE":
            monkeypatch.setenv("LLKV_HIP_TILE_ROWS", "8192")  # (the tile is the unit of the reduction: the late form's tile length)
        got, sigs2 = run_all()
        monkeypatch.delenv(switch)
        monkeypatch.delenv("LLKV_HIP_TILE_ROWS", raising=False)
        assert got == base, switch
        if switch == "LLKV_HIP_SCAN_NO_LATE":
            assert all(s.endswith(",0>") for s in sigs2), sigs2
        if switch == "LLKV_HIP_KEY_IMAGE_MIN_ROWS":  # the 4-byte key images: l_orderkey; o_orderkey and o_custkey
            assert lt.key_images()[0] == 1 and lt.key_images()[1] >= rows * 4 and ot_.key_images()[0] == 2


@pytest.mark.parametrize("n_orders,limit", [(1500, 10), (40_000, 10), (40_000, 300)])
def test_join_groupby_topk_with_tied_sums(rt, abi, n_orders, limit):
    """Top-k by selection (slice winners → threshold → candidates → exact host order): thousands of groups share
    a handful of sums, so the LIMIT cut falls inside runs of equal sums and equal dates; order = sum DESC,
    payload[0] ASC, then dim row.  Expected values computed with numpy from the same inputs (sums of ≤ 3 small
    integers-as-f64 are exact in any order)."""
    rng = np.random.default_rng(n_orders + limit)
    okey = np.arange(1, n_orders + 1, dtype=np.int64) * 3
    odate = rng.integers(9000, 9004, size=n_orders).astype(np.int32)
    lines = rng.integers(1, 4, size=n_orders)
    lkey = np.repeat(okey, lines)
    price = rng.choice(np.array([100.0, 200.0, 300.0]), size=len(lkey))
    price[rng.random(len(lkey)) < 0.0005] = 1e6  # a few clear winners
    orphan = rng.random(len(lkey)) < 0.05        # fact rows without an order
    lkey = np.where(orphan, lkey + 1, lkey)
    ot_ = rt.HipTable(2, [n_orders]); ot_.append_column(1, abi.DT_INT64, okey); ot_.append_column(2, abi.DT_DATE32, odate)
    lt = rt.HipTable(1, [len(lkey)]); lt.append_column(7, abi.DT_INT64, lkey); lt.append_column(8, abi.DT_FLOAT64, price)
    got, total = rt.join_groupby_topk(lt, [], 7, ot_, [abi.Filter(2, abi.Operator.LessThan(9003))], 1, abi.col(8) * 1.0, payload_fields=[2], limit=limit)
    sums, counts = np.zeros(n_orders), np.zeros(n_orders, dtype=np.int64)
    idx = (lkey // 3 - 1)[~orphan]
    np.add.at(sums, idx, price[~orphan]); np.add.at(counts, idx, 1)
    live = np.flatnonzero((counts > 0) & (odate < 9003))
    order = sorted(live.tolist(), key=lambda i: (-sums[i], odate[i], i))[:limit]
    assert total == len(live)
    assert [(r[0], r[1], r[2], r[3]) for r in got] == [(int(okey[i]), float(sums[i]), int(counts[i]), int(odate[i])) for i in order]


@pytest.mark.parametrize("rows_per_group", [1, 7, 60, 700, 5000])
def test_join_groupby_sums_every_group_left_to_right_across_stripes(rt, abi, rows_per_group, monkeypatch):
    """Every group of the join → GROUP BY pipeline, not only the top few: 200 groups whose fact rows are runs of
    1 … 5 000 rows, so runs end inside a probe stripe (2 048 rows of a tile), span several stripes and whole tiles.
    The sums are the reference's sequential f64 chain per group (SumFloat64: 0.0 then += in row order,
    llkv-aggregate/src/lib.rs:870-888) — compared bit for bit with a sequential numpy accumulate, through the sums
    taken straight from the stripes and through the compacted pairs."""
    rng = np.random.default_rng(rows_per_group)
    n_groups = 200
    sizes = np.maximum(1, rng.integers(rows_per_group // 2, rows_per_group * 3 // 2 + 1, size=n_groups))
    okey = np.arange(1, n_groups + 1, dtype=np.int64) * 5
    lkey = np.repeat(okey, sizes)
    hole = rng.random(len(lkey)) < 0.1  # fact rows that do not pass the filter: the stripes hold fewer pairs than rows
    price = rng.normal(1000.0, 300.0, size=len(lkey))
    flag = np.where(hole, 0, 1).astype(np.int64)
    ot_ = rt.HipTable(2, [n_groups]); ot_.append_column(1, abi.DT_INT64, okey); ot_.append_column(2, abi.DT_DATE32, np.full(n_groups, 9000, dtype=np.int32))
    lt = rt.HipTable(1, tpch_chunks(len(lkey))); lt.append_column(7, abi.DT_INT64, lkey); lt.append_column(8, abi.DT_FLOAT64, price); lt.append_column(9, abi.DT_INT64, flag)
    want = {}
    lo = 0
    for k, sz in zip(okey, sizes):
        v = price[lo:lo + sz][~hole[lo:lo + sz]]
        if len(v):
            want[int(k)] = (np.add.accumulate(np.concatenate([[0.0], v]))[-1], len(v))  # 0.0 + v0 + v1 + … left to right
        lo += sz
    for env in (None, "LLKV_HIP_JOIN_COMPACT"):
        if env:
            monkeypatch.setenv(env, "1")
        got, total = rt.join_groupby_topk(lt, [abi.Filter(9, abi.Operator.Equals(1))], 7, ot_, [], 1, abi.col(8) * 1.0, payload_fields=[2], limit=256)
        assert total == len(want) == len(got)
        assert {r[0]: (np.float64(r[1]).tobytes(), r[2]) for r in got} == {k: (np.float64(s).tobytes(), c) for k, (s, c) in want.items()}


def tpch_chunks(n, chunk=131072):
    return [min(chunk, n - lo) for lo in range(0, n, chunk)] or [0]


@pytest.mark.parametrize("switches", [("LLKV_HIP_JOIN_HASH",), ("LLKV_HIP_JOIN_SORT",), ("LLKV_HIP_TOPK_SORT",), ("LLKV_HIP_TOPK_SORT", "LLKV_HIP_TOPK_FULL"),
                                      ("LLKV_HIP_JOIN_HASH", "LLKV_HIP_JOIN_SORT", "LLKV_HIP_TOPK_SORT"), ("LLKV_HIP_SELECT_TWO_PASS",),
                                      ("LLKV_HIP_JOIN_UNSORTED",), ("LLKV_HIP_JOIN_UNSORTED", "LLKV_HIP_JOIN_NO_SINK"), ("LLKV_HIP_READBACK_SYNC",),
                                      ("LLKV_HIP_JOIN_COMPACT",), ("LLKV_HIP_JOIN_COMPACT", "LLKV_HIP_JOIN_SORT"),
                                      ("LLKV_HIP_JOIN_LISTED",), ("LLKV_HIP_JOIN_LISTED", "LLKV_HIP_JOIN_SORT"), ("LLKV_HIP_JOIN_RANK_SCAN",),
                                      ("LLKV_HIP_JOIN_RANK_SCAN", "LLKV_HIP_TOPK_SORT"), ("LLKV_HIP_JOIN_SORT", "LLKV_HIP_JOIN_COMPACT", "LLKV_HIP_TOPK_SORT")])
def test_join_pipeline_fallback_forms_give_the_same_rows(rt, abi, tpch, monkeypatch, switches):
    """The join → aggregate pipeline has a general form behind every shortcut (hash table behind bitmap + rank, pair
    sort behind run sums, radix-sort top-k behind the threshold selection): forced through the switches of DESIGN §10
    they return the same rows, bit for bit."""
    rows, scale = 120_000, 0.02
    D = tpch.DATE_1995_03_15
    li = tpch.gen_lineitem(rows, scale)
    n_ord = tpch.orders_for_lineitems(rows)
    od = tpch.gen_orders(n_ord, scale)
    n_cust = tpch.customers_for_scale(scale)
    cu = tpch.gen_customer(n_cust, scale)
    lt = rt.HipTable(1, tpch.chunk_rows(rows, 32768))
    for c in ("l_orderkey", "l_shipdate", "l_extendedprice", "l_discount"):
        lt.append_column(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], li[c])
    ot_ = rt.HipTable(2, [n_ord])
    for c, (fid, dt) in tpch.ORDERS_SCHEMA.items():
        ot_.append_column(fid, dt, od[c])
    ct = rt.HipTable(3, [n_cust])
    ct.append_column(tpch.C_CUSTKEY, abi.DT_INT64, cu["c_custkey"])
    ct.append_utf8_column(tpch.C_MKTSEGMENT, [tpch.SEGMENTS[c] for c in cu["c_mktsegment"]])
    F, O, col = abi.Filter, abi.Operator, abi.col
    run = lambda: rt.join_groupby_topk(lt, [F(tpch.L_SHIPDATE, O.GreaterThan(D))], tpch.L_ORDERKEY, ot_, [F(tpch.O_ORDERDATE, O.LessThan(D))], tpch.O_ORDERKEY,
                                       col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT)), payload_fields=[tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], limit=10,
                                       dim_fk=tpch.O_CUSTKEY, dim2=ct, dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))], dim2_key=tpch.C_CUSTKEY)
    bits = lambda res: ([(r[0], np.float64(r[1]).tobytes(), r[2], r[3], r[4]) for r in res[0]], res[1])
    want = bits(run())
    assert len(want[0]) == 10
    for name in switches:
        monkeypatch.setenv(name, "1")
    assert bits(run()) == want


def test_q3_chain_with_the_dimension_out_of_key_order_and_with_a_repeated_key(rt, abi, tpch):
    """The orders bitmap is filled while the orders selection is compacted; a selection in key order makes a key's rank
    among the set bits its group id (no rank → row table).  Orders in a random row order take the general form and
    return the same rows; an order key that occurs twice among the selected rows is reported (the bitmap then holds
    fewer bits than the selection has rows)."""
    rows, scale = 150_000, 0.025
    D = tpch.DATE_1995_03_15
    li = tpch.gen_lineitem(rows, scale)
    n_ord = tpch.orders_for_lineitems(rows)
    od = tpch.gen_orders(n_ord, scale)
    n_cust = tpch.customers_for_scale(scale)
    cu = tpch.gen_customer(n_cust, scale)
    lt = rt.HipTable(1, tpch.chunk_rows(rows, 32768))
    for c in ("l_orderkey", "l_shipdate", "l_extendedprice", "l_discount"):
        lt.append_column(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], li[c])
    ct = rt.HipTable(3, [n_cust])
    ct.append_column(tpch.C_CUSTKEY, abi.DT_INT64, cu["c_custkey"])
    ct.append_utf8_column(tpch.C_MKTSEGMENT, [tpch.SEGMENTS[c] for c in cu["c_mktsegment"]])
    F, O, col = abi.Filter, abi.Operator, abi.col

    def orders_table(cols, chunk):
        t = rt.HipTable(2, tpch.chunk_rows(len(cols["o_orderkey"]), chunk))
        for c, (fid, dt) in tpch.ORDERS_SCHEMA.items():
            t.append_column(fid, dt, cols[c])
        return t

    def run(ot_):
        return rt.join_groupby_topk(lt, [F(tpch.L_SHIPDATE, O.GreaterThan(D))], tpch.L_ORDERKEY, ot_, [F(tpch.O_ORDERDATE, O.LessThan(D))], tpch.O_ORDERKEY,
                                    col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT)), payload_fields=[tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], limit=10,
                                    dim_fk=tpch.O_CUSTKEY, dim2=ct, dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))], dim2_key=tpch.C_CUSTKEY)

    bits = lambda res: ([(r[0], np.float64(r[1]).tobytes(), r[2], r[3], r[4]) for r in res[0]], res[1])
    want = bits(run(orders_table(od, 8192)))
    assert len(want[0]) == 10
    perm = np.random.default_rng(5).permutation(n_ord)
    assert bits(run(orders_table({c: v[perm] for c, v in od.items()}, 8192))) == want
    # one swap inside a stripe, one across two stripes (rows 2047 | 2048 are neighbours of different waves' stripes)
    for a, b in ((10, 11), (2047, 2048)):
        sw = {c: v.copy() for c, v in od.items()}
        for v in sw.values():
            v[[a, b]] = v[[b, a]]
        assert bits(run(orders_table(sw, 8192))) == want
    # a selected order twice
    sel = np.flatnonzero((od["o_orderdate"] < D) & np.isin(od["o_custkey"], cu["c_custkey"][np.array(cu["c_mktsegment"]) == tpch.SEGMENTS.index("BUILDING")]))
    twice = {c: np.concatenate([v, v[sel[:1]]]) for c, v in od.items()}
    with pytest.raises(abi.LlkvError) as e:
        run(orders_table(twice, 8192))
    assert "not unique" in str(e.value)


@pytest.mark.parametrize("chunks", [[9], [4096, 4097, 5], [65536, 30000]])
def test_distinct_aggregates_inside_group_by_match_oracle(rt, orc, abi, chunks):
    """COUNT / SUM / AVG / TOTAL (DISTINCT x) with GROUP BY: every group runs the reference's distinct accumulator over
    its own rows (llkv-executor/src/lib.rs:5222-5247 → llkv-aggregate/src/lib.rs:95-249: Int64 by value, Float64 by bit
    pattern — ±0 and NaN payloads are different values —, NULL cells skipped).  On the GPU the argument column is the
    least significant sort key of the sort-based route and the reduction takes the first row of every run of equal
    values; plain aggregates ride along in the same query.  Integer results exact, f64 sums within 1e-9 (the oracle adds
    the distinct values in order of first appearance, the GPU in a tree over the sorted rows)."""
    import dataclasses
    rng = np.random.default_rng(23 + len(chunks))
    n = sum(chunks)
    k_int = rng.integers(0, 50, size=n).astype(np.int64) * 1_000_003
    k_tag = [("x", "y", "zz", "")[k] for k in rng.integers(0, 4, size=n)]
    v_int = rng.integers(-40, 40, size=n).astype(np.int64)
    v_f = (rng.integers(-30, 30, size=n) / 4.0).astype(np.float64)
    v_f[rng.random(n) < 0.02] = -0.0
    v_f[rng.random(n) < 0.02] = np.nan
    v_f[rng.random(n) < 0.01] = np.inf
    vi, vf, vk = rng.random(n) > 0.2, rng.random(n) > 0.1, rng.random(n) > 0.1
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT64, k_int), (2, abi.DT_UTF8, k_tag, vk), (3, abi.DT_INT64, v_int, vi), (4, abi.DT_FLOAT64, v_f, vf),
                                       (5, abi.DT_FLOAT64, np.abs(v_f) + 1.0)], chunks)
    A, F, O = abi.AggregateSpec, abi.Filter, abi.Operator
    D = lambda a: dataclasses.replace(a, distinct=True)
    ints = [A.count_star(), D(A.count(3)), D(A.sum(3)), D(A.avg(3)), D(A.total(3)), A.sum(3), A.count(3), D(A.min(3)), A.max(4)]
    flts = [A.count_star(), D(A.count(4)), D(A.total(4)), A.count(4), A.min(5), D(A.max(4))]
    finite = [D(A.sum(5)), D(A.avg(5)), D(A.count(5)), A.sum(5)]
    for keys in ([1], [2], [2, 1]):
        for aggs in (ints, flts, finite):
            for pred in (None, [F(3, O.GreaterThan(-10))]):
                for order in (True, False):
                    got, want = rt.groupby(ht, pred, keys, aggs, order), orc.groupby(ot, pred, keys, aggs, order)
                    assert [[k.value for k in r.keys] for r in got] == [[k.value for k in r.keys] for r in want], (keys, order)
                    for x, y in zip(got, want):
                        assert_values(x.values, y.values, f"distinct/groupby {keys}")
    # a computed argument (r04): the group's temp column holds the PlanValue of every row (llkv-executor/src/lib.rs:5186-5199) —
    # Int ∘ Int through f64, a NULL operand or x % 0 makes the cell NULL — and the distinct accumulator runs over it; on the GPU
    # a projection plan evaluates it once for the selected rows and the values sort as a column's cells would
    col = abi.col
    computed = [[A.count_star(), D(A.count(col(3) * 2)), D(A.sum(col(3) * 2)), D(A.avg(col(3) * 2)), D(A.total(col(3) * 2)), A.sum(3)],
                [D(A.count(col(3) % 7)), D(A.sum(col(3) % 7)), A.count(3)],
                [D(A.count(col(4) * 2.0)), D(A.total(col(4) * 2.0)), A.count(4)],
                [D(A.sum(col(5) + col(3))), D(A.avg(col(5) + col(3))), D(A.count(col(5) + col(3)))],
                [D(A.count(col(3) / col(3))), D(A.sum(col(3) / col(3)))]]  # x / 0 is NULL
    for keys in ([1], [2, 1]):
        for aggs in computed:
            for pred in (None, [F(3, O.GreaterThan(-10))]):
                for order in (True, False):
                    got, want = rt.groupby(ht, pred, keys, aggs, order), orc.groupby(ot, pred, keys, aggs, order)
                    assert [[k.value for k in r.keys] for r in got] == [[k.value for k in r.keys] for r in want], (keys, order)
                    for x, y in zip(got, want):
                        assert_values(x.values, y.values, f"distinct over a computed argument / groupby {keys}")
    for bad in ([D(A.count(3)), D(A.count(4))], [D(A.sum(col(3) * 2)), D(A.count(col(3) * 3))], [D(A.count(3)), D(A.count(col(3) * 2))]):  # two distinct arguments
        with pytest.raises(abi.LlkvError) as e:
            rt.groupby(ht, None, [1], bad, True)
        assert e.value.kind == "Unsupported"


@pytest.mark.parametrize("chunks", [[700], [4096, 4097, 5]])
def test_distinct_aggregates_inside_group_by_over_string_boolean_date_and_decimal_keys(rt, orc, abi, chunks):
    """COUNT / SUM / TOTAL / AVG (DISTINCT x) per group over a Utf8, Boolean, Date32 or Decimal128 column: the key is the cell (a string's
    dictionary code, DistinctKey::from_array llkv-aggregate/src/lib.rs:261-331), what the Float64 accumulators add is its numeric
    image (array_value_to_numeric :400-449).  On the sort-based route the cell is the least significant sort key and the head of
    every run of equal cells adds `dict_num[code]` / 1.0 or 0.0 / the day number — a decimal its raw value, summed in i128 and finalized
    with the column's (precision, scale), AVG half away from zero, SUM / AVG NULL for a group without a value.  Counts and decimals
    exact, f64 sums within 1e-9 (the oracle adds the images in order of first appearance)."""
    rng = np.random.default_rng(57 + len(chunks))
    n = sum(chunks)
    words = ["12", " 3.5 ", "abc", "", "1e2", "12.0", "-7.25", ".5", "1.", "+4", "0.125", "1000000.5", "1", "1.0"]
    txt = [words[i] for i in rng.integers(0, len(words), size=n)]
    boo = rng.integers(0, 2, size=n).astype(np.uint8)
    day = rng.integers(-20, 20, size=n).astype(np.int32) * 365
    g_int = rng.integers(0, 23, size=n).astype(np.int64)
    g_tag = [("a", "b", "", "dd")[k] for k in rng.integers(0, 4, size=n)]
    val = rng.integers(-9, 9, size=n).astype(np.int64)
    dec = [int(v) for v in rng.integers(-40, 40, size=n)]
    vt, vb, vd, vg, vc = rng.random(n) > 0.15, rng.random(n) > 0.1, rng.random(n) > 0.1, rng.random(n) > 0.1, rng.random(n) > 0.5
    ht = rt.HipTable(1, chunks)
    ht.append_utf8_column(1, txt, valid=vt)
    ht.append_column(2, abi.DT_BOOLEAN, boo, valid=vb)
    ht.append_column(3, abi.DT_DATE32, day, valid=vd)
    ht.append_column(4, abi.DT_INT64, g_int)
    ht.append_utf8_column(5, g_tag, valid=vg)
    ht.append_column(6, abi.DT_INT64, val)
    ht.append_decimal128_column(7, 12, 2, dec, valid=vc)
    ot = orc.OracleTable(n)
    ot.add(1, abi.DT_UTF8, [t if ok else None for t, ok in zip(txt, vt)]).add(2, abi.DT_BOOLEAN, boo, list(vb)).add(3, abi.DT_DATE32, day, list(vd))
    ot.add(4, abi.DT_INT64, g_int).add(5, abi.DT_UTF8, [t if ok else None for t, ok in zip(g_tag, vg)]).add(6, abi.DT_INT64, val)
    ot.add(7, abi.DT_DECIMAL128, dec, list(vc), precision=12, scale=2)
    A, F, O = abi.AggregateSpec, abi.Filter, abi.Operator

    def D(kind, e):
        s = getattr(A, kind)(e)
        s.distinct = True
        return s

    for field in (1, 2, 3, 7):
        aggs = [A.count_star()] + [D(k, field) for k in ("count", "sum", "total", "avg")] + [A.sum(6), A.count(field)]
        for keys in ([4], [5], [5, 4]):
            for pred in (None, [F(6, O.GreaterThan(-3))]):
                for order in (True, False):
                    got, want = rt.groupby(ht, pred, keys, aggs, order), orc.groupby(ot, pred, keys, aggs, order)
                    assert [[k.value for k in r.keys] for r in got] == [[k.value for k in r.keys] for r in want], (field, keys, order)
                    for x, y in zip(got, want):
                        assert_values(x.values, y.values, f"distinct/groupby over field {field} by {keys}")


def test_sorted_input_shortcut_of_the_sort_based_group_by(rt, abi, monkeypatch):
    """GROUP BY a column that is in key order already (a clustered primary key) skips the sort; forcing the sort
    (LLKV_HIP_GROUP_ALWAYS_SORT) gives the same groups and the same bits — the stable sort leaves such rows where
    they are."""
    rng = np.random.default_rng(31)
    n = 200_000
    key = np.sort(rng.integers(0, 60_000, size=n)).astype(np.int64)
    val = rng.normal(size=n)
    qty = rng.integers(1, 50, size=n).astype(np.int64)
    t = rt.HipTable(1, [70_000, 130_000])
    t.append_column(1, abi.DT_INT64, key); t.append_column(2, abi.DT_FLOAT64, val); t.append_column(3, abi.DT_INT64, qty)
    A = abi.AggregateSpec
    flat = lambda rows: [(r.keys[0].value, r.values[0].value, r.values[1].value, np.float64(r.values[2].value).tobytes()) for r in rows]
    run = lambda: flat(rt.groupby(t, [abi.Filter(3, abi.Operator.GreaterThan(5))], [1], [A.count_star(), A.sum(3), A.sum(2)], True))
    fast = run()
    sel = qty > 5
    uniq, counts = np.unique(key[sel], return_counts=True)
    assert [r[0] for r in fast] == uniq.tolist() and [r[1] for r in fast] == counts.tolist()
    monkeypatch.setenv("LLKV_HIP_GROUP_ALWAYS_SORT", "1")
    assert run() == fast


def test_c_program_runs_q6_through_the_abi(tmp_path):
    """examples/q6.c: a C99 program (no Python, no C++) stages lineitem, prepares and runs TPC-H Q6 through
    include/llkv_hip.h and checks the answer against its own host loop."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "rust-llkv_amd")
    exe = tmp_path / "q6"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O2", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "q6.c"), "-L", libdir, "-lllkv_hip", "-lllkv_tpch", "-Wl,-rpath," + libdir, "-o", str(exe)])
    out = subprocess.run([str(exe), "2000000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.returncode, out.stdout, out.stderr)


def test_c_example_joins_through_the_abi(tmp_path):
    """examples/q3_join.c: a C99 program (no Python, no C++) stages lineitem / orders / customer, takes lineitem ⋈ orders as the
    reference's joined RecordBatches (llkv_hip_join_stream_batches: names, cells and totals checked against a host join) and
    runs the Q3 pipeline (llkv_hip_join_groupby_topk) against its own host loop, revenue bit for bit."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "rust-llkv_amd")
    exe = tmp_path / "q3_join"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O2", "-I", os.path.join(root, "include"),
                           os.path.join(root, "examples", "q3_join.c"), "-L", libdir, "-lllkv_hip", "-lllkv_tpch", "-Wl,-rpath," + libdir, "-o", str(exe)])
    out = subprocess.run([str(exe), "900000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), (out.returncode, out.stdout, out.stderr)


@pytest.mark.parametrize("order_by_keys", [True, False])
def test_sort_based_group_by_over_a_sharded_table(rt, abi, order_by_keys, monkeypatch):
    """SURVEY §8e for GROUP BY of any cardinality: every rank reduces its own chunks, the partial groups are merged
    in rank order (llkv_hip_query_partial_groups / merge_groups; 2 / 4 / 8 ranks emulated on one device).  Keys,
    order, counts, integer sums, MIN / MAX and NULL keys equal the single-GPU answer exactly, f64 sums within 1e-9."""
    monkeypatch.setenv("LLKV_HIP_GROUP_NO_IMAGE", "1")  # the partial-groups exchange belongs to the sort-based route
    rng = np.random.default_rng(37)
    chunks = [6000, 9000, 300, 20_000, 4096, 17_000, 123, 8000]
    n = sum(chunks)
    k1 = rng.integers(0, 3000, size=n).astype(np.int64)
    k2 = [("x", "yy", "zzz", "")[i] for i in rng.integers(0, 4, size=n)]
    v = rng.normal(size=n)
    q = rng.integers(-100, 100, size=n).astype(np.int64)
    valid1 = rng.random(n) > 0.05
    words = sorted(set(k2))
    A, col = abi.AggregateSpec, abi.col
    aggs = [A.count_star(), A.sum(3), A.sum(col(4) * 2.0), A.min(3), A.max(4), A.avg(3)]

    def shard(rank, world):
        t = rt.HipTable(1, chunks, rank, world)
        lo = sum(chunks[:t.first_chunk])
        hi = lo + t.local_rows
        t.append_column(1, abi.DT_INT64, k1[lo:hi], valid=valid1[lo:hi])
        t.append_utf8_column(2, k2[lo:hi], words)
        t.append_column(3, abi.DT_INT64, q[lo:hi])
        t.append_column(4, abi.DT_FLOAT64, v[lo:hi])
        if world > 1:
            t.set_column_stats(1, 0, 2999)
            t.set_column_stats(3, -100, 99)
        return t

    def cells(rows):
        return [tuple(k.value for k in r.keys) for r in rows], [[x.value for x in r.values] for r in rows]

    whole = rt.PreparedQuery(shard(0, 1), [abi.Filter(3, abi.Operator.GreaterThan(-90))], aggs, [1, 2], order_by_keys)
    want_keys, want_vals = cells(whole.run())
    assert len(want_keys) > 5000
    for world in (2, 4, 8):
        pqs = [rt.PreparedQuery(shard(r, world), [abi.Filter(3, abi.Operator.GreaterThan(-90))], aggs, [1, 2], order_by_keys) for r in range(world)]
        parts = []
        for pq in pqs:
            pq.launch(0)
            pq.finish_only()
            parts.append(pq.partial_groups())
        assert sum(p[2].shape[0] for p in parts) > len(want_keys)  # groups straddle the shards
        last = pqs[-1]
        last.merge_groups(parts)
        got_keys, got_vals = cells(last.rows())
        assert got_keys == want_keys, world
        for g, w in zip(got_vals, want_vals):
            assert g[0] == w[0] and g[1] == w[1] and g[3] == w[3] and g[4] == w[4], (world, g, w)
            assert abs(g[2] - w[2]) <= REL * max(1.0, abs(w[2])) and abs(g[5] - w[5]) <= 1e-12 * max(1.0, abs(w[5]))


@pytest.mark.parametrize("order_by_keys", [True, False])
def test_partitioned_group_by_over_a_sharded_table(rt, abi, order_by_keys):
    """The partitioned route over a sharded table (2 / 4 ranks emulated on one device): every rank reduces its own
    chunks into order-free lanes, the partial groups are merged in rank order like the sort-based route's — keys,
    order (first appearance across the shards, or key order), counts, sums, MIN / MAX / AVG equal the single-GPU answer."""
    rng = np.random.default_rng(43)
    chunks = [6000, 9000, 300, 20_000, 4096, 17_000, 123, 8000]
    n = sum(chunks)
    k1 = rng.integers(0, 150_000, size=n).astype(np.int64)
    q = rng.integers(-100, 100, size=n).astype(np.int64)
    A = abi.AggregateSpec
    aggs = [A.count_star(), A.sum(3), A.min(3), A.max(3), A.avg(3)]
    pred = [abi.Filter(3, abi.Operator.GreaterThan(-90))]

    def shard(rank, world):
        t = rt.HipTable(1, chunks, rank, world)
        lo = sum(chunks[:t.first_chunk])
        hi = lo + t.local_rows
        t.append_column(1, abi.DT_INT64, k1[lo:hi])
        t.append_column(3, abi.DT_INT64, q[lo:hi])
        if world > 1:
            t.set_column_stats(1, 0, 149_999)
            t.set_column_stats(3, -100, 99)
        return t

    def cells(rows):
        return [tuple(k.value for k in r.keys) for r in rows], [[x.value for x in r.values] for r in rows]

    whole = rt.PreparedQuery(shard(0, 1), pred, aggs, [1], order_by_keys)
    assert whole.route_note.startswith("partitioned"), whole.route_note
    want_keys, want_vals = cells(whole.run())
    assert len(want_keys) > 40_000
    for world in (2, 4):
        pqs = [rt.PreparedQuery(shard(r, world), pred, aggs, [1], order_by_keys) for r in range(world)]
        assert all(pq.route_note.startswith("partitioned") for pq in pqs), pqs[0].route_note
        parts = []
        for pq in pqs:
            pq.launch(0)
            pq.finish_only()
            parts.append(pq.partial_groups())
        assert sum(p[2].shape[0] for p in parts) > len(want_keys)  # groups straddle the shards
        last = pqs[-1]
        last.merge_groups(parts)
        got_keys, got_vals = cells(last.rows())
        assert got_keys == want_keys, world
        assert got_vals == want_vals, world


def test_distinct_aggregates_over_a_sharded_table(rt, orc, abi):
    """COUNT / SUM / AVG / TOTAL (DISTINCT) with the table sharded over 2 / 4 ranks (emulated on one device): each
    rank exports the distinct values of its rows, the merge runs over their union in rank order = order of first
    appearance, so even the f64 sums equal the oracle's bit for bit; the non-DISTINCT aggregates beside them combine
    through the exchange image as usual."""
    import dataclasses
    rng = np.random.default_rng(41)
    chunks = [3000, 5000, 70, 9000, 4096, 2000, 11, 6000]
    n = sum(chunks)
    a = rng.integers(-40, 40, size=n).astype(np.int64)
    b = rng.choice(rng.normal(size=500) * 1e3, size=n)
    D = lambda s: dataclasses.replace(s, distinct=True)
    A = abi.AggregateSpec
    aggs = [D(A.count(1)), D(A.sum(1)), D(A.sum(2)), D(A.avg(1)), D(A.total(2)), A.count_star(), A.sum(1)]
    pred = [abi.Filter(1, abi.Operator.GreaterThan(-35))]
    want = [v.value for v in orc.aggregate(orc.OracleTable(n).add(1, abi.DT_INT64, a).add(2, abi.DT_FLOAT64, b), pred, aggs)]

    def shard(rank, world):
        t = rt.HipTable(1, chunks, rank, world)
        lo = sum(chunks[:t.first_chunk])
        t.append_column(1, abi.DT_INT64, a[lo:lo + t.local_rows])
        t.append_column(2, abi.DT_FLOAT64, b[lo:lo + t.local_rows])
        if world > 1:
            t.set_column_stats(1, -40, 39)
        return t

    for world in (1, 2, 4):
        pqs = [rt.PreparedQuery(shard(r, world), pred, aggs) for r in range(world)]
        for pq in pqs:
            pq.launch(0)
        images = [pq.read_exchange() for pq in pqs]
        total = images[0].copy()
        for im in images[1:]:
            total += im  # the integer-SUM all-reduce, by hand
        last = pqs[-1]
        last.finish_from_host(total)
        if world > 1:
            for agg in range(5):
                last.merge_distinct(agg, [pq.distinct_partial(agg) for pq in pqs])
        got = [v.value for v in last.rows()[0].values]
        assert got[0] == want[0] and got[1] == want[1] and got[5] == want[5] and got[6] == want[6], (world, got, want)
        assert np.float64(got[2]).tobytes() == np.float64(want[2]).tobytes() and np.float64(got[4]).tobytes() == np.float64(want[4]).tobytes(), world
        assert abs(got[3] - want[3]) <= 1e-12 * abs(want[3])


def _device_i64(ptr, n):
    """int64 torch tensor aliasing a raw device pointer (what the RCCL all-reduce is given on the GPU box)."""
    import torch

    class Raw:
        pass
    raw = Raw()
    raw.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<i8", "data": (int(ptr), False), "version": 2}
    return torch.as_tensor(raw, device="cuda")


@pytest.mark.parametrize("clustered", [True, False])
def test_q3_sharded_fact_table_is_rank_count_invariant(rt, abi, tpch, clustered):
    """dimension tables replicated, lineitem sharded by chunk over 2 / 4 / 8 ranks (emulated on
    one device: the collectives are done by hand).  Orders whose lineitems straddle a shard boundary are summed
    from their raw values in global row order, so keys, counts, group totals and the revenue BITS equal the
    single-GPU answer for every rank count — also when the fact table is NOT clustered by the key and nearly
    every group straddles."""
    D = tpch.DATE_1995_03_15
    rows, scale = 60175, 0.01
    li = tpch.gen_lineitem(rows, scale)
    if not clustered:
        perm = np.random.default_rng(3).permutation(rows)
        li = {k: v[perm] for k, v in li.items()}
    n_ord = tpch.orders_for_lineitems(rows)
    od = tpch.gen_orders(n_ord, scale)
    n_cust = tpch.customers_for_scale(scale)
    cu = tpch.gen_customer(n_cust, scale)
    seg = [tpch.SEGMENTS[c] for c in cu["c_mktsegment"]]
    ot_ = rt.HipTable(2, tpch.chunk_rows(n_ord, 65536))
    for c, (fid, dt) in tpch.ORDERS_SCHEMA.items():
        ot_.append_column(fid, dt, od[c])
    ct = rt.HipTable(3, tpch.chunk_rows(n_cust, 65536))
    ct.append_column(tpch.C_CUSTKEY, abi.DT_INT64, cu["c_custkey"])
    ct.append_utf8_column(tpch.C_MKTSEGMENT, seg)
    F, O, col = abi.Filter, abi.Operator, abi.col
    revenue = col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT))
    chunks = tpch.chunk_rows(rows, 1000)  # 61 chunks: ragged shards, many boundaries inside orders

    def fact(rank, world):
        t = rt.HipTable(1, chunks, rank, world)
        lo = sum(chunks[:t.first_chunk])
        for c in ("l_orderkey", "l_shipdate", "l_extendedprice", "l_discount"):
            t.append_column(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], li[c][lo:lo + t.local_rows])
        return t

    args = lambda t: dict(fact=t, fact_filters=[F(tpch.L_SHIPDATE, O.GreaterThan(D))], fact_key=tpch.L_ORDERKEY, dim=ot_,
                          dim_filters=[F(tpch.O_ORDERDATE, O.LessThan(D))], dim_key=tpch.O_ORDERKEY, sum_expr=revenue,
                          payload_fields=[tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], dim_fk=tpch.O_CUSTKEY, dim2=ct,
                          dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))], dim2_key=tpch.C_CUSTKEY)
    whole = fact(0, 1)
    want, want_total = rt.join_groupby_topk(limit=10, **args(whole))
    assert len(want) == 10 and want_total > 100
    bits = lambda rws: [(r[0], np.float64(r[1]).tobytes(), r[2], r[3], r[4]) for r in rws]
    for world in (2, 4, 8):
        shards = [fact(r, world) for r in range(world)]
        with pytest.raises(abi.LlkvError):  # the one-call form refuses a shard: its answer would be partial
            rt.join_groupby_topk(limit=10, **args(shards[0]))
        joins = [rt.JoinAgg(**args(t)) for t in shards]
        bufs = [j.counts_buffer() for j in joins]
        assert len({n for _, n in bufs}) == 1  # same groups on every rank: the qualifying dim rows
        tensors = [_device_i64(p, n) for p, n in bufs]
        total = sum(t.clone() for t in tensors)  # ncclAllReduce(SUM) by hand
        for t in tensors:
            t.copy_(total)
        import torch
        torch.cuda.synchronize()
        strad = [j.straddlers() for j in joins]
        n_strad_pairs = sum(len(g) for g, _ in strad)
        if clustered:
            assert n_strad_pairs <= world * 7  # ≤ one order per shard boundary, ≤ 7 lines each
        else:
            assert n_strad_pairs > 100
        folded = rt.fold_straddlers([g for g, _ in strad], [v for _, v in strad])
        parts = [j.candidates(folded, r, 10) for r, j in enumerate(joins)]
        got = rt.merge_join_rows([row for rows_r, _ in parts for row in rows_r], 2, 10)
        assert bits(got) == bits(want), world
        assert sum(n for _, n in parts) == want_total
        # the range form: each rank selects the orders of its own key range only, nothing is exchanged per group — the
        # ranks publish the first and last run of their pair streams (a few hundred bytes) and fold the shared ones
        del joins
        ranged = [rt.JoinAgg(ranged=True, **args(t)) for t in shards]
        blocks = [j.boundary() for j in ranged]
        assert all(len(b) == (8 + 128) * 8 for b in blocks)
        if not clustered:  # pairs out of key order: every rank refuses alike, the caller takes the general form above
            for r, j in enumerate(ranged):
                with pytest.raises(abi.LlkvError) as e:
                    j.finish_ranged(blocks, r, 10)
                assert e.value.kind == "Unsupported"
            continue
        parts = [j.finish_ranged(blocks, r, 10) for r, j in enumerate(ranged)]
        got = rt.merge_join_rows([row for rows_r, _ in parts for row in rows_r], 2, 10)
        assert bits(got) == bits(want), ("ranged", world)
        assert sum(n for _, n in parts) == want_total
        with pytest.raises(abi.LlkvError):
            ranged[0].counts_buffer()
        # … and the same blocks, bit for bit, from a rank that compacts its pairs first (the runs are read off the stripes by default)
        del ranged
        os.environ["LLKV_HIP_JOIN_RANGE_COMPACT"] = "1"
        try:
            compacting = [rt.JoinAgg(ranged=True, **args(t)) for t in shards]
            assert [bytes(j.boundary()) for j in compacting] == [bytes(b) for b in blocks], world
            parts = [j.finish_ranged(blocks, r, 10) for r, j in enumerate(compacting)]
        finally:
            del os.environ["LLKV_HIP_JOIN_RANGE_COMPACT"]
        assert bits(rt.merge_join_rows([row for rows_r, _ in parts for row in rows_r], 2, 10)) == bits(want), ("ranged, compacted", world)


def test_table_staged_from_arr0_chunk_blobs(rt, orc, abi, tpch):
    """§8f-1: ingest llkv-column-map chunk blobs (`ARR0`) instead of raw buffers; results identical."""
    n = 20_000
    chunks = tpch.chunk_rows(n, 8192)
    d = tpch.gen_lineitem(n, 0.01)
    q = tpch.q1()
    ht = rt.HipTable(1, chunks)
    ot = orc.OracleTable(n)
    for c in q.columns:
        fid, dt = tpch.LINEITEM_SCHEMA[c]
        ot.add(fid, dt, d[c])
        blobs, off = [], 0
        for r in chunks:
            part = d[c][off:off + r]
            blobs.append(rt.arr0_serialize(dt, [chr(int(v)) for v in part] if dt == abi.DT_UTF8 else part))
            off += r
        ht.append_arr0_column(fid, blobs)
    # Boolean (bit-packed) and Decimal128 (precision / scale in the header) chunk blobs
    rng = np.random.default_rng(4)
    flags = rng.random(n) < 0.3
    money = [int(v) for v in rng.integers(-10**11, 10**11, size=n)]
    fb, fd = 90, 91
    ot.add(fb, abi.DT_BOOLEAN, flags.astype(np.uint8))
    ot.add(fd, abi.DT_DECIMAL128, money, precision=15, scale=2)
    bb, bd, off = [], [], 0
    for r in chunks:
        bb.append(rt.arr0_serialize(abi.DT_BOOLEAN, flags[off:off + r]))
        bd.append(rt.arr0_serialize(abi.DT_DECIMAL128, money[off:off + r], precision=15, scale=2))
        off += r
    ht.append_arr0_column(fb, bb)
    ht.append_arr0_column(fd, bd)
    got, want = rt.groupby(ht, q.predicate, q.keys, q.aggs, True), orc.groupby(ot, q.predicate, q.keys, q.aggs, True)
    assert [[k.value for k in r.keys] for r in got] == [[k.value for k in r.keys] for r in want]
    for g, w in zip(got, want):
        assert_values(g.values, w.values, "arr0")
    A = abi.AggregateSpec
    aggs = [A.count_star(), A.sum(fd), A.avg(fd), A.min(fd), A.max(fd), A.sum(fb)]
    assert_values(rt.aggregate(ht, q.predicate, aggs), orc.aggregate(ot, q.predicate, aggs), "arr0 boolean / decimal")
    g, w = rt.scan_stream(ht, [fb, fd], q.predicate), orc.scan_stream(ot, [fb, fd], q.predicate)
    assert [c for b in g for c in b[0]] == [c for b in w for c in b[0]]


def test_mvcc_visibility_fused_into_the_scan(rt, orc, abi):
    """§8f-2: the MVCC row filter as a predicate leaf, ANDed with the user predicate inside the kernels
    (aggregate, selection) — against the oracle's row-by-row restatement of is_visible_for."""
    rng = np.random.default_rng(11)
    n = 50_000
    NONE = 2**64 - 1
    created = rng.choice(np.array([1, 2, 3, 4, 5, 6, 7, 9, NONE], dtype=np.uint64), size=n)
    deleted = rng.choice(np.array([NONE, NONE, NONE, 1, 3, 4, 5, 6, 7, 8], dtype=np.uint64), size=n)
    val = rng.integers(0, 1000, size=n).astype(np.int64)
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_UINT64, created), (2, abi.DT_UINT64, deleted), (3, abi.DT_INT64, val)], [8192] * 6 + [848])
    F, O, E, A = abi.Filter, abi.Operator, abi.Expr, abi.AggregateSpec
    for txn, snap, un in ((7, 5, [4]), (1, 6, [4, 9]), (9, 9, []), (2, 0, [3, 4, 5, 6])):
        vis = F(1, O.MvccVisible(2, txn_id=txn, snapshot_id=snap, uncommitted=un))
        assert np.array_equal(rt.filter_row_ids(ht, [vis]), orc.filter_row_ids(ot, [vis]))
        pred = E.all_of([F(3, O.LessThan(500)), vis])
        assert_values(rt.aggregate(ht, pred, [A.count_star(), A.sum(3), A.max(3)]), orc.aggregate(ot, pred, [A.count_star(), A.sum(3), A.max(3)]), "mvcc")


@pytest.mark.parametrize("n_uncommitted", [5, 20, 32])
def test_mvcc_leaf_with_many_non_committed_transactions(rt, orc, abi, n_uncommitted):
    """A busy engine: the snapshot's non-committed set holds up to 32 transaction ids (one literal slot each in the fused leaf; the
    round-3 limit was 4).  Row versions created / deleted by committed, non-committed and the asking transaction, against the
    oracle's RowVersion::is_visible_for; one id more than 32 is handed back."""
    rng = np.random.default_rng(n_uncommitted)
    n = 40_000
    NONE = np.uint64(2**64 - 1)
    created = rng.integers(1, 80, size=n).astype(np.uint64)
    deleted = np.where(rng.random(n) < 0.5, NONE, rng.integers(1, 80, size=n).astype(np.uint64))
    val = rng.integers(0, 1000, size=n).astype(np.int64)
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_UINT64, created), (2, abi.DT_UINT64, deleted), (3, abi.DT_INT64, val)], [16384, 16384, n - 32768])
    uncommitted = [int(v) for v in rng.choice(np.arange(2, 80), size=n_uncommitted, replace=False)]
    F, O, A = abi.Filter, abi.Operator, abi.AggregateSpec
    for txn, snap in ((uncommitted[0], 50), (1, 70), (79, 79)):
        vis = F(1, O.MvccVisible(2, txn_id=txn, snapshot_id=snap, uncommitted=uncommitted))
        want = orc.filter_row_ids(ot, [vis, F(3, O.LessThan(600))])
        assert np.array_equal(rt.filter_row_ids(ht, [vis, F(3, O.LessThan(600))]), want) and 0 < len(want) < n
        assert_values(rt.aggregate(ht, [vis], [A.count_star(), A.sum(3)]), orc.aggregate(ot, [vis], [A.count_star(), A.sum(3)]))
    with pytest.raises(abi.LlkvError) as e:
        rt.aggregate(ht, [F(1, O.MvccVisible(2, txn_id=5, snapshot_id=9, uncommitted=list(range(100, 133))))], [A.count_star()])
    assert e.value.kind == "Unsupported"


@pytest.mark.parametrize("case", golden("mvcc.json")["cases"], ids=lambda c: c["name"])
def test_mvcc_reference_visibility_sequence(rt, abi, case):
    """llkv-transaction/src/mvcc.rs:528-556 (test_row_visibility_simple) through the fused MVCC leaf: the reference's own
    five is_visible_for assertions, each with the ids, snapshot and Active set the manager holds at that point."""
    t = rt.HipTable(1, [1])
    t.append_column(1, abi.DT_UINT64, np.array([case["created_by"]], dtype=np.uint64))
    t.append_column(2, abi.DT_UINT64, np.array([case["deleted_by"]], dtype=np.uint64))
    F, O = abi.Filter, abi.Operator
    vis = F(1, O.MvccVisible(2, txn_id=case["txn_id"], snapshot_id=case["snapshot_id"], uncommitted=case["uncommitted"]))
    assert (rt.filter_row_ids(t, [vis]).tolist() == [0]) == case["expect"]
    assert rt.aggregate(t, [vis], [abi.AggregateSpec.count_star()])[0].value == int(case["expect"])


@pytest.mark.parametrize("case", golden("mvcc.json")["count_cases"], ids=lambda c: c["name"])
def test_mvcc_count_star_with_transaction_local_changes(rt, abi, case):
    """llkv-slt-tester/tests/slt/duckdb/transactions/count_star_transactions.slt through the fused MVCC leaf: COUNT(*) while another
    connection's deletes / appends are uncommitted, and after they commit — the reference's own answers."""
    created = np.concatenate([np.full(v["rows"], v["created_by"], dtype=np.uint64) for v in case["versions"]])
    deleted = np.concatenate([np.full(v["rows"], v["deleted_by"], dtype=np.uint64) for v in case["versions"]])
    t = rt.HipTable(1, [len(created)])
    t.append_column(1, abi.DT_UINT64, created)
    t.append_column(2, abi.DT_UINT64, deleted)
    vis = abi.Filter(1, abi.Operator.MvccVisible(2, txn_id=case["txn_id"], snapshot_id=case["snapshot_id"], uncommitted=case["uncommitted"]))
    assert rt.aggregate(t, [vis], [abi.AggregateSpec.count_star()])[0].value == case["expect"]
    assert len(rt.filter_row_ids(t, [vis])) == case["expect"]


def test_q1_qualifies_against_an_oracle_answer_set(rt, orc, abi, tpch):
    """§8f-3: the qualification harness (order-insensitive diff, exact ints/strings, ABSOLUTE 1e-9 on floats) run on
    the GPU path's Q1 rows against an answer set rendered from the oracle in dbgen's `|` format.  The absolute
    tolerance is far below one ulp of a 1e8-sized f64 sum, so the data here is dyadic (prices integral, discount and
    tax multiples of 1/2): every sum is exact in f64 and therefore independent of the summation order."""
    qual = __import__("importlib").import_module("rust-llkv_amd.qualify")
    n = 50_000
    d = tpch.gen_lineitem(n, 0.01)
    rng = np.random.default_rng(3)
    d["l_extendedprice"] = rng.integers(1, 2**20, size=n).astype(np.float64)
    d["l_discount"] = rng.integers(0, 2, size=n) * 0.5
    d["l_tax"] = rng.integers(0, 3, size=n) * 0.5
    q1 = tpch.q1()
    ht, ot = stage_both(rt, orc, abi, [(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], d[c]) for c in q1.columns], tpch.chunk_rows(n, 8192))
    want = orc.groupby(ot, q1.predicate, q1.keys, q1.aggs, True)
    lines = ["l_returnflag|l_linestatus|sum_qty|sum_base_price|sum_disc_price|sum_charge|avg_qty|avg_price|avg_disc|count_order"]
    for r in reversed(want):  # shuffled on purpose: the diff is order-insensitive
        lines.append("|".join([k.value for k in r.keys] + [repr(v.value) for v in r.values]))

    def run():
        return [[k.value for k in r.keys] + [v.value for v in r.values] for r in rt.groupby(ht, q1.predicate, q1.keys, q1.aggs, True)]

    diff, elapsed = qual.qualify(run, "\n".join(lines), qual.Q1_TOKENS)
    assert diff.ok, (diff.missing, diff.extra)
    assert elapsed > 0


def test_q1_q6_q3_under_the_qualification_rule_on_real_shaped_data(rt, orc, abi, tpch):
    """§8f-3 with the default substitution parameters (qualify.render_query) on the synthetic TPC-H-shaped data — prices in
    cents, discounts in hundredths: nothing is dyadic, every f64 sum rounds.  What must hold at any size: strings and
    integers exact, every number within the 1e-9 RELATIVE of the contract.  What the reference's harness asks
    (llkv-tpch/src/qualification.rs:708-745) is stricter and size dependent — `avg` columns within an ABSOLUTE 1e-9,
    `sum` columns equal as decimals after Decimal::from_f64 (15 significant digits): at SF0.01 the avg columns pass;
    whether a sum column passes depends on the last bits of two different summation orders (tools/qualify_report.py
    records it at SF1 / SF10; DESIGN.md "Qualification")."""
    qual = __import__("importlib").import_module("rust-llkv_amd.qualify")
    rows, scale = tpch.LINEITEM_ROWS["sf0.01"], 0.01
    d = tpch.gen_lineitem(rows, scale)
    ht, ot = stage_both(rt, orc, abi, [(fid, dt, d[name]) for name, (fid, dt) in tpch.LINEITEM_SCHEMA.items()], tpch.chunk_rows(rows, 8192))
    q1 = qual.render_query(tpch, abi, 1)
    cells = lambda rws: [[k.value for k in r.keys] + [v.value for v in r.values] for r in rws]
    want1, got1 = cells(orc.groupby(ot, q1.predicate, q1.keys, q1.aggs, True)), cells(rt.groupby(ht, q1.predicate, q1.keys, q1.aggs, True))
    rep = qual.compare_report(want1, got1, qual.Q1_TOKENS)
    assert rep["rows"] == 4
    for c in rep["columns"]:
        assert c["max_rel_diff"] <= REL, c
        if c["kind"] in ("string", "integer", "float"):
            assert c["passes_reference_rule"], c  # flags, count_order; avg_qty / avg_price / avg_disc within 1e-9 absolute
    assert rep["columns"][2]["passes_reference_rule"]  # sum_qty: an exact Int64 sum, whatever the order
    # the harness end to end: the answer set as text, the order-insensitive diff over the columns that must agree
    tokens = ["str", "str", "sum", "avg", "avg", "avg", "cnt"]
    pick = lambda r: [r[0], r[1], r[2], r[6], r[7], r[8], r[9]]
    text = qual.format_answer_set(["l_returnflag", "l_linestatus", "sum_qty", "avg_qty", "avg_price", "avg_disc", "count_order"],
                                  [pick(r) for r in reversed(want1)], [qual.kind_from_token(t) for t in tokens])
    diff, elapsed = qual.qualify(lambda: [pick(r) for r in got1], text, tokens)
    assert diff.ok and elapsed >= 0
    q6 = qual.render_query(tpch, abi, 6)
    rep6 = qual.compare_report([[orc.aggregate(ot, q6.predicate, q6.aggs)[0].value]], [[rt.aggregate(ht, q6.predicate, q6.aggs)[0].value]], qual.Q6_TOKENS)
    assert rep6["columns"][0]["max_rel_diff"] <= REL


def test_integer_and_multi_key_group_by(rt, orc, abi, tpch):
    """GroupKeyValue::Int keys (every integer width and Date32 collapse to Int, llkv-executor/src/lib.rs:9362-9456)
    through dense ids from the staging statistics; up to 64 dense groups in the LDS accumulator image."""
    n = 100_000
    d = tpch.gen_lineitem(n, 0.02)
    rng = np.random.default_rng(5)
    day = (9000 + rng.integers(0, 9, size=n)).astype(np.int32)
    cols = [(4, abi.DT_INT64, d["l_linenumber"]), (9, abi.DT_UTF8, d["l_returnflag"]), (5, abi.DT_INT64, d["l_quantity"]),
            (6, abi.DT_FLOAT64, d["l_extendedprice"]), (20, abi.DT_DATE32, day)]
    ht, ot = stage_both(rt, orc, abi, cols, tpch.chunk_rows(n, 8192))
    A, F, O, col = abi.AggregateSpec, abi.Filter, abi.Operator, abi.col
    wide = [A.count_star(), A.sum(5), A.avg(6), A.min(5), A.max(6)]  # 7–8 lanes per group
    narrow = [A.count_star(), A.avg(6)]                              # 2–3 lanes per group
    # the LDS accumulator image holds groups × lanes ≤ 79
    for keys, pred, aggs in (([4], None, wide), ([4, 9], [F(5, O.LessThan(30))], narrow), ([20], None, wide),
                             ([9, 20], [F(6, O.GreaterThan(20000.0))], narrow[:1] + [A.sum(5)])):
        for ordered in (True, False):
            got, want = rt.groupby(ht, pred, keys, aggs, ordered), orc.groupby(ot, pred, keys, aggs, ordered)
            assert [[k.value for k in r.keys] for r in got] == [[k.value for k in r.keys] for r in want], (keys, ordered)
            for g, w in zip(got, want):
                assert_values(g.values, w.values, str(keys))
    for keys, aggs in (([5, 4, 20], [A.count_star()]), ([4, 9], wide)):  # 50·7·9 groups / 21 groups × 7 lanes: the sort-based route
        for ordered in (True, False):
            got, want = rt.groupby(ht, None, keys, aggs, ordered), orc.groupby(ot, None, keys, aggs, ordered)
            assert [[k.value for k in r.keys] for r in got] == [[k.value for k in r.keys] for r in want], (keys, ordered)
            for g, w in zip(got, want):
                assert_values(g.values, w.values, str(keys))


@pytest.mark.parametrize("case", golden("string_scans.json")["cases"], ids=lambda c: c["name"])
def test_reference_string_scan_known_answers(rt, abi, case):
    """llkv-table/tests/fusion_tests.rs:110-318 and table.rs:1725-1771 through llkv_hip_scan_stream: string predicates
    (and their same-field AND) run on dictionary codes.  A column with more than 256 distinct strings is not staged."""
    from conftest import build_string_operator
    from test_oracle_golden import _string_scan_values
    values = _string_scan_values(case)
    ht = rt.HipTable(1, [len(values)])
    if len(set(values)) > 256:
        with pytest.raises(abi.LlkvError) as e:
            ht.append_utf8_column(1, values)
        assert e.value.kind == "Unsupported"
        return
    ht.append_utf8_column(1, values)
    filters = [abi.Filter(1, build_string_operator(abi, op)) for op in case["ops"]]
    got = [v for cols, _ in rt.scan_stream(ht, [1], filters) for v in cols[0]]
    if "expect_in_order" in case:
        assert got == case["expect_in_order"]
    if "expect_sorted" in case:
        assert sorted(got) == case["expect_sorted"]


def _staged(rt, abi, columns):
    n = len(columns[0]["values"])
    ht = rt.HipTable(1, [n])
    for c in columns:
        dt = DTYPES[c["dtype"]]
        if dt == abi.DT_UTF8:
            ht.append_utf8_column(c["field_id"], c["values"])
        else:
            ht.append_column(c["field_id"], dt, np.array(c["values"], dtype=abi.NUMPY_OF_DTYPE[dt]))
    return ht


def test_reference_join_with_expression_filters(rt, abi):
    """llkv-join/tests/join_tests.rs:299-561 through llkv_hip_scan_stream (Expr::Compare filters) and
    llkv_hip_join_stream (inner join on Int32 keys): the filtered id sets, 8 joined rows, and their split over the
    two filters are the reference's assertions."""
    from conftest import build_predicate
    c = golden("join_filters.json")["expression_filters"]
    left, right = _staged(rt, abi, c["left"]["columns"]), _staged(rt, abi, c["right"]["columns"])
    lids = [v for cols, _ in rt.scan_stream(left, [c["left_filter_project"]], build_predicate(abi, c["left_filter"])) for v in cols[0]]
    rids = [v for cols, _ in rt.scan_stream(right, [c["right_filter_project"]], build_predicate(abi, c["right_filter"])) for v in cols[0]]
    assert sorted(set(lids)) == c["expect_left_ids"]
    assert len(set(rids)) == c["expect_right_id_count"] and set(c["expect_right_ids_include"]) <= set(rids)
    pairs = [(l, r) for b in rt.join_stream(left, right, [tuple(k) for k in c["join_keys"]], abi.JOIN_INNER) for l, r in zip(b[0], b[1])]
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


@pytest.mark.parametrize("case", golden("join_filters.json")["cartesian"], ids=lambda c: c["name"])
def test_reference_cartesian_known_answers(rt, abi, case):
    """join_tests.rs:711-806 and llkv-executor/src/lib.rs:14065-14150 through llkv_hip_join_stream without keys."""
    if "tables" in case:
        cols = [[v] for v in case["tables"][0]]
        for nxt in case["tables"][1:]:
            lt = rt.HipTable(1, [len(cols)]); lt.append_column(1, abi.DT_INT64, np.arange(len(cols)))
            rt_ = rt.HipTable(2, [len(nxt)]); rt_.append_column(1, abi.DT_INT64, np.array(nxt, dtype=np.int64))
            pairs = [(l, r) for b in rt.join_stream(lt, rt_, [], abi.JOIN_INNER) for l, r in zip(b[0], b[1])]
            cols = [cols[l] + [nxt[r]] for l, r in pairs]
        assert len(cols) == case["expect_rows"]
        assert [[row[i] for row in cols] for i in range(len(case["tables"]))] == case["expect_columns"]
        return
    left, right = _join_side(rt, abi, case["left"]), _join_side(rt, abi, case["right"])
    pairs = [(l, r) for b in rt.join_stream(left, right, [], abi.JOIN_INNER) for l, r in zip(b[0], b[1])]
    assert len(pairs) == case["expect_rows"]
    if "expect_combinations" in case:
        got = {(case["left"][l][0], case["left"][l][1], case["right"][r][0], case["right"][r][1]) for l, r in pairs}
        assert got == {tuple(x) for x in case["expect_combinations"]}


@pytest.mark.parametrize("dt", ["Int64", "Int32", "UInt32", "UInt64"])
def test_sorted_range_scans_of_shuffled_integers(rt, abi, dt):
    """The property llkv-column-map/tests/integer_scan_tests.rs:226-… asserts for every integer type — a sorted scan of
    200 000 shuffled values, with and without a range, yields exactly the sorted (filtered) values with their row
    ids — through llkv_hip_scan_stream with ScanStreamOptions.order (the reference's test draws random data; so does
    this one, from a fixed seed)."""
    n = 200_000
    rng = np.random.default_rng(226)
    code = DTYPES[dt]
    npdt = np.dtype(abi.NUMPY_OF_DTYPE[code])
    info = np.iinfo(npdt)
    lo, hi = max(info.min, -10**12), min(info.max, 10**12)
    vals = rng.integers(lo, hi, size=n, dtype=np.int64 if npdt.kind == "i" else np.uint64).astype(npdt)
    ht = rt.HipTable(1, [70_000, 65_536, 64_464])
    ht.append_column(1, code, vals)
    transform = abi.ORDER_IDENTITY_INT64 if dt == "Int64" else abi.ORDER_IDENTITY_INT32 if dt == "Int32" else None
    if transform is None:
        pytest.skip("ScanOrderTransform has identity transforms for Int64 / Int32 / Utf8 only (llkv-scan/src/lib.rs:41-46)")
    order = (1, False, False, transform)
    got_v, got_r = [], []
    for cols, rids in rt.scan_stream(ht, [1], None, include_row_ids=True, order=order):
        got_v.extend(cols[0]); got_r.extend(rids)
    idx = np.argsort(vals, kind="stable")
    assert got_v == vals[idx].tolist() and got_r == idx.tolist()
    a, b = int(np.percentile(vals.astype(np.float64), 25)), int(np.percentile(vals.astype(np.float64), 75))
    F, O, B = abi.Filter, abi.Operator, abi.Bound
    got_v, got_r = [], []
    for cols, rids in rt.scan_stream(ht, [1], [F(1, O.Range(B.Included(a), B.Included(b)))], include_row_ids=True, order=order):
        got_v.extend(cols[0]); got_r.extend(rids)
    keep = idx[(vals[idx] >= a) & (vals[idx] <= b)]
    assert got_v == vals[keep].tolist() and got_r == keep.tolist()


def test_randomized_parity_window_of_the_fuzz_tool():
    """tools/fuzz_parity.py (random predicate trees, aggregate lists, computed projections bit for bit, join key lists,
    DISTINCT aggregates, ordered scans, partitioned GROUP BY shapes, decimal aggregate arguments; GPU vs oracle) over a small window of seeds the other
    tests do not use — the long runs of DESIGN §2 are the same script with a wider window (LLKV_FUZZ_SEEDS=lo:hi)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LLKV_FUZZ_SEEDS="700:702", LLKV_FUZZ_PROJECTIONS="3", LLKV_FUZZ_JOINS="20", LLKV_FUZZ_DISTINCT="8", LLKV_FUZZ_ORDERED="4",
               LLKV_FUZZ_PARTITIONED="1", LLKV_FUZZ_DECIMALS="6")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py")], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    verdicts = [line for line in out.stdout.splitlines() if "FAILURES:" in line]
    assert len(verdicts) == 7, verdicts  # (… and the decimal-argument family, r04)
    assert verdicts[0].strip() == "FAILURES: []", verdicts
    for line in verdicts[1:]:
        assert line.strip().endswith("FAILURES: 0"), verdicts


def test_plan_kernels_are_taken_from_the_seed_directory_when_it_holds_them(tmp_path):
    """`jit_seed/` beside libllkv_hip.so holds code objects of earlier hiprtc compilations under the key of the user's
    cache (source + compiler identity).  A plan compiled once (process 1: seeds off, its code object lands in cache
    directory A) is loaded from the seed directory by a process with an empty cache (process 2) — nothing compiled, the
    same answer."""
    import glob
    import json
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seeds = os.path.join(root, "rust-llkv_amd", "jit_seed")
    script = tmp_path / "seed_probe.py"
    script.write_text(
        "import importlib, sys, json, ctypes as C, numpy as np\n"
        "sys.path.insert(0, sys.argv[1])\n"
        "abi = importlib.import_module('rust-llkv_amd.abi'); rt = importlib.import_module('rust-llkv_amd.runtime')\n"
        "rt.init(0)\n"
        "t = rt.HipTable(1, [1000])\n"
        "t.append_column(1, abi.DT_INT64, np.arange(1000, dtype=np.int64))\n"
        "t.append_column(2, abi.DT_FLOAT64, np.arange(1000, dtype=np.float64) / 8)\n"
        "A = abi.AggregateSpec\n"
        "vals = [v.value for v in rt.aggregate(t, [abi.Filter(1, abi.Operator.GreaterThan(17))], [A.sum(abi.col(2) * 3.25 + abi.col(1) * 7), A.max(abi.col(1) % 13), A.count_star()])]\n"
        "x = [C.c_uint64(), C.c_uint64(), C.c_uint64()]\n"
        "rt.lib().llkv_hip_jit_stats(*[C.byref(v) for v in x])\n"
        "print(json.dumps({'values': vals, 'compiled': x[0].value, 'from_cache': x[1].value, 'from_seed': x[2].value}))\n")
    a, b = tmp_path / "cache_a", tmp_path / "cache_b"
    a.mkdir(mode=0o700)
    b.mkdir(mode=0o700)

    def run(cache, no_seed):
        env = dict(os.environ, LLKV_HIP_CACHE_DIR=str(cache))
        env.pop("LLKV_HIP_NO_JIT_SEED", None)
        if no_seed:
            env["LLKV_HIP_NO_JIT_SEED"] = "1"
        out = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])

    first = run(a, True)
    assert first["compiled"] >= 1 and first["from_seed"] == 0, first
    made = sorted(glob.glob(str(a / "*.hsaco")))
    assert len(made) == first["compiled"], (made, first)
    os.makedirs(seeds, exist_ok=True)
    planted = []
    try:
        for f in made:
            dst = os.path.join(seeds, os.path.basename(f))
            if not os.path.exists(dst):
                shutil.copy(f, dst)
                planted.append(dst)
        second = run(b, False)
        assert second["compiled"] == 0 and second["from_seed"] == first["compiled"], (first, second)
        assert second["values"] == first["values"]
    finally:
        for f in planted:
            os.remove(f)
