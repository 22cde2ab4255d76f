"""GPU (-m gpu): incremental growth of a staged table (llkv_hip_table_append_chunks — ColumnStore::append
llkv-column-map/src/store/core.rs:787): the grown image answers like a table staged whole, the oracle's answers over the grown
table, and only the new chunks cross the host → HBM link."""
import numpy as np
import pytest

from test_gpu_parity import assert_values

pytestmark = pytest.mark.gpu


def stage(rt, abi, tpch, d, rows, chunk):
    t = rt.HipTable(1, tpch.chunk_rows(rows, chunk))
    for c, (fid, dt) in tpch.LINEITEM_SCHEMA.items():
        if c not in d:
            continue
        t.append_utf8_column(fid, d[c][:rows]) if dt == abi.DT_UTF8 else t.append_column(fid, dt, d[c][:rows])
    return t


def test_three_appended_chunks_answer_like_the_whole_table(rt, orc, abi, tpch):
    """SF0.01 lineitem staged without its last three chunks, which are then appended one call at a time: C1 / Q6 / Q1 and
    scan_stream equal the oracle over the grown table; the staging bytes of an append are those of its chunks alone; a query
    prepared before an append refuses to launch until it is prepared again."""
    n, chunk = tpch.LINEITEM_ROWS["sf0.01"], 8192
    cols = ["l_quantity", "l_extendedprice", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate"]
    d = tpch.gen_lineitem(n, 0.01, cols)
    chunks = tpch.chunk_rows(n, chunk)
    head = sum(chunks[:-3])
    t = stage(rt, abi, tpch, d, head, chunk)
    ot = orc.OracleTable(n)
    for c in cols:
        ot.add(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], d[c])
    q1 = tpch.q1()
    stale = rt.PreparedQuery(t, q1.predicate, q1.aggs, q1.keys, True)
    stale.run()
    at = head
    row_bytes = sum(8 if tpch.LINEITEM_SCHEMA[c][1] in (abi.DT_INT64, abi.DT_FLOAT64) else 4 if tpch.LINEITEM_SCHEMA[c][1] == abi.DT_DATE32 else 1 for c in cols)
    for r in chunks[-3:]:
        b0, _ = rt.staging_stats()
        t.append_chunks([r], {tpch.LINEITEM_SCHEMA[c][0]: d[c][at:at + r] for c in cols})
        b1, _ = rt.staging_stats()
        assert r * row_bytes <= b1 - b0 <= r * row_bytes + 16 * len(cols) + 4096, (r, b1 - b0)  # (+ padding to 16 rows; the statistics' few bytes come back, they do not go out)
        at += r
    assert t.total_rows == n and t.generation == 3
    with pytest.raises(abi.LlkvError) as e:
        stale.launch()
    assert e.value.kind == "InvalidArgumentError" and "prepare it again" in e.value.message
    stale.close()
    for name in ("c1", "q6"):
        q = tpch.QUERIES[name]()
        assert_values(rt.aggregate(t, q.predicate, q.aggs), orc.aggregate(ot, q.predicate, q.aggs), name)
    for ordered in (True, False):
        got, want = rt.groupby(t, q1.predicate, q1.keys, q1.aggs, ordered), orc.groupby(ot, q1.predicate, q1.keys, q1.aggs, ordered)
        assert [[k.value for k in r.keys] for r in got] == [[k.value for k in r.keys] for r in want]
        for g, w in zip(got, want):
            assert_values(g.values, w.values, "q1 over the grown table")
    F, O, col = abi.Filter, abi.Operator, abi.col
    pred = [F(tpch.L_QUANTITY, O.LessThan(3))]
    projs = [tpch.L_EXTENDEDPRICE, tpch.L_RETURNFLAG, col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT))]
    assert rt.scan_stream(t, projs, pred, include_row_ids=True) == orc.scan_stream(ot, projs, pred, include_row_ids=True)
    assert np.array_equal(rt.filter_row_ids(t, pred), orc.filter_row_ids(ot, pred))
    # … and bit for bit what a table staged whole gives (same tiles over the same chunk list: same reduction association)
    whole = stage(rt, abi, tpch, d, n, chunk)
    a, b = rt.groupby(t, q1.predicate, q1.keys, q1.aggs, True), rt.groupby(whole, q1.predicate, q1.keys, q1.aggs, True)
    assert [(r.keys, r.values) for r in a] == [(r.keys, r.values) for r in b]


def test_append_grows_dictionaries_validity_decimals_and_row_ids(rt, orc, abi):
    """What else an append carries: new strings join the dictionary (old codes stay), a column without NULL cells gets its mask
    when the new chunks bring the first NULL, Decimal128 cells are narrowed, row ids with gaps continue — GROUP BY, filters, scans
    and reported ids against the oracle over the grown table; what the data can refuse leaves the table untouched."""
    rng = np.random.default_rng(9)
    n0, n1, n2 = 5000, 3000, 4097
    n = n0 + n1 + n2
    k = rng.integers(0, 7, size=n).astype(np.int64)
    v = rng.integers(-1000, 1000, size=n).astype(np.int64)
    f = rng.normal(size=n) * 100
    words = np.array(["a", "bb", "ccc"])[rng.integers(0, 3, size=n)].tolist()
    for i in range(n0, n):
        if rng.random() < 0.3:
            words[i] = ["dddd", "e"][i % 2]      # strings the staged dictionary has never seen
    vv = np.ones(n, dtype=bool)
    vv[n0 + n1:] = rng.random(n2) > 0.2          # the first NULL cells arrive with the second append
    money = rng.integers(-10**12, 10**12, size=n).astype(np.int64)
    ids = np.cumsum(rng.integers(1, 4, size=n)).astype(np.uint64) + np.uint64(1000)
    t = rt.HipTable(1, [n0])
    t.append_column(1, abi.DT_INT64, k[:n0])
    t.append_column(2, abi.DT_INT64, v[:n0])
    t.append_column(3, abi.DT_FLOAT64, f[:n0])
    t.append_utf8_column(4, words[:n0])
    t.append_decimal128_column(5, 15, 2, money[:n0])
    t.set_row_ids(ids[:n0])
    cols = lambda lo, hi: {1: k[lo:hi], 2: v[lo:hi], 3: f[lo:hi], 4: words[lo:hi], 5: money[lo:hi]}
    t.append_chunks([n1], cols(n0, n0 + n1), row_ids=ids[n0:n0 + n1])
    # refused appends change nothing: a missing column, ids that do not ascend beyond the table's last one
    with pytest.raises(abi.LlkvError):
        t.append_chunks([n2], {1: k[n0 + n1:], 2: v[n0 + n1:]}, row_ids=ids[n0 + n1:])
    with pytest.raises(abi.LlkvError) as e:
        t.append_chunks([n2], cols(n0 + n1, n), row_ids=ids[:n2])
    assert "ascend" in e.value.message and t.total_rows == n0 + n1 and t.generation == 1
    t.append_chunks([4096, 1], cols(n0 + n1, n), valid={2: vv[n0 + n1:]}, row_ids=ids[n0 + n1:])
    assert t.total_rows == n and t.generation == 2
    ot = orc.OracleTable(n)
    ot.add(1, abi.DT_INT64, k).add(2, abi.DT_INT64, v, list(vv)).add(3, abi.DT_FLOAT64, f).add(4, abi.DT_UTF8, words).add(5, abi.DT_DECIMAL128, money, precision=15, scale=2)
    A, F, O = abi.AggregateSpec, abi.Filter, abi.Operator
    aggs = [A.count_star(), A.sum(2), A.count(2), A.min(2), A.sum(3), A.sum(5), A.avg(5), A.max(5)]
    for keys in ([1], [4], [4, 1]):
        for ordered in (True, False):
            got, want = rt.groupby(t, None, keys, aggs, ordered), orc.groupby(ot, None, keys, aggs, ordered)
            assert [[x.value for x in r.keys] for r in got] == [[x.value for x in r.keys] for r in want], (keys, ordered)
            for g, w in zip(got, want):
                assert_values(g.values, w.values, f"group by {keys} after appends")
    for pred in ([F(4, O.Equals("dddd"))], [F(2, O.GreaterThan(0))], [F(4, O.In(["e", "a"])), F(1, O.LessThan(4))]):
        want_pos = orc.filter_row_ids(ot, pred)
        assert np.array_equal(rt.filter_row_ids(t, pred), ids[want_pos.astype(np.int64)])
        assert_values(rt.aggregate(t, pred, aggs), orc.aggregate(ot, pred, aggs))
    got = rt.scan_stream(t, [4, 2, 5], [F(1, O.Equals(3))], include_nulls=True)
    want = orc.scan_stream(ot, [4, 2, 5], [F(1, O.Equals(3))], include_nulls=True)
    assert got == want
