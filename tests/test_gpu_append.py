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


def test_key_images_are_dropped_by_an_append_and_built_again(rt, abi, tpch, monkeypatch):
    """The join pipeline's 4-byte key images (KeyImage, csrc/engine.hpp) belong to a generation of the table: an append drops them
    (the column may have moved, its statistics have changed — a new key may no longer fit 32 bits) and the next join builds them over
    the grown image.  Q3's star over tables staged without their last chunks, then grown: the top ten before and after equal what
    numpy computes from the same rows; an order key beyond 32 bits arriving with an append turns the image off for that column."""
    monkeypatch.setenv("LLKV_HIP_KEY_IMAGE_MIN_ROWS", "1")
    rows, scale = 120_000, 0.02
    li = tpch.gen_lineitem(rows, scale, ["l_orderkey", "l_shipdate", "l_extendedprice", "l_discount"])
    n_ord = tpch.orders_for_lineitems(rows)
    od = tpch.gen_orders(n_ord, scale)
    D = tpch.DATE_1995_03_15
    F, O, col = abi.Filter, abi.Operator, abi.col
    rev = col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT))

    def expected(n_li, n_or, okeys=None):
        okey = od["o_orderkey"][:n_or] if okeys is None else okeys
        ok = od["o_orderdate"][:n_or] < D
        sums, counts = {}, {}
        keep = dict(zip(okey[ok].tolist(), np.flatnonzero(ok).tolist()))
        lk, ls = li["l_orderkey"][:n_li], li["l_shipdate"][:n_li]
        val = li["l_extendedprice"][:n_li] * (1 - li["l_discount"][:n_li])
        for i in np.flatnonzero(ls > D).tolist():
            k = int(lk[i])
            if k in keep:
                sums[k] = sums.get(k, 0.0) + float(val[i])  # row order: the pipeline adds an order's rows left to right
                counts[k] = counts.get(k, 0) + 1
        order = sorted(sums, key=lambda k: (-sums[k], int(od["o_orderdate"][keep[k]]), keep[k]))[:10]
        return [(k, sums[k], counts[k], int(od["o_orderdate"][keep[k]])) for k in order], len(sums)

    def run(lt, ot_):
        got, total = rt.join_groupby_topk(lt, [F(tpch.L_SHIPDATE, O.GreaterThan(D))], tpch.L_ORDERKEY, ot_, [F(tpch.O_ORDERDATE, O.LessThan(D))], tpch.O_ORDERKEY, rev,
                                          payload_fields=[tpch.O_ORDERDATE], limit=10)
        return [(g[0], g[1], g[2], g[3]) for g in got], total

    li_chunks, or_chunks = tpch.chunk_rows(rows, 16384), tpch.chunk_rows(n_ord, 8192)
    li_head, or_head = sum(li_chunks[:-2]), sum(or_chunks[:-1])
    lt = rt.HipTable(1, li_chunks[:-2])
    for c in li:
        lt.append_column(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], li[c][:li_head])
    ot_ = rt.HipTable(2, or_chunks[:-1])
    for c, (fid, dt) in tpch.ORDERS_SCHEMA.items():
        ot_.append_column(fid, dt, od[c][:or_head])
    assert run(lt, ot_) == expected(li_head, or_head)
    assert lt.key_images()[0] == 1 and ot_.key_images()[0] == 1
    lt.append_chunks(li_chunks[-2:], {tpch.LINEITEM_SCHEMA[c][0]: li[c][li_head:] for c in li})
    ot_.append_chunks(or_chunks[-1:], {fid: od[c][or_head:] for c, (fid, dt) in tpch.ORDERS_SCHEMA.items()})
    assert lt.key_images() == (0, 0) and ot_.key_images() == (0, 0)  # a new generation: the images went with the old one
    assert run(lt, ot_) == expected(rows, n_ord)
    assert lt.key_images()[0] == 1 and ot_.key_images()[0] == 1
    # an order whose key does not fit 32 bits arrives: the orders' key column is read as it is from now on (the lineitem image stays)
    big = {fid: od[c][-3:].copy() for c, (fid, dt) in tpch.ORDERS_SCHEMA.items()}
    big[tpch.O_ORDERKEY] = np.array([2**33 + 1, 2**33 + 5, 2**33 + 9], dtype=np.int64)
    ot_.append_chunks([3], big)
    keys = np.concatenate([od["o_orderkey"], big[tpch.O_ORDERKEY]])
    got = run(lt, ot_)
    assert ot_.key_images()[0] == 0 and lt.key_images()[0] == 1
    want = expected(rows, n_ord)  # (no lineitem names the three new orders)
    assert got == want and len(keys) == n_ord + 3
