"""GPU: the shared-image GROUP BY kernel (image_scan_body, fused_scan.hip.h) — hundreds to thousands of groups, one
accumulator image per workgroup in LDS, every lane order-free (integer adds, min / max, f64 sums as exact two-level
pairs).  Reference semantics: execute_group_by_with_aggregates llkv-executor/src/lib.rs:5028-5355 through the oracle.
  * parity with the oracle: groups, first-appearance / key order, NULL keys, counts and integer results exact,
    f64 sums within 1e-9 (in fact far closer: the sums are exact up to the final rounding);
  * exactness makes the bits independent of EVERYTHING but the data: workgroup count, tile list, shards;
  * what has no bound from the statistics (a division inside a sum) keeps the sort-based route."""
import os

import numpy as np
import pytest

from conftest import mod
from test_gpu_parity import REL, assert_values, stage_both

pytestmark = pytest.mark.gpu


def _flat(rows):
    return [(tuple(k.value for k in r.keys), tuple(np.float64(v.value).tobytes() if isinstance(v.value, float) else v.value for v in r.values)) for r in rows]


def _image(pq):
    import re
    return re.search(r",2(,\d+(,-1,1)?)?>$", pq.kernel_signature) is not None


@pytest.mark.parametrize("chunks", [[7], [4096, 4097, 5], [65536, 70000, 30011]])
def test_shared_image_group_by_matches_oracle(rt, orc, abi, chunks):
    rng = np.random.default_rng(11 + len(chunks))
    n = sum(chunks)
    k_day = rng.integers(8000, 10526, size=n).astype(np.int32)            # Date32, 2 526 values
    k_part = rng.integers(-700, 700, size=n).astype(np.int64)              # negative base
    k_tag = [("N", "R", "A", "")[k] for k in rng.integers(0, 4, size=n)]
    i64 = rng.integers(-10**6, 10**6, size=n).astype(np.int64)
    big = rng.integers(-2**62, 2**62, size=n).astype(np.int64)             # SUM needs the exact 96-bit split lanes
    f64 = rng.normal(size=n) * 10.0 ** rng.integers(-6, 7, size=n)         # 13 decades of magnitude
    price = rng.integers(90000, 10494950, size=n).astype(np.float64) / 100.0
    disc = rng.integers(0, 11, size=n).astype(np.float64) / 100.0
    vk, va = rng.random(n) > 0.03, rng.random(n) > 0.2
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_DATE32, k_day, vk), (2, abi.DT_INT64, k_part), (3, abi.DT_UTF8, k_tag), (4, abi.DT_INT64, i64, va),
                                       (5, abi.DT_FLOAT64, f64), (6, abi.DT_FLOAT64, price), (7, abi.DT_FLOAT64, disc), (8, abi.DT_INT64, big)], chunks)
    A, F, O, E, col = abi.AggregateSpec, abi.Filter, abi.Operator, abi.Expr, abi.col
    q3 = [A.count_star(), A.sum(4), A.sum(col(6) * (1 - col(7)))]
    wide = [A.count_star(), A.count(4), A.sum(4), A.avg(4), A.min(4), A.max(4), A.total(4), A.sum(5), A.avg(5), A.min(5), A.max(5), A.sum(col(4) * col(5)),
            A.sum(col(6) * (1 - col(7)) * (1 + col(7))), A.total(5)]
    mid = [A.count_star(), A.sum(4), A.min(5), A.max(5), A.avg(6)]
    # (keys, aggregates, must it be the shared-image kernel?)  None: whichever route the capacity rules pick —
    # [1] × wide is too wide for the LDS image and stays on the sort-based route, [3] × wide is too wide for the
    # per-thread accumulator columns and moves to the shared image
    cases = [([1], q3, True), ([2], q3, True), ([2, 3], q3, None), ([3, 2], mid, None), ([3], wide, None), ([1], wide, None)]
    preds = [None, [F(4, O.GreaterThan(0))], E.not_(E.any_of([F(5, O.LessThan(0.0)), F(2, O.Equals(3))]))]
    routes = set()
    for keys, aggs, image in cases:
        for pred in preds if len(chunks) < 3 else preds[:2]:
            for order in (True, False):
                pq = rt.PreparedQuery(ht, pred, aggs, keys, order)
                if image is not None:
                    assert _image(pq) == image, (keys, pq.kernel_signature)
                routes.add((tuple(keys), len(aggs), _image(pq)))
                got, exp = pq.run(), orc.groupby(ot, pred, keys, aggs, order)
                assert [[k.value for k in r.keys] for r in got] == [[k.value for k in r.keys] for r in exp], (keys, order)
                for g, w in zip(got, exp):
                    assert_values(g.values, w.values, f"image group by {keys}", abs_floor=1e-6)
    assert any(r[2] for r in routes)
    # exact Int64 sums whose chain may overflow: the 96-bit split lanes are order-free too
    pq = rt.PreparedQuery(ht, None, [A.min(8), A.max(8), A.count_star()], [1], True)
    assert _image(pq)
    got, exp = pq.run(), orc.groupby(ot, None, [1], [A.min(8), A.max(8), A.count_star()], True)
    assert _flat(got) == _flat(exp)


@pytest.mark.parametrize("chunks", [[4096, 4097, 5], [65536, 70000, 30011]])
def test_counts_and_bounded_integer_sums_take_four_byte_image_cells(rt, orc, abi, chunks, monkeypatch):
    """Plan::CELL32: every lane a count or an integer sum the statistics bound below 2^31 per workgroup image (and the
    first-row lane): 4-byte cells — 15 156 groups × 3 lanes in ONE slice instead of three — with the same answers as the
    8-byte cells and the oracle; a column whose values could overflow 32 bits per image keeps the wide cells."""
    rng = np.random.default_rng(23 + len(chunks))
    n = sum(chunks)
    k_day = rng.integers(8000, 10526, size=n).astype(np.int32)
    k_tag = [("N", "R", "A", "O", "F", "")[k] for k in rng.integers(0, 6, size=n)]
    small = rng.integers(-50, 51, size=n).astype(np.int64)
    huge = rng.integers(-2**40, 2**40, size=n).astype(np.int64)
    f64 = rng.normal(size=n)
    va = rng.random(n) > 0.2
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_DATE32, k_day), (2, abi.DT_UTF8, k_tag), (3, abi.DT_INT64, small, va), (4, abi.DT_INT64, huge), (5, abi.DT_FLOAT64, f64),
                                       (6, abi.DT_INT64, small)], chunks)
    A, F, O = abi.AggregateSpec, abi.Filter, abi.Operator
    narrow_aggs = [A.count_star(), A.sum(6), A.count(3), A.sum(3), A.avg(3), A.count_nulls(3)]
    for keys in ([2, 1], [1]):
        for order in (True, False):
            for pred in (None, [F(6, O.GreaterThan(-10))]):
                pq = rt.PreparedQuery(ht, pred, narrow_aggs, keys, order)
                assert pq.kernel_signature.endswith(",-1,1>"), pq.kernel_signature
                got = pq.run()
                pq.close()
                want = orc.groupby(ot, pred, keys, narrow_aggs, order)
                assert _flat(got) == _flat(want), (keys, order)
                monkeypatch.setenv("LLKV_HIP_IMAGE_WIDE_CELLS", "1")
                pq = rt.PreparedQuery(ht, pred, narrow_aggs, keys, order)
                assert not pq.kernel_signature.endswith(",-1,1>")
                assert _flat(pq.run()) == _flat(got)
                pq.close()
                monkeypatch.delenv("LLKV_HIP_IMAGE_WIDE_CELLS")
    # the slices: (tag, day) = 6 × 2 526 groups × 5 lanes are two slices of 4-byte cells, four of 8-byte ones
    pq = rt.PreparedQuery(ht, None, narrow_aggs, [2, 1], True)
    assert pq.kernel_signature.endswith(",2,2,-1,1>"), pq.kernel_signature
    pq.close()
    monkeypatch.setenv("LLKV_HIP_IMAGE_WIDE_CELLS", "1")
    pq = rt.PreparedQuery(ht, None, narrow_aggs, [2, 1], True)
    assert pq.kernel_signature.endswith(",2,4>"), pq.kernel_signature
    pq.close()
    monkeypatch.delenv("LLKV_HIP_IMAGE_WIDE_CELLS")
    for wide in ([A.count_star(), A.sum(4)], [A.count_star(), A.sum(5)], [A.count_star(), A.min(6)]):
        pq = rt.PreparedQuery(ht, None, wide, [1], True)
        assert _image(pq) and not pq.kernel_signature.endswith(",-1,1>"), pq.kernel_signature
        pq.close()


def test_shared_image_results_do_not_depend_on_launch_geometry_or_shards(rt, abi, tpch):
    """Every lane of a shared-image plan is order-free (f64 sums are exact): the bits depend on the data alone — not
    on how many workgroups scan, not on the tile list, not on how the table is cut over ranks."""
    n = 400_003
    chunks = tpch.chunk_rows(n, 32768)
    d = tpch.gen_lineitem(n, 1.0)
    A, col = abi.AggregateSpec, abi.col
    S = tpch.LINEITEM_SCHEMA
    aggs = [A.count_star(), A.sum(S["l_quantity"][0]), A.sum(col(S["l_extendedprice"][0]) * (1 - col(S["l_discount"][0]))), A.max(S["l_quantity"][0])]
    cols = ["l_shipdate", "l_quantity", "l_extendedprice", "l_discount"]

    many = (np.arange(n, dtype=np.int64) * 7919) % 10_000  # 10 000 groups × 5 lanes: an image of three LDS-sized slices

    def stage(rank, world):
        ht = rt.HipTable(1, chunks, rank, world)
        lo = sum(chunks[:ht.first_chunk])
        for c in cols:
            fid, dt = S[c]
            ht.append_column(fid, dt, d[c][lo:lo + ht.local_rows])
        ht.append_column(99, abi.DT_INT64, many[lo:lo + ht.local_rows])
        return ht

    one = stage(0, 1)
    # 10 000 groups × 5 lanes: the image no longer fits the LDS, the groups are cut into slices, one scan each (",2,P>")
    wide = aggs[:3]
    pw = rt.PreparedQuery(one, None, wide, [99], False)
    tail = pw.kernel_signature[pw.kernel_signature.rindex(">,") + 2:-1].split(",")  # unroll, accumulator placement[, group slices]
    assert _image(pw) and len(tail) == 3 and tail[1] == "2" and int(tail[2]) > 1, (pw.route_note, pw.kernel_signature)
    os.environ["LLKV_HIP_GROUP_NO_IMAGE"] = "1"
    try:
        ps = rt.PreparedQuery(one, None, wide, [99], False)
    finally:
        del os.environ["LLKV_HIP_GROUP_NO_IMAGE"]
    gw, gs = pw.run(), ps.run()
    assert [r.keys[0].value for r in gw] == [r.keys[0].value for r in gs]  # first-appearance order on both routes
    for a, b in zip(gw, gs):
        for x, y in zip(a.values, b.values):
            assert x.value == y.value if isinstance(y.value, int) else abs(x.value - y.value) <= REL * abs(y.value)
    pq = rt.PreparedQuery(one, None, aggs, [S["l_shipdate"][0]], True)
    assert _image(pq), pq.route_note
    want = _flat(pq.run())
    assert len(want) > 2000
    ex1 = pq.read_exchange()
    assert _flat(pq.run()) == want
    for env, values in (("LLKV_HIP_IMAGE_WGS", ("1", "3", "77", "1000")), ("LLKV_HIP_TILE_ROWS", ("2048", "16384")), ("LLKV_HIP_UNROLL", ("1", "4"))):
        for v in values:
            os.environ[env] = v
            try:
                pg = rt.PreparedQuery(one, None, aggs, [S["l_shipdate"][0]], True)
                assert _flat(pg.run()) == want, (env, v)
                pg.close()
            finally:
                del os.environ[env]
    # shards: the exchange images summed as integers (the RCCL all-reduce) fold to the same bits; the statistics a
    # sharded table gets from the binding are the table-wide ones
    lib = rt.lib()
    for world in (2, 8):
        total, last = np.zeros_like(ex1).view(np.int64), None
        for rank in range(world):
            ht = stage(rank, world)
            ht.set_column_stats(99, 0, 9_999)
            for c in cols:  # the table-wide statistics, as share_metadata installs them on every rank
                fid, dt = S[c]
                if dt != abi.DT_FLOAT64:
                    ht.set_column_stats(fid, int(d[c].min()), int(d[c].max()))
                else:
                    ht.set_column_float_stats(fid, *one.local_column_float_stats(fid))
                    ht.set_column_all_finite(fid, one.local_column_all_finite(fid))
            pr = rt.PreparedQuery(ht, None, aggs, [S["l_shipdate"][0]], True)
            assert _image(pr), pr.route_note
            pr.launch()
            total += pr.read_exchange().view(np.int64)
            last = pr
        assert _flat(last.finish_from_host(total.view(np.uint64))) == want, world


def test_exact_sums_and_hostile_values(rt, orc, abi):
    """±∞, NaN and −0.0 inside an otherwise well-conditioned column: the statistics look at the finite values only, the
    plan stays on the shared-image kernel, and groups with a NaN / both infinities sum to NaN, with one infinity to it —
    like the oracle.  A column spanning 600 orders of magnitude (subnormals included) has no grid that resolves its
    smallest values: no exact sum, the sort-based route answers; so does an argument without any bound (a division)."""
    rng = np.random.default_rng(5)
    n = 50_000
    key = rng.integers(0, 500, size=n).astype(np.int64)
    v = rng.uniform(0.5, 1000.0, size=n) * rng.choice([-1.0, 1.0], size=n)
    v[(key == 7) & (rng.random(n) < 0.05)] = np.inf
    v[(key == 8) & (rng.random(n) < 0.05)] = -np.inf
    v[(key == 9) & (rng.random(n) < 0.05)] = np.nan
    v[key == 10] = np.where(rng.random(int((key == 10).sum())) < 0.5, np.inf, -np.inf)
    v[key == 11] = -0.0
    wild = rng.normal(size=n) * 10.0 ** rng.integers(-300, 300, size=n)
    wild[key == 12] = 5e-324
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT64, key), (2, abi.DT_FLOAT64, v), (3, abi.DT_FLOAT64, wild)], [20_000, 30_000])
    A, col = abi.AggregateSpec, abi.col

    def check(aggs, image, f64_cols, scale=1000.0):
        pq = rt.PreparedQuery(ht, None, aggs, [1], True)
        assert _image(pq) == image, pq.kernel_signature
        got, exp = pq.run(), orc.groupby(ot, None, [1], aggs, True)
        assert [r.keys[0].value for r in got] == [r.keys[0].value for r in exp]
        for g, w in zip(got, exp):
            for a in range(len(aggs)):
                x, y = g.values[a].value, w.values[a].value
                if a not in f64_cols or y is None or x is None:
                    assert x == y, (g.keys[0].value, a)
                elif np.isnan(y):
                    assert np.isnan(x), g.keys[0].value
                elif np.isinf(y):
                    assert x == y, g.keys[0].value
                else:  # sums of either sign: relative to the magnitude of what was added (≤ scale · rows of the group)
                    assert abs(x - y) <= REL * max(abs(y), scale * g.values[1].value), (g.keys[0].value, a, x, y)
        return got, exp

    got, exp = check([A.sum(2), A.count_star(), A.total(2), A.avg(2)], True, {0, 2, 3})
    assert np.isinf(exp[7].values[0].value) and np.isnan(exp[9].values[0].value) and np.isnan(exp[10].values[0].value)
    assert got[11].values[0].value == 0.0
    got, exp = check([A.sum(3), A.count_star()], False, {0}, scale=1e299)  # 600 decades: the sort-based route
    assert got[12].values[0].value == exp[12].values[0].value  # a group of subnormals only
    check([A.sum(col(2) / col(2)), A.count_star()], False, {0})


def test_order_free_f64_sums_are_bounded_error_not_exact(rt, orc, abi):
    """The order-free f64 sums of the shared-image / partitioned routes round every row to one grid (2^-30 of the column's
    smallest non-zero magnitude) or cut it into exactly-added levels: bit-identical over geometries, and within 1e-9 of the
    magnitude of what was added — NOT of a result that cancels.  Wide dynamic range (1e-6 … 1e9), mixed signs, groups whose
    large terms cancel exactly: the error against the exactly rounded sum (math.fsum) stays below 1e-9 · Σ|v| of the group,
    and the oracle's sequential sum is no closer to it than that either."""
    import math
    rng = np.random.default_rng(17)
    n = 120_000
    key = rng.integers(0, 300, size=n).astype(np.int64)
    mag = 10.0 ** rng.uniform(-6, 9, size=n)
    v = mag * rng.choice([-1.0, 1.0], size=n)
    # cancellation: in the groups 0..49 every large value gets its negative in the same group
    big = np.flatnonzero((key < 50) & (np.abs(v) > 1e6))
    half = len(big) // 2
    v[big[half:2 * half]] = -v[big[:half]]
    key[big[half:2 * half]] = key[big[:half]]
    ht, ot = stage_both(rt, orc, abi, [(1, abi.DT_INT64, key), (2, abi.DT_FLOAT64, v)], [50_000, 70_000])
    A = abi.AggregateSpec
    pq = rt.PreparedQuery(ht, None, [A.sum(2), A.count_star()], [1], True)
    assert _image(pq), pq.route_note
    got, exp = pq.run(), orc.groupby(ot, None, [1], [A.sum(2), A.count_star()], True)
    assert [r.keys[0].value for r in got] == [r.keys[0].value for r in exp]
    worst = 0.0
    for g, w in zip(got, exp):
        sel = key == g.keys[0].value
        exact, added = math.fsum(v[sel].tolist()), float(np.abs(v[sel]).sum())
        assert g.values[1].value == w.values[1].value == int(sel.sum())
        assert abs(g.values[0].value - exact) <= REL * added, (g.keys[0].value, g.values[0].value, exact, added)
        assert abs(w.values[0].value - exact) <= REL * added
        worst = max(worst, abs(g.values[0].value - exact) / added)
    assert worst < 5e-10
