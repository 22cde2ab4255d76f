"""GPU (-m gpu): join → GROUP BY with ANY aggregate list (llkv_hip_join_groupby_prepare / _rows) against the oracle's
`join_groupby` — the executor's multi-table route (try_execute_hash_join llkv-executor/src/lib.rs:3780-4052 →
execute_group_by_from_batches :4544-4755, ORDER BY :13762-13868, LIMIT :10925-10955).  Keys, payload, counts, integer sums,
MIN / MAX, decimals and the order are exact; f64 sums / averages within 1e-9 (a group's rows are added by a wave's tree on the
GPU, strictly left to right in the reference)."""
import numpy as np
import pytest

from conftest import same_value

pytestmark = pytest.mark.gpu
REL = 1e-9


def q3_tables(rt, orc, abi, tpch, rows, scale, chunk=65536, rank=0, world=1, decimal=False):
    li = tpch.gen_lineitem(rows, scale, ["l_orderkey", "l_shipdate", "l_extendedprice", "l_discount", "l_quantity", "l_tax"])
    if decimal:
        li = tpch.lineitem_as_decimal(li)
    n_ord = tpch.orders_for_lineitems(rows)
    od = tpch.gen_orders(n_ord, scale)
    n_cust = tpch.customers_for_scale(scale)
    cu = tpch.gen_customer(n_cust, scale)
    chunks = tpch.chunk_rows(rows, chunk)
    lt = rt.HipTable(1, chunks, rank, world)
    lo = sum(chunks[:lt.first_chunk])
    hi = lo + lt.local_rows
    for c in li:
        fid, dt = tpch.LINEITEM_SCHEMA[c][0], tpch.lineitem_dtype(c, decimal)
        if dt == abi.DT_DECIMAL128:
            lt.append_decimal128_column(fid, 15, 2, li[c][lo:hi])
        else:
            lt.append_column(fid, dt, li[c][lo:hi])
    if world > 1:  # table-wide statistics, as share_metadata installs them
        for c in li:
            if li[c].dtype in (np.int64, np.int32):
                lt.set_column_stats(tpch.LINEITEM_SCHEMA[c][0], int(li[c].min()), int(li[c].max()))
    ot_ = rt.HipTable(2, tpch.chunk_rows(n_ord, chunk))
    for c, (fid, dt) in tpch.ORDERS_SCHEMA.items():
        ot_.append_column(fid, dt, od[c])
    ct = rt.HipTable(3, tpch.chunk_rows(n_cust, chunk))
    ct.append_column(tpch.C_CUSTKEY, abi.DT_INT64, cu["c_custkey"])
    ct.append_utf8_column(tpch.C_MKTSEGMENT, [tpch.SEGMENTS[c] for c in cu["c_mktsegment"]])
    oracle = None
    if orc is not None:
        lo_t = orc.OracleTable(rows)
        for c in li:
            fid, dt = tpch.LINEITEM_SCHEMA[c][0], tpch.lineitem_dtype(c, decimal)
            if dt == abi.DT_DECIMAL128:
                lo_t.add(fid, dt, li[c], precision=15, scale=2)
            else:
                lo_t.add(fid, dt, li[c])
        oo = orc.OracleTable(n_ord)
        for c, (fid, dt) in tpch.ORDERS_SCHEMA.items():
            oo.add(fid, dt, od[c])
        oc = orc.OracleTable(n_cust).add(tpch.C_CUSTKEY, abi.DT_INT64, cu["c_custkey"]).add(tpch.C_MKTSEGMENT, abi.DT_UTF8, [tpch.SEGMENTS[c] for c in cu["c_mktsegment"]])
        oracle = (lo_t, oo, oc)
    return (lt, ot_, ct), oracle


def q3_args(abi, tpch):
    F, O = abi.Filter, abi.Operator
    D = tpch.DATE_1995_03_15
    return dict(fact_filters=[F(tpch.L_SHIPDATE, O.GreaterThan(D))], fact_key=tpch.L_ORDERKEY, dim_filters=[F(tpch.O_ORDERDATE, O.LessThan(D))],
                dim_key=tpch.O_ORDERKEY, dim_fk=tpch.O_CUSTKEY, dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))], dim2_key=tpch.C_CUSTKEY)


def same_rows(got, want, ctx=""):
    assert len(got) == len(want), (ctx, len(got), len(want))
    for g, w in zip(got, want):
        assert (g.key, g.payload, g.group_index) == (w[0], w[1], w[3]), (ctx, g, w)
        for x, y in zip(g.values, w[2]):
            assert x.dtype == y.dtype and x.is_null == y.is_null, (ctx, g.key, x, y)
            if isinstance(y.value, float):
                assert same_value(x.value, y.value, REL), (ctx, g.key, x, y)
            else:
                assert x == y, (ctx, g.key, x, y)


SHAPES = {
    "two sums and a count": lambda A, col, t: [A.sum(col(t.L_EXTENDEDPRICE) * (1 - col(t.L_DISCOUNT))), A.sum(t.L_QUANTITY), A.count_star()],
    "avg and min": lambda A, col, t: [A.avg(t.L_EXTENDEDPRICE), A.min(t.L_QUANTITY), A.max(col(t.L_EXTENDEDPRICE) * col(t.L_TAX))],
    "count only": lambda A, col, t: [A.count_star()],
}


@pytest.mark.parametrize("shape", list(SHAPES))
def test_join_groupby_aggregate_lists_match_oracle(rt, orc, abi, tpch, shape):
    """The Q3 star (customer ⋉ orders ⋈ lineitem) under three aggregate lists and several ORDER BY / LIMIT forms."""
    rows, scale = 120_000, 0.02
    (lt, ot_, ct), (lo_t, oo, oc) = q3_tables(rt, orc, abi, tpch, rows, scale)
    A, col = abi.AggregateSpec, abi.col
    aggs = SHAPES[shape](A, col, tpch)
    args = q3_args(abi, tpch)
    jq = rt.JoinGroupBy(lt, args["fact_filters"], args["fact_key"], ot_, args["dim_filters"], args["dim_key"], aggs, dim_fk=args["dim_fk"], dim2=ct,
                        dim2_filters=args["dim2_filters"], dim2_key=args["dim2_key"])
    assert jq.route_note.startswith("join → GROUP BY")
    jq.launch()
    jq.finish_only()
    pay = [tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY]
    orders = [
        ([(abi.JOIN_ORDER_AGGREGATE, 0, True), (abi.JOIN_ORDER_PAYLOAD, 0, False)], 10),
        ([(abi.JOIN_ORDER_PAYLOAD, 0, True), (abi.JOIN_ORDER_AGGREGATE, len(aggs) - 1, False), (abi.JOIN_ORDER_KEY, 0, True)], 25),
        ([(abi.JOIN_ORDER_KEY, 0, False)], None),
        ([], 7),
    ]
    for order, limit in orders:
        got, total = jq.result(pay, order, limit)
        want, want_total = orc.join_groupby(lo_t, args["fact_filters"], args["fact_key"], oo, args["dim_filters"], args["dim_key"], aggs, payload_fields=pay,
                                            order=order, limit=limit, dim_fk=args["dim_fk"], dim2=oc, dim2_filters=args["dim2_filters"], dim2_key=args["dim2_key"])
        assert total == want_total and total > 100
        same_rows(got, want, f"{shape} order={order} limit={limit}")
    jq.close()


def test_join_groupby_without_dim2_and_with_decimal_arguments(rt, orc, abi, tpch):
    """orders ⋈ lineitem alone, money columns as DECIMAL(15,2): the aggregate arguments take the PlanValue Decimal arm — sums of
    scale-4 products, AVG rounded half away from zero, MIN / MAX — every cell bit-equal to the oracle's."""
    rows, scale = 60_000, 0.01
    (lt, ot_, ct), (lo_t, oo, oc) = q3_tables(rt, orc, abi, tpch, rows, scale, decimal=True)
    A, col, F, O = abi.AggregateSpec, abi.col, abi.Filter, abi.Operator
    aggs = [A.sum(col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT))), A.avg(tpch.L_EXTENDEDPRICE), A.min(tpch.L_QUANTITY), A.max(tpch.L_TAX), A.count(tpch.L_DISCOUNT)]
    ff, df = [F(tpch.L_SHIPDATE, O.GreaterThan(tpch.DATE_1995_03_15))], [F(tpch.O_ORDERDATE, O.LessThan(tpch.DATE_1995_06_17))]
    jq = rt.JoinGroupBy(lt, ff, tpch.L_ORDERKEY, ot_, df, tpch.O_ORDERKEY, aggs)
    jq.launch()
    jq.finish_only()
    order = [(abi.JOIN_ORDER_AGGREGATE, 0, True), (abi.JOIN_ORDER_KEY, 0, False)]
    got, total = jq.result([tpch.O_ORDERDATE], order, 50)
    want, want_total = orc.join_groupby(lo_t, ff, tpch.L_ORDERKEY, oo, df, tpch.O_ORDERKEY, aggs, payload_fields=[tpch.O_ORDERDATE], order=order, limit=50)
    assert total == want_total and total > 50
    same_rows(got, want, "decimal arguments")
    assert got[0].values[0].dtype == abi.DT_DECIMAL128 and got[0].values[0].scale == 4


def test_join_groupby_agrees_with_the_q3_pipeline(rt, abi, tpch):
    """The general route over Q3's own shape names the same top ten orders as the hand-tuned pipeline (whose sums are the
    reference's bits; the general route's are within 1e-9 of them)."""
    rows, scale = 600_000, 0.1
    (lt, ot_, ct), _ = q3_tables(rt, None, abi, tpch, rows, scale, chunk=131072)
    A, col = abi.AggregateSpec, abi.col
    args = q3_args(abi, tpch)
    rev = col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT))
    want, want_total = rt.join_groupby_topk(lt, args["fact_filters"], args["fact_key"], ot_, args["dim_filters"], args["dim_key"], rev,
                                            payload_fields=[tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], limit=10, dim_fk=args["dim_fk"], dim2=ct,
                                            dim2_filters=args["dim2_filters"], dim2_key=args["dim2_key"])
    jq = rt.JoinGroupBy(lt, args["fact_filters"], args["fact_key"], ot_, args["dim_filters"], args["dim_key"], [A.sum(rev), A.count_star()], dim_fk=args["dim_fk"],
                        dim2=ct, dim2_filters=args["dim2_filters"], dim2_key=args["dim2_key"])
    jq.launch()
    jq.finish_only()
    got, total = jq.result([tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], [(abi.JOIN_ORDER_AGGREGATE, 0, True), (abi.JOIN_ORDER_PAYLOAD, 0, False)], 10)
    assert total == want_total
    assert [g.key for g in got] == [w[0] for w in want]
    for g, w in zip(got, want):
        assert same_value(g.values[0].value, w[1], REL) and g.values[1].value == w[2] and g.payload == [w[3], w[4]]


@pytest.mark.parametrize("world", [2, 8])
def test_join_groupby_over_a_sharded_fact_table(rt, abi, tpch, world):
    """Fact table sharded by chunk, dimension tables replicated: every rank groups its own rows, the partial groups are merged lane
    by lane in rank order (what llkv_hip_query_finish_sharded does over the communicator — here by hand, the ranks emulated one
    after the other on one device): keys, order, payload, counts, integer sums and MIN / MAX are those of the whole table, f64 sums
    within 1e-9."""
    rows, scale = 200_000, 0.04
    A, col = abi.AggregateSpec, abi.col
    aggs = [A.sum(col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT))), A.sum(tpch.L_QUANTITY), A.min(tpch.L_QUANTITY), A.count_star()]
    args = q3_args(abi, tpch)
    order = [(abi.JOIN_ORDER_AGGREGATE, 1, True), (abi.JOIN_ORDER_AGGREGATE, 0, True)]
    pay = [tpch.O_ORDERDATE]

    def prepared(rank, w):
        (lt, ot_, ct), _ = q3_tables(rt, None, abi, tpch, rows, scale, chunk=8192, rank=rank, world=w)
        jq = rt.JoinGroupBy(lt, args["fact_filters"], args["fact_key"], ot_, args["dim_filters"], args["dim_key"], aggs, dim_fk=args["dim_fk"], dim2=ct,
                            dim2_filters=args["dim2_filters"], dim2_key=args["dim2_key"])
        jq.launch()
        jq.finish_only()
        return jq, (lt, ot_, ct)
    whole, keep0 = prepared(0, 1)
    want, want_total = whole.result(pay, order, 40)
    parts, qs = [], []
    for r in range(world):
        jq, keep = prepared(r, world)
        parts.append(jq.partial_groups())
        qs.append((jq, keep))
    qs[0][0].merge_groups(parts)
    got, total = qs[0][0].result(pay, order, 40)
    assert total == want_total and sum(p[2].shape[0] for p in parts) >= total  # (groups that straddle two shards arrive twice)
    assert [(g.key, g.payload, g.group_index) for g in got] == [(w.key, w.payload, w.group_index) for w in want]
    for g, w in zip(got, want):
        assert same_value(g.values[0].value, w.values[0].value, REL)
        assert [v.value for v in g.values[1:]] == [v.value for v in w.values[1:]]


def test_join_groupby_hands_back_what_it_does_not_take(rt, abi):
    """A dimension key that occurs twice among the qualifying rows, a key column without statistics-bounded range (Float keys): the
    caller keeps its own route."""
    n = 1000
    fact = rt.HipTable(1, [n])
    fact.append_column(1, abi.DT_INT64, np.arange(n, dtype=np.int64) % 50)
    fact.append_column(2, abi.DT_FLOAT64, np.ones(n))
    dim = rt.HipTable(2, [60])
    dim.append_column(1, abi.DT_INT64, np.arange(60, dtype=np.int64) % 30)  # every key twice
    dim.append_column(2, abi.DT_FLOAT64, np.zeros(60))
    A = abi.AggregateSpec
    with pytest.raises(abi.LlkvError) as e:
        rt.JoinGroupBy(fact, [], 1, dim, [], 1, [A.count_star()])
    assert e.value.kind == "Unsupported" and "unique" in e.value.message
    with pytest.raises(abi.LlkvError) as e:
        rt.JoinGroupBy(fact, [], 1, dim, [], 2, [A.count_star()])
    assert e.value.kind == "Unsupported"
    # a filter that leaves no dimension row: no group, no error
    jq = rt.JoinGroupBy(fact, [], 1, dim, [abi.Filter(1, abi.Operator.GreaterThan(1000))], 1, [A.count_star(), A.sum(2)])
    jq.launch()
    jq.finish_only()
    assert jq.result([], [], None) == ([], 0)
