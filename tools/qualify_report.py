#!/usr/bin/env python3
"""Q1 / Q6 / Q3 with the default substitution parameters (rust-llkv_amd/qualify.py: render_query) on the GPU path against
the oracle's rows, column by column under the reference's qualification rule (llkv-tpch/src/qualification.rs:708-745:
strings / integers exact, `sum` columns as exact decimals after Decimal::from_f64 = 15 significant digits, `avg` columns
within an ABSOLUTE 1e-9) — on the synthetic TPC-H-shaped data (prices and discounts are NOT dyadic: every f64 sum rounds).
Prints one JSON document; profiles/r02/qualification_report.json is its output."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
abi = importlib.import_module("rust-llkv_amd.abi"); rt = importlib.import_module("rust-llkv_amd.runtime"); tpch = importlib.import_module("rust-llkv_amd.tpch")
qual = importlib.import_module("rust-llkv_amd.qualify")
from oracle import oracle as orc
rt.init(0)
out = {}
for sf in sys.argv[1:] or ["sf0.01", "sf1"]:
    rows, scale = tpch.LINEITEM_ROWS[sf], tpch.SCALE[sf]
    d = tpch.gen_lineitem(rows, scale)
    ht, ot = rt.HipTable(1, tpch.chunk_rows(rows)), orc.OracleTable(rows)
    for name, (fid, dt) in tpch.LINEITEM_SCHEMA.items():
        ot.add(fid, dt, d[name])
        ht.append_utf8_column(fid, d[name]) if dt == abi.DT_UTF8 else ht.append_column(fid, dt, d[name])
    res = {}
    q1 = qual.render_query(tpch, abi, 1)
    cells = lambda rws: [[k.value for k in r.keys] + [v.value for v in r.values] for r in rws]
    res["q1"] = qual.compare_report(cells(orc.groupby(ot, q1.predicate, q1.keys, q1.aggs, True)), cells(rt.groupby(ht, q1.predicate, q1.keys, q1.aggs, True)), qual.Q1_TOKENS)
    # Q1 against the decimal-exact answer (prices are cents, discounts and taxes hundredths: integer arithmetic, no
    # rounding anywhere), each `sum` cut to the 15 digits Decimal::from_f64 keeps: the sequential f64 chain of the
    # reference (oracle), the GPU path's fixed-order tree, and the GPU path with llkv_hip_set_exact_f64_sums
    from decimal import Decimal
    cutoff = qual._date32("1998-12-01") - int(qual.render_parameters(1)[0])
    keep = d["l_shipdate"] <= cutoff
    pc = np.rint(d["l_extendedprice"] * 100).astype(np.int64); dh = np.rint(d["l_discount"] * 100).astype(np.int64); th = np.rint(d["l_tax"] * 100).astype(np.int64)
    exact_rows = []
    for r in orc.groupby(ot, q1.predicate, q1.keys, q1.aggs, True):
        f, st = r.keys[0].value, r.keys[1].value
        m = keep & (d["l_returnflag"] == ord(f)) & (d["l_linestatus"] == ord(st))
        n, sq = int(m.sum()), int(d["l_quantity"][m].sum())
        sp, sd = int(pc[m].sum()), int(dh[m].sum())
        sdp, sch = int((pc[m] * (100 - dh[m])).sum()), int((pc[m] * (100 - dh[m]) * (100 + th[m])).sum())
        exact_rows.append([f, st, sq, qual.decimal_15_digits(Decimal(sp) / 100), qual.decimal_15_digits(Decimal(sdp) / 10**4), qual.decimal_15_digits(Decimal(sch) / 10**6),
                           sq / n, float(Decimal(sp) / 100 / n), float(Decimal(sd) / 100 / n), n])
    res["q1_vs_decimal_exact"] = {"reference_order_chain": qual.compare_report(exact_rows, cells(orc.groupby(ot, q1.predicate, q1.keys, q1.aggs, True)), qual.Q1_TOKENS),
                                  "gpu_default": qual.compare_report(exact_rows, cells(rt.groupby(ht, q1.predicate, q1.keys, q1.aggs, True)), qual.Q1_TOKENS)}
    rt.set_exact_f64_sums(True)
    try:
        res["q1_vs_decimal_exact"]["gpu_exact_f64_sums"] = qual.compare_report(exact_rows, cells(rt.groupby(ht, q1.predicate, q1.keys, q1.aggs, True)), qual.Q1_TOKENS)
    finally:
        rt.set_exact_f64_sums(False)
    q6 = qual.render_query(tpch, abi, 6)
    res["q6"] = qual.compare_report([[orc.aggregate(ot, q6.predicate, q6.aggs)[0].value]], [[rt.aggregate(ht, q6.predicate, q6.aggs)[0].value]], qual.Q6_TOKENS)
    n_ord = tpch.orders_for_lineitems(rows); od = tpch.gen_orders(n_ord, scale)
    n_cust = tpch.customers_for_scale(scale); cu = tpch.gen_customer(n_cust, scale)
    seg = [tpch.SEGMENTS[c] for c in cu["c_mktsegment"]]
    hot, oot = rt.HipTable(2, tpch.chunk_rows(n_ord)), orc.OracleTable(n_ord)
    for c, (fid, dt) in tpch.ORDERS_SCHEMA.items():
        hot.append_column(fid, dt, od[c]); oot.add(fid, dt, od[c])
    hct, oct_ = rt.HipTable(3, tpch.chunk_rows(n_cust)), orc.OracleTable(n_cust)
    hct.append_column(tpch.C_CUSTKEY, abi.DT_INT64, cu["c_custkey"]); oct_.add(tpch.C_CUSTKEY, abi.DT_INT64, cu["c_custkey"])
    hct.append_utf8_column(tpch.C_MKTSEGMENT, seg); oct_.add(tpch.C_MKTSEGMENT, abi.DT_UTF8, seg)
    q3 = qual.render_query(tpch, abi, 3)
    got, _ = rt.join_groupby_topk(fact=ht, dim=hot, dim2=hct, **q3)
    # the reference's order of additions — an order's lineitems in scan order — restated with numpy (np.add.at applies
    # its updates one by one, in index order); the composition of the oracle's operators is tests/test_gpu_parity.py::
    # test_q3_join_groupby_topk_matches_oracle
    D = qual._date32(qual.render_parameters(3)[1])
    cust = cu["c_custkey"][np.array(seg) == qual.render_parameters(3)[0]]
    o_ok = (od["o_orderdate"] < D) & np.isin(od["o_custkey"], cust)
    keys = od["o_orderkey"][o_ok]
    l_ok = (d["l_shipdate"] > D) & np.isin(d["l_orderkey"], keys)
    uniq, inv = np.unique(d["l_orderkey"][l_ok], return_inverse=True)
    rev = np.zeros(len(uniq))
    np.add.at(rev, inv, d["l_extendedprice"][l_ok] * (1 - d["l_discount"][l_ok]))
    pos = np.searchsorted(od["o_orderkey"], uniq)  # the generator emits orders in key order
    order = np.lexsort((od["o_orderdate"][pos], -rev))[:10]
    want = [(int(uniq[i]), float(rev[i]), 0, int(od["o_orderdate"][pos[i]]), int(od["o_shippriority"][pos[i]])) for i in order]
    row = lambda r: [r[0], r[1], r[3], r[4]]  # l_orderkey, revenue, o_orderdate, o_shippriority (Q3_TOKENS)
    res["q3"] = qual.compare_report([row(r) for r in want], [row(r) for r in got], qual.Q3_TOKENS)
    res["q3"]["revenue_bits_equal"] = [np.float64(r[1]).tobytes() for r in want] == [np.float64(r[1]).tobytes() for r in got]
    res["q3"]["same_orders_in_the_same_order"] = [r[0] for r in want] == [r[0] for r in got]
    out[sf] = res
    ht.close(); hot.close(); hct.close()
print(json.dumps(out, indent=1))
