#!/usr/bin/env python3
"""TPC-H Q3 shape at SF10 on one GPU (BASELINE.json configs[4], single-GPU form): time of
llkv_hip_join_groupby_topk with all inputs resident in HBM; algorithmic bytes per SURVEY.md §8(d)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
abi = importlib.import_module("rust-llkv_amd.abi"); rt = importlib.import_module("rust-llkv_amd.runtime"); tpch = importlib.import_module("rust-llkv_amd.tpch")
sf = sys.argv[1] if len(sys.argv) > 1 else "sf10"
rt.init(0)
rows, scale = tpch.LINEITEM_ROWS[sf], tpch.SCALE[sf]
D = tpch.DATE_1995_03_15
li = tpch.gen_lineitem(rows, scale, ["l_orderkey", "l_shipdate", "l_extendedprice", "l_discount"])
n_ord = tpch.orders_for_lineitems(rows); od = tpch.gen_orders(n_ord, scale)
n_cust = tpch.customers_for_scale(scale); cu = tpch.gen_customer(n_cust, scale)
lt = rt.HipTable(1, tpch.chunk_rows(rows))
for c in li: lt.append_column(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], li[c])
ot = rt.HipTable(2, tpch.chunk_rows(n_ord))
for c, (fid, dt) in tpch.ORDERS_SCHEMA.items(): ot.append_column(fid, dt, od[c])
ct = rt.HipTable(3, tpch.chunk_rows(n_cust))
ct.append_column(tpch.C_CUSTKEY, abi.DT_INT64, cu["c_custkey"])
ct.append_utf8_column(tpch.C_MKTSEGMENT, [tpch.SEGMENTS[c] for c in cu["c_mktsegment"]])
F, O, col = abi.Filter, abi.Operator, abi.col
rev = col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT))
def run():
    return rt.join_groupby_topk(lt, [F(tpch.L_SHIPDATE, O.GreaterThan(D))], tpch.L_ORDERKEY, ot, [F(tpch.O_ORDERDATE, O.LessThan(D))], tpch.O_ORDERKEY, rev,
                                payload_fields=[tpch.O_ORDERDATE, tpch.O_SHIPPRIORITY], limit=10, dim_fk=tpch.O_CUSTKEY, dim2=ct,
                                dim2_filters=[F(tpch.C_MKTSEGMENT, O.Equals("BUILDING"))], dim2_key=tpch.C_CUSTKEY)
out, total = run()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); out, total = run(); ts.append(time.perf_counter() - t0)
alg = rows * 28 + n_ord * 28 + n_cust * 9
best = min(ts)
print(json.dumps({"workload": f"q3_{sf}", "lineitem_rows": rows, "orders": n_ord, "customers": n_cust, "groups": total, "seconds_best": best, "seconds_all": ts,
                  "rows_per_s": rows / best, "algorithmic_bytes": alg, "gbs": alg / best / 1e9, "top": out[:3]}))
