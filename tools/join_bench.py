#!/usr/bin/env python3
"""llkv_hip_join_stream at SF10 on one GPU: lineitem ⋈ orders on the order key (every lineitem
matches one order: 59 986 052 pairs), pair batches delivered to a consumer that only counts them."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
abi = importlib.import_module("rust-llkv_amd.abi"); rt = importlib.import_module("rust-llkv_amd.runtime"); tpch = importlib.import_module("rust-llkv_amd.tpch")
sf = sys.argv[1] if len(sys.argv) > 1 else "sf10"
rt.init(0)
rows, scale = tpch.LINEITEM_ROWS[sf], tpch.SCALE[sf]
li = tpch.gen_lineitem(rows, scale, ["l_orderkey"])
n_ord = tpch.orders_for_lineitems(rows); od = tpch.gen_orders(n_ord, scale)
lt = rt.HipTable(1, tpch.chunk_rows(rows)); lt.append_column(tpch.L_ORDERKEY, abi.DT_INT64, li["l_orderkey"])
ot = rt.HipTable(2, tpch.chunk_rows(n_ord)); ot.append_column(tpch.O_ORDERKEY, abi.DT_INT64, od["o_orderkey"]); ot.append_column(tpch.O_CUSTKEY, abi.DT_INT64, od["o_custkey"])
out = []
for name, (l, r, keys, jt, bs) in {
    "lineitem x orders, inner, batch 8192": (lt, ot, [(tpch.L_ORDERKEY, tpch.O_ORDERKEY)], abi.JOIN_INNER, 8192),
    "lineitem x orders, inner, batch 65536": (lt, ot, [(tpch.L_ORDERKEY, tpch.O_ORDERKEY)], abi.JOIN_INNER, 65536),
    "lineitem x orders, semi": (lt, ot, [(tpch.L_ORDERKEY, tpch.O_ORDERKEY)], abi.JOIN_SEMI, 65536),
    "orders x orders on (orderkey, custkey), inner": (ot, ot, [(tpch.O_ORDERKEY, tpch.O_ORDERKEY), (tpch.O_CUSTKEY, tpch.O_CUSTKEY)], abi.JOIN_INNER, 65536),
}.items():
    seen = [0, 0]
    def consume(n):
        seen[0] += n; seen[1] += 1
    rt.join_stream(l, r, keys, jt, bs, consume=consume)
    ts = []
    for _ in range(2):
        seen[0] = seen[1] = 0
        t0 = time.perf_counter(); rt.join_stream(l, r, keys, jt, bs, consume=consume); ts.append(time.perf_counter() - t0)
    out.append({"case": name, "pairs": seen[0], "batches": seen[1], "seconds": min(ts), "probe_rows_per_s": l.local_rows / min(ts), "pairs_per_s": seen[0] / min(ts)})
print(json.dumps({"workload": f"join_{sf}", "cases": out}))
