#!/usr/bin/env python3
"""Register-state scans with argument-only columns read late (Plan::EARLY) against the eager form (LLKV_HIP_SCAN_NO_LATE=1),
by predicate selectivity: SELECT sum(l_extendedprice), sum(l_extendedprice * l_discount) FROM lineitem WHERE l_quantity < Q at SF10."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
abi = importlib.import_module("rust-llkv_amd.abi"); rt = importlib.import_module("rust-llkv_amd.runtime"); tpch = importlib.import_module("rust-llkv_amd.tpch")
sf = sys.argv[1] if len(sys.argv) > 1 else "sf10"
rt.init(0)
rows, scale = tpch.LINEITEM_ROWS[sf], tpch.SCALE[sf]
cols = ["l_quantity", "l_extendedprice", "l_discount"]
li = tpch.gen_lineitem(rows, scale, cols)
t = rt.HipTable(1, tpch.chunk_rows(rows))
for c in cols:
    fid, dt = tpch.LINEITEM_SCHEMA[c]
    t.append_column(fid, dt, li[c])
A, F, O, col = abi.AggregateSpec, abi.Filter, abi.Operator, abi.col
S = tpch.LINEITEM_SCHEMA
out = {}
for q in (2, 6, 13, 24, 50, 51):
    pq = rt.PreparedQuery(t, [F(S["l_quantity"][0], O.LessThan(q))], [A.sum(S["l_extendedprice"][0]), A.sum(col(S["l_extendedprice"][0]) * col(S["l_discount"][0]))])
    pq.set_profiling(True)
    for _ in range(30):
        pq.run()
    ms, n, _ = pq.kernel_time()
    out[f"qty<{q}"] = {"selectivity": float((li["l_quantity"] < q).mean()), "kernel_us": ms / n * 1e3, "late": ",1," in pq.kernel_signature.rsplit(">", 1)[0][-8:]}
    pq.close()
print(json.dumps({"workload": f"late_{sf}", "rows": rows, "mode": "eager" if os.environ.get("LLKV_HIP_SCAN_NO_LATE") else "late", **out}))
