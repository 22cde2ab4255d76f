#!/usr/bin/env python3
"""Q1 / Q6 at SF10 with llkv_hip_set_exact_f64_sums on and off: kernel time and the values (the exact sums are the correctly
rounded sums of the rows' f64 values; the default ones a fixed-order tree)."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
abi = importlib.import_module("rust-llkv_amd.abi"); rt = importlib.import_module("rust-llkv_amd.runtime"); tpch = importlib.import_module("rust-llkv_amd.tpch")
sf = sys.argv[1] if len(sys.argv) > 1 else "sf10"
rt.init(0)
rows, scale = tpch.LINEITEM_ROWS[sf], tpch.SCALE[sf]
q1, q6 = tpch.q1(), tpch.q6()
cols = sorted(set(q1.columns) | set(q6.columns))
li = tpch.gen_lineitem(rows, scale, cols)
t = rt.HipTable(1, tpch.chunk_rows(rows))
for c in cols:
    fid, dt = tpch.LINEITEM_SCHEMA[c]
    (t.append_utf8_column if dt == abi.DT_UTF8 else lambda f, v, d=dt: t.append_column(f, d, v))(fid, li[c])
out = {}
for exact in (False, True):
    rt.set_exact_f64_sums(exact)
    for q in (q1, q6):
        pq = rt.PreparedQuery(t, q.predicate, q.aggs, q.keys, q.order_by_keys)
        pq.set_profiling(True)
        for _ in range(20):
            res = pq.run()
        ms, n, _ = pq.kernel_time()
        out[f"{q.name}_{'exact' if exact else 'default'}"] = {"kernel_us": ms / n * 1e3, "lanes": pq.kernel_signature.count("SumF64Q2<"),
                                                            "first_row": [v.value for v in res[0].values][:6]}
        pq.close()
rt.set_exact_f64_sums(False)
print(json.dumps({"workload": f"exact_{sf}", "rows": rows, **out}))
