#!/usr/bin/env python3
"""What a 1-byte dictionary image of a low-cardinality Float64 column would buy a dense GROUP BY: the machinery exists for
Utf8 columns (codes in HBM, `DictNum<slot>` decodes them to numbers), so the same values staged as strings give the kernel such a layout
would run.  SF10 lineitem: sum / avg of l_quantity, l_discount, l_tax GROUP BY (l_returnflag, l_linestatus) WHERE l_shipdate <= D —
once over the Float64 columns (30 B per row), once over their dictionary-coded string forms (9 B per row).  Same sums (the strings
parse to the same doubles)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
abi = importlib.import_module("rust-llkv_amd.abi"); rt = importlib.import_module("rust-llkv_amd.runtime"); tpch = importlib.import_module("rust-llkv_amd.tpch")
sf = sys.argv[1] if len(sys.argv) > 1 else "sf10"
rt.init(0)
rows, scale = tpch.LINEITEM_ROWS[sf], tpch.SCALE[sf]
cols = ["l_quantity", "l_discount", "l_tax", "l_returnflag", "l_linestatus", "l_shipdate"]
li = tpch.gen_lineitem(rows, scale, cols)
S = tpch.LINEITEM_SCHEMA
A, F, O = abi.AggregateSpec, abi.Filter, abi.Operator
q1 = tpch.q1()
out = {"workload": f"value_image_probe_{sf}", "rows": rows}
import ctypes as C


def append_fixed_width_strings(t, field_id, values):
    """l_quantity … as 5-character decimal strings ("04.00", "00.04"), chunk by chunk, through the C call (no Python string per row)."""
    vals, codes = np.unique(values, return_inverse=True)
    names = np.array([("%05.2f" % v).encode() for v in vals], dtype="S5")
    assert all(float(n) == v for n, v in zip(names, vals)), "the strings must parse to the same doubles"
    offs, datas, at = [], [], 0
    for r in t.local_chunk_rows:
        datas.append(np.frombuffer(names[codes[at:at + r]].tobytes(), dtype=np.uint8).copy())
        offs.append((np.arange(r + 1, dtype=np.int32) * 5))
        at += r
    poff = (C.c_void_p * len(offs))(*[c.ctypes.data for c in offs])
    pdat = (C.c_void_p * len(datas))(*[c.ctypes.data for c in datas])
    rt.check(rt.lib().llkv_hip_table_append_utf8_column(t.handle, C.c_uint32(field_id), poff, pdat, C.c_uint32(len(offs)), None, C.c_uint32(0)))
    t._utf8_fields.add(field_id)


for form in ("float64", "dictionary"):
    t = rt.HipTable(1, tpch.chunk_rows(rows))
    for c in ("l_returnflag", "l_linestatus"):
        t.append_utf8_column(S[c][0], li[c])
    t.append_column(S["l_shipdate"][0], S["l_shipdate"][1], li["l_shipdate"])
    for c in ("l_quantity", "l_discount", "l_tax"):
        if form == "float64":
            t.append_column(S[c][0], abi.DT_FLOAT64, li[c])
        else:
            append_fixed_width_strings(t, S[c][0], li[c])
    aggs = [A.count_star()] + [f(S[c][0]) for c in ("l_quantity", "l_discount", "l_tax") for f in (A.sum, A.avg)]
    q = rt.PreparedQuery(t, q1.predicate, aggs, q1.keys, True)
    q.set_profiling(True)
    ts = []
    for i in range(8):
        t0 = time.perf_counter(); q.launch(0); res = q.finish(); ts.append(time.perf_counter() - t0)
    ms, n_k, name = q.kernel_time()
    out[form] = {"signature": q.kernel_signature[:200], "seconds_best": min(ts), "kernel_ms_avg": ms / max(1, n_k), "bytes_per_row": 8 * 3 + 2 + 4 if form == "float64" else 3 + 2 + 4,
                 "first_group": [v.value for v in res[0].values][:5]}
    q.close(); t.close()
print(json.dumps(out))
