#!/usr/bin/env python3
"""Selection-vector route at SF10 on one GPU: llkv_hip_filter_row_ids and
llkv_hip_scan_stream over the HBM-resident lineitem columns; the consumer only counts rows, so the time is
the library's: selection kernels, 65 536-row window gathers, device → pinned host copies, callbacks."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
abi = importlib.import_module("rust-llkv_amd.abi"); rt = importlib.import_module("rust-llkv_amd.runtime"); tpch = importlib.import_module("rust-llkv_amd.tpch")
sf = sys.argv[1] if len(sys.argv) > 1 else "sf10"
rt.init(0)
rows, scale = tpch.LINEITEM_ROWS[sf], tpch.SCALE[sf]
cols = ["l_quantity", "l_extendedprice", "l_discount", "l_shipdate"]
li = tpch.gen_lineitem(rows, scale, cols)
lt = rt.HipTable(1, tpch.chunk_rows(rows))
for c in cols: lt.append_column(tpch.LINEITEM_SCHEMA[c][0], tpch.LINEITEM_SCHEMA[c][1], li[c])
F, O, col = abi.Filter, abi.Operator, abi.col
preds = {"q6 (1.9 %)": tpch.q6().predicate, "l_quantity < 24 (46 %)": [F(tpch.L_QUANTITY, O.LessThan(24))], "all rows": [F(tpch.L_QUANTITY, O.GreaterThan(0))]}
projs = [tpch.L_EXTENDEDPRICE, tpch.L_DISCOUNT, col(tpch.L_EXTENDEDPRICE) * (1 - col(tpch.L_DISCOUNT))]
out = []
for name, pred in preds.items():
    n_sel = rt.filter_row_ids(lt, pred, count_only=True)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); n_sel = rt.filter_row_ids(lt, pred, count_only=True); ts.append(time.perf_counter() - t0)
    seen = [0, 0]
    def consume(b):
        seen[0] += int(b.num_rows); seen[1] += 1
    rt.scan_stream(lt, projs, pred, consume=consume)
    tt = []
    for _ in range(3):
        seen[0] = seen[1] = 0
        t0 = time.perf_counter(); rt.scan_stream(lt, projs, pred, consume=consume); tt.append(time.perf_counter() - t0)
    assert seen[0] == n_sel
    out.append({"predicate": name, "selected": n_sel, "filter_row_ids_s": min(ts), "filter_rows_per_s": rows / min(ts),
                "scan_stream_s": min(tt), "scan_rows_per_s": rows / min(tt), "windows": seen[1], "host_gbs": n_sel * 24 / min(tt) / 1e9})
print(json.dumps({"workload": f"scan_{sf}", "rows": rows, "cases": out}))
