#!/usr/bin/env python3
"""GROUP BY at SF10 on one GPU beyond the per-thread accumulators: l_orderkey (≈15 M groups: sort-based route), l_partkey
(2 M groups: partitioned route), l_shipdate (≈2500 groups: shared-image kernel) and the wide-state Q1 shape with every
aggregate doubled — inputs resident in HBM."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
abi = importlib.import_module("rust-llkv_amd.abi"); rt = importlib.import_module("rust-llkv_amd.runtime"); tpch = importlib.import_module("rust-llkv_amd.tpch")
sf = sys.argv[1] if len(sys.argv) > 1 else "sf10"
rt.init(0)
if os.environ.get("LLKV_BENCH_EXACT_SUMS"):  # the exact-sum planning option: f64 sums as two int64 fixed-point lanes
    rt.set_exact_f64_sums(True)
rows, scale = tpch.LINEITEM_ROWS[sf], tpch.SCALE[sf]
cols = ["l_orderkey", "l_partkey", "l_shipdate", "l_quantity", "l_extendedprice", "l_discount", "l_returnflag", "l_linestatus"]
li = tpch.gen_lineitem(rows, scale, cols)
t = rt.HipTable(1, tpch.chunk_rows(rows))
for c in cols:
    fid, dt = tpch.LINEITEM_SCHEMA[c]
    (t.append_utf8_column if dt == abi.DT_UTF8 else lambda f, v, d=dt: t.append_column(f, d, v))(fid, li[c])
A, col = abi.AggregateSpec, abi.col
S = tpch.LINEITEM_SCHEMA
rev = col(S["l_extendedprice"][0]) * (1 - col(S["l_discount"][0]))
narrow = [A.count_star(), A.sum(S["l_quantity"][0]), A.sum(rev)]
wide = [A.count_star(), A.sum(S["l_quantity"][0]), A.avg(S["l_quantity"][0]), A.min(S["l_quantity"][0]), A.max(S["l_quantity"][0]), A.sum(S["l_extendedprice"][0]),
        A.avg(S["l_extendedprice"][0]), A.min(S["l_extendedprice"][0]), A.max(S["l_extendedprice"][0]), A.sum(rev), A.avg(S["l_discount"][0]), A.total(S["l_discount"][0])]
mid = [A.count_star(), A.sum(S["l_quantity"][0]), A.sum(rev), A.avg(S["l_discount"][0])]
out = {}
only = sys.argv[2] if len(sys.argv) > 2 else ""  # "image_only": just the shared-image cases (counter runs)
for name, keys, aggs in (("by_orderkey", [S["l_orderkey"][0]], narrow), ("by_partkey", [S["l_partkey"][0]], narrow), ("by_shipdate", [S["l_shipdate"][0]], narrow),
                         ("by_shipdate_count_only", [S["l_shipdate"][0]], narrow[:1]), ("by_shipdate_4aggs", [S["l_shipdate"][0]], mid), ("by_flag_status_shipdate", [S["l_returnflag"][0], S["l_linestatus"][0], S["l_shipdate"][0]], narrow[:2]),
                         ("q1_wide_state", [S["l_returnflag"][0], S["l_linestatus"][0]], wide)):
    if only == "image_only" and name not in ("by_shipdate", "by_shipdate_count_only"):
        continue
    if only and only != "image_only" and name not in only.split(","):
        continue
    q = rt.PreparedQuery(t, None, aggs, keys, True)
    image = q.kernel_signature.endswith(",2>")
    ts = []
    q.set_profiling(True)
    for i in range(5):
        t0 = time.perf_counter(); q.launch(0)  # device pipeline …
        assert rt.lib().llkv_hip_query_finish(q._h, None) == 0  # … + copy-out and host fold of every group
        ts.append(time.perf_counter() - t0)
        ng = rt.lib().llkv_hip_query_num_groups(q._h)
    kms, kn, _ = q.kernel_time()
    best = min(ts)
    out[name] = {"route": q.route_note.split(" (")[0], "groups": int(ng), "seconds_best": best, "rows_per_s": rows / best}
    if q.algorithmic_bytes:
        out[name].update({"alg_bytes": q.algorithmic_bytes, "gbs_end_to_end": q.algorithmic_bytes / best / 1e9})
        if kn:
            out[name].update({"kernel_ms": kms / kn, "kernel_gbs": q.algorithmic_bytes / (kms / kn) / 1e6, "frac_of_8TBs": q.algorithmic_bytes / (kms / kn) / 1e6 / 8000.0})
    q.close()
print(json.dumps({"workload": f"groupby_{sf}", "rows": rows, **out}))
