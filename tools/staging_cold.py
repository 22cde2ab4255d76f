#!/usr/bin/env python3
"""Why the FIRST table of a process stages slower than the second (bench.py `staging` vs `staging.second_table`): the seven Q1
columns of SF10 lineitem into table A (cold), B (other host memory, never registered before) and C (A's memory again), per
column; `warm` first stages a 64 MB column so that whatever the process pays once is paid before A."""
import importlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] in ("plain", "warm"):
    import numpy as np
    abi = importlib.import_module("rust-llkv_amd.abi"); rt = importlib.import_module("rust-llkv_amd.runtime"); tpch = importlib.import_module("rust-llkv_amd.tpch")
    q = tpch.q1()
    n = tpch.LINEITEM_ROWS["sf10"]
    d1 = tpch.gen_lineitem(n, 10.0, q.columns)
    d2 = {k: v.copy() for k, v in d1.items()}
    t_init = time.perf_counter(); rt.init(0); t_init = time.perf_counter() - t_init
    out = {"init_seconds": t_init}
    if sys.argv[1] == "warm":
        t0 = time.perf_counter()
        w = rt.HipTable(9, [8 << 20])
        w.append_column(1, abi.DT_INT64, np.arange(8 << 20, dtype=np.int64))
        w.close()
        out["warmup_seconds"] = time.perf_counter() - t0
    tables = []
    for label, d in (("A", d1), ("B", d2), ("C", d1)):
        t = rt.HipTable(len(tables) + 1, tpch.chunk_rows(n)); tables.append(t)
        cols = []
        b0, s0 = rt.staging_stats(); t0 = time.perf_counter()
        for c in q.columns:
            fid, dt = tpch.LINEITEM_SCHEMA[c]
            bb, ss = rt.staging_stats(); tt = time.perf_counter()
            if dt == abi.DT_UTF8: t.append_utf8_column(fid, d[c])
            else: t.append_column(fid, dt, d[c])
            be, se = rt.staging_stats()
            cols.append({"col": c, "wall_ms": round((time.perf_counter() - tt) * 1e3, 2), "copy_ms": round((se - ss) * 1e3, 2), "mb": round((be - bb) / 1e6, 1)})
        wall = time.perf_counter() - t0; b1, s1 = rt.staging_stats()
        out[label] = {"wall_seconds": wall, "copy_seconds": s1 - s0, "host_to_hbm_gbs": (b1 - b0) / (s1 - s0) / 1e9, "columns": cols}
    print(json.dumps(out))
else:
    for mode in ("plain", "warm"):
        r = subprocess.run([sys.executable, __file__, mode], capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(mode, line[-1] if line else r.stderr[-600:], flush=True)
