#!/usr/bin/env python3
"""Host -> HBM staging of the seven Q1 columns of SF10 lineitem (2.28 GB) under the staging modes of csrc/memory.cpp:
pinned in place (hipHostRegister + direct DMA, the default) against the bounce copy through pinned rings, over lane counts."""
import importlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "one":
    import numpy as np
    abi = importlib.import_module("rust-llkv_amd.abi"); rt = importlib.import_module("rust-llkv_amd.runtime"); tpch = importlib.import_module("rust-llkv_amd.tpch")
    rt.init(0)
    q = tpch.q1()
    n = tpch.LINEITEM_ROWS["sf10"]
    d = tpch.gen_lineitem(n, 10.0, q.columns)
    best = None
    for rep in range(3):
        t = rt.HipTable(1, tpch.chunk_rows(n))
        b0, s0 = rt.staging_stats(); t0 = time.perf_counter()
        for c in q.columns:
            fid, dt = tpch.LINEITEM_SCHEMA[c]
            if dt == abi.DT_UTF8: t.append_utf8_column(fid, d[c])
            else: t.append_column(fid, dt, d[c])
        wall = time.perf_counter() - t0; b1, s1 = rt.staging_stats()
        r = {"wall_seconds": wall, "copy_seconds": s1 - s0, "copy_bytes": b1 - b0, "host_to_hbm_gbs": (b1 - b0) / (s1 - s0) / 1e9}
        if best is None or r["copy_seconds"] < best["copy_seconds"]: best = r
        res = rt.groupby(t, q.predicate, q.keys, q.aggs, True)
        t.close()
    best["q1_groups"] = len(res)
    print(json.dumps(best))
else:
    out = {}
    for mode in ("inplace", "bounce"):
        for lanes in (2, 4, 6, 8, 12, 16):
            env = dict(os.environ, LLKV_HIP_STAGE_MODE=mode, LLKV_HIP_STAGE_LANES=str(lanes))
            r = subprocess.run([sys.executable, __file__, "one"], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            out[f"{mode}_{lanes}"] = json.loads(line[-1]) if line else {"error": r.stderr[-300:]}
            print(mode, lanes, out[f"{mode}_{lanes}"], flush=True)
    print(json.dumps(out))
