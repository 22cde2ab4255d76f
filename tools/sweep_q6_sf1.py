#!/usr/bin/env python3
"""Q6 SF1 (configs[1], a 168 MB table): workgroup count of the register-state scan.  0 = one workgroup per 4 096-row tile
(1 465 of them); g > 0 = g workgroups, each streaming a contiguous run of tiles (results are bit-identical for every g).
Kernel time (HIP events) and host-timed step, one process per setting."""
import importlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "one":
    abi = importlib.import_module("rust-llkv_amd.abi"); rt = importlib.import_module("rust-llkv_amd.runtime"); tpch = importlib.import_module("rust-llkv_amd.tpch")
    rt.init(0)
    q = tpch.q6(); n = tpch.LINEITEM_ROWS[sys.argv[2]]
    t = rt.HipTable(1, tpch.chunk_rows(n)); d = tpch.gen_lineitem(n, tpch.SCALE[sys.argv[2]], q.columns)
    for c in q.columns:
        fid, dt = tpch.LINEITEM_SCHEMA[c]; t.append_column(fid, dt, d[c])
    pq = rt.PreparedQuery(t, q.predicate, q.aggs, q.keys, q.order_by_keys)
    for _ in range(20): pq.run()
    pq.set_profiling(True)
    for _ in range(200): pq.run()
    ms, k, _ = pq.kernel_time()
    pq.set_profiling(False)
    t0 = time.perf_counter()
    for _ in range(400): pq.run()
    step = (time.perf_counter() - t0) / 400
    print(json.dumps({"kernel_us": ms / k * 1e3, "step_us": step * 1e6, "value": pq.rows()[0].values[0].value}))
else:
    sf = sys.argv[1] if len(sys.argv) > 1 else "sf1"
    grids = sys.argv[2].split(",") if len(sys.argv) > 2 else ("0", "128", "192", "256", "384", "512", "768", "1024")
    for g in grids:
        env = dict(os.environ)
        if g != "0": env["LLKV_HIP_SCAN_WGS"] = g
        else: env.pop("LLKV_HIP_SCAN_WGS", None)
        r = subprocess.run([sys.executable, __file__, "one", sf], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(sf, "workgroups", g, line[-1] if line else r.stderr[-300:], flush=True)
        if not line: sys.exit(1)  # a faulting setting ends the sweep
