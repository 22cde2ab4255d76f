#!/usr/bin/env python3
"""Round 2 sweep: canonical tile length × workgroup count of the LDS-accumulator scan (Q1), on the whole SF10
table and on one 1/8 shard of it (rank 0 of 8 emulated on one device).  One subprocess per table (the knobs are read
at prepare time, so one staged table serves every configuration)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, json, os, sys
sys.path.insert(0, %r)
import torch
abi = importlib.import_module("rust-llkv_amd.abi"); rt = importlib.import_module("rust-llkv_amd.runtime"); tpch = importlib.import_module("rust-llkv_amd.tpch")
rt.init(0)
name, world = sys.argv[1], int(sys.argv[2]); steps = 40
qn, sf = name.split("_"); q = tpch.QUERIES[qn](); n = tpch.LINEITEM_ROWS[sf]
chunks = tpch.chunk_rows(n)
t = rt.HipTable(1, chunks, 0, world)
d = tpch.gen_lineitem(t.local_rows, tpch.SCALE[sf], q.columns)
full = tpch.gen_lineitem(n, tpch.SCALE[sf], [c for c in q.columns if tpch.LINEITEM_SCHEMA[c][1] not in (abi.DT_UTF8, abi.DT_FLOAT64)]) if world > 1 else None
for c in q.columns:
    fid, dt = tpch.LINEITEM_SCHEMA[c]
    if dt == abi.DT_UTF8:
        t.append_utf8_column(fid, d[c], ["A", "N", "R"] if c == "l_returnflag" else ["F", "O"])
    else:
        t.append_column(fid, dt, d[c])
        if world > 1 and t.local_column_stats(fid) is not None:
            t.set_column_stats(fid, int(full[c].min()), int(full[c].max()))
for tile in sys.argv[3].split(","):
  for unroll in os.environ.get("SWEEP_UNROLL", "auto").split(","):
    for tpw in sys.argv[4].split(","):
        os.environ["LLKV_HIP_TILE_ROWS"] = tile; os.environ["LLKV_HIP_SCAN_WGS"] = tpw
        if tpw == "auto": del os.environ["LLKV_HIP_SCAN_WGS"]
        if tile == "auto": del os.environ["LLKV_HIP_TILE_ROWS"]
        if unroll != "auto": os.environ["LLKV_HIP_UNROLL"] = unroll; os.environ["LLKV_HIP_FORCE_JIT"] = "1"
        pq = rt.PreparedQuery(t, q.predicate, q.aggs, q.keys, q.order_by_keys)
        for _ in range(5): pq.run()
        pq.set_profiling(True)
        for _ in range(steps): pq.run()
        ms, k, _ = pq.kernel_time()
        print(json.dumps({"workload": name, "world": world, "local_rows": t.local_rows, "tile_rows": tile, "unroll": unroll, "workgroups": tpw, "kernel_us": round(1e3 * ms / k, 2),
                          "gbs": round(pq.algorithmic_bytes / (ms / k) / 1e6, 1)}), flush=True)
        pq.close()
''' % ROOT
if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "q1_sf10"
    tiles = sys.argv[2] if len(sys.argv) > 2 else "auto,8192,32768"
    tpws = sys.argv[3] if len(sys.argv) > 3 else "auto,128,192,224,256,320,384,512,768,1024"
    for world in (1, 8):
        out = subprocess.run([sys.executable, "-c", CHILD, name, str(world), tiles, tpws], capture_output=True, text=True, timeout=900)
        print(out.stdout, end="", flush=True)
        if out.returncode:
            print("ERROR", out.stderr[-800:], flush=True)
