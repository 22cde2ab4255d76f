#!/usr/bin/env python3
"""Kernel tuning sweep on one GPU: tile rows × unroll × load policy for Q1 / Q6 at SF10.
Each configuration runs in a fresh subprocess (the knobs are read at plan-lowering / JIT time)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, json, os, sys, time
sys.path.insert(0, %r)
import torch
abi = importlib.import_module("rust-llkv_amd.abi"); rt = importlib.import_module("rust-llkv_amd.runtime"); tpch = importlib.import_module("rust-llkv_amd.tpch")
rt.init(0)
name = sys.argv[1]; steps = 30
qn, sf = name.split("_"); q = tpch.QUERIES[qn](); n = tpch.LINEITEM_ROWS[sf]
t = rt.HipTable(1, tpch.chunk_rows(n)); d = tpch.gen_lineitem(n, tpch.SCALE[sf], q.columns)
for c in q.columns:
    fid, dt = tpch.LINEITEM_SCHEMA[c]
    t.append_utf8_column(fid, d[c]) if dt == abi.DT_UTF8 else t.append_column(fid, dt, d[c])
pq = rt.PreparedQuery(t, q.predicate, q.aggs, q.keys, q.order_by_keys)
for _ in range(5): pq.run()
pq.set_profiling(True)
for _ in range(steps): pq.run()
ms, k, _ = pq.kernel_time()
print(json.dumps({"kernel_ms": ms / k, "gbs": pq.algorithmic_bytes / (ms / k) / 1e6}))
''' % ROOT
def run(name, env):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, "-c", CHILD, name], env=e, capture_output=True, text=True, timeout=300)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return json.loads(line[-1]) if line else {"error": (out.stderr or out.stdout)[-400:]}
if __name__ == "__main__":
    names = sys.argv[1:] or ["q1_sf10", "q6_sf10"]
    for name in names:
        for nt in ("0", "1"):
            for u in ("2", "4", "8"):
                for tile in ("4096", "8192", "16384", "32768", "65536"):
                    env = {"LLKV_HIP_FORCE_JIT": "1", "LLKV_HIP_UNROLL": u, "LLKV_HIP_TILE_ROWS": tile,
                           "LLKV_HIP_JIT_DEFINES": f"-DLLKV_NT_LOADS={nt}"}
                    r = run(name, env)
                    print(name, "nt", nt, "U", u, "tile", tile, r, flush=True)
