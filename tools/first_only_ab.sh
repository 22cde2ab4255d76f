#!/usr/bin/env bash
# A/B of LLKV_FIRST_ONLY_MODE (fused_scan.hip.h) on ONE box: the 12-aggregate wide state, Q1 over DECIMAL(15,2) columns, and the
# Float64 Q1 in first-appearance order (its first-row lane).  Run-time compiled kernels (LLKV_HIP_FORCE_JIT=1), a private cache per mode.
set -euo pipefail
ROUND="${1:-r04}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/$ROUND"; mkdir -p "$OUT"
cd "$ROOT"
export LLKV_HIP_FORCE_JIT=1 LLKV_HIP_NO_JIT_SEED=1
for mode in 0 1 2; do
  export LLKV_HIP_JIT_DEFINES="-DLLKV_FIRST_ONLY_MODE=$mode" LLKV_HIP_CACHE_DIR="/tmp/jit_first_only_$mode"
  mkdir -p "$LLKV_HIP_CACHE_DIR"; chmod 700 "$LLKV_HIP_CACHE_DIR"
  echo "== LLKV_FIRST_ONLY_MODE=$mode"
  python3 tools/groupby_bench.py sf10 q1_wide_state 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); v=d['q1_wide_state']; print(f\"  q1_wide_state (12 aggregates)      kernel {v['kernel_ms']*1e3:8.1f} us  frac {v['frac_of_8TBs']:.3f}\")"
  python3 - <<'PY'
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
abi = importlib.import_module("rust-llkv_amd.abi"); rt = importlib.import_module("rust-llkv_amd.runtime"); tpch = importlib.import_module("rust-llkv_amd.tpch")
rt.init(0)
n = tpch.LINEITEM_ROWS["sf10"]; q = tpch.q1()
base = tpch.gen_lineitem(n, 10.0, q.columns)
for label, decimal, ordered in (("q1 decimal, key order", True, True), ("q1 f64, first-appearance order", False, False), ("q1 f64, key order", False, True)):
    d = tpch.lineitem_as_decimal(base) if decimal else base
    t = rt.HipTable(1, tpch.chunk_rows(n))
    for c in q.columns:
        fid, dt = tpch.LINEITEM_SCHEMA[c][0], tpch.lineitem_dtype(c, decimal)
        if dt == abi.DT_DECIMAL128: t.append_decimal128_column(fid, 15, 2, d[c])
        elif dt == abi.DT_UTF8: t.append_utf8_column(fid, d[c])
        else: t.append_column(fid, dt, d[c])
    pq = rt.PreparedQuery(t, q.predicate, q.aggs, q.keys, ordered)
    pq.set_profiling(True)
    for _ in range(12):
        pq.launch(0); pq.finish_only()
    ms, k, _ = pq.kernel_time()
    print(f"  {label:34s} kernel {ms / k * 1e3:8.1f} us  frac {pq.algorithmic_bytes / (ms / k) / 1e6 / 8000:.3f}")
    pq.close(); t.close()
PY
done | tee "$OUT/first_only_lanes.txt"
