#!/usr/bin/env bash
# HBM traffic of the Q3 pipeline's kernels (rocprofv3 PMC, one counter per pass) → gpurun_out/pmc_q3.json.  Run on the GPU box.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_q3"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc "$c" --kernel-trace --output-format csv -d "$OUT/$c" -o pmc -- python3 "$ROOT/tools/q3_bench.py" sf10 > "$OUT/$c.log" 2>&1
done
python3 - "$ROOT" "$OUT" <<'PY'
import csv, glob, json, os, sys
root, out = sys.argv[1:3]
doc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    path = glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # the pipeline of one query starts at hj_fill_zero_ranges_kernel: name the kernels by their position in it
    pipes, cur = [], None
    for r in rows:
        n = r["Kernel_Name"].split("(")[0].replace("llkv::", "")
        if n.startswith("hj_fill_zero_ranges"):
            cur = []; pipes.append(cur)
        if cur is not None: cur.append((n, float(r["Counter_Value"])))
    pipes = [p for p in pipes if len(p) == len(pipes[-1])][1:]  # drop the first (compiles, cold)
    for i in range(len(pipes[-1])):
        name = f"{i}:{pipes[-1][i][0]}"
        doc.setdefault(name, {})[c + "_KB_avg"] = sum(p[i][1] for p in pipes) / len(pipes)
for k, v in doc.items():
    v["traffic_bytes"] = 2 * v.get("FETCH_SIZE_KB_avg", 0) * 1024 + v.get("WRITE_SIZE_KB_avg", 0) * 1024
json.dump(doc, open(os.path.join(root, "gpurun_out", "pmc_q3.json"), "w"), indent=1)
for k, v in doc.items(): print(k, round(v["traffic_bytes"] / 1e6, 1), "MB")
PY
