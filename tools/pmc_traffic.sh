#!/usr/bin/env bash
# HBM traffic of the Q1 / Q6 scan kernels from rocprofv3 PMC passes (one counter per pass, no tracing beside
# --kernel-trace), summarised into profiles/<round>/pmc_traffic.json.  Run on the GPU box:
#   tools/pmc_traffic.sh r01
set -euo pipefail
ROUND="${1:-r01}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for w in q1_sf10 q6_sf10 q1_sf10_decimal; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc "$c" --kernel-trace --output-format csv -d "$OUT/${w}_$c" -o pmc -- \
      python3 "$ROOT/bench.py" --workload "$w" --steps 10 --warmup 2 --also "" --no-cpu-baseline > "$OUT/${w}_$c.log" 2>&1
  done
done
python3 - "$ROOT" "$ROUND" "$OUT" <<'PY'
import csv, glob, json, os, sys
root, rnd, out = sys.argv[1:4]
alg = {"q1_sf10": 59986052 * 38, "q6_sf10": 59986052 * 28, "q1_sf10_decimal": 59986052 * 38}
doc = {}
for w in ("q1_sf10", "q6_sf10", "q1_sf10_decimal"):
    avg = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        path = glob.glob(os.path.join(out, f"{w}_{c}", "**", "*counter_collection.csv"), recursive=True)[0]
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
                if r["Counter_Name"] == c and "fused_scan_kernel" in r["Kernel_Name"]]
        vals = vals[2:]  # warm-up launches
        avg[c] = sum(vals) / len(vals)
        n = len(vals)
    traffic = 2 * avg["FETCH_SIZE"] * 1024 + avg["WRITE_SIZE"] * 1024
    doc[w] = {"launches": n, "FETCH_SIZE_KB_avg": avg["FETCH_SIZE"], "WRITE_SIZE_KB_avg": avg["WRITE_SIZE"],
              "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": alg[w], "ratio": traffic / alg[w],
              "correction": "traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE reads exactly half of a coalesced stream, "
                            "MI355X_MICROARCH.md §HBM; separate --pmc passes); 2-byte loads of the flag columns are uncalibrated"}
dst = os.path.join(root, "gpurun_out", "pmc_traffic.json")
json.dump(doc, open(dst, "w"), indent=1)
print(json.dumps({w: doc[w]["ratio"] for w in doc}))
PY
