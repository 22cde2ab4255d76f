#!/usr/bin/env bash
# HBM traffic of the partitioned GROUP BY's kernels (tools/groupby_bench.py sf10 by_partkey), one rocprofv3 --pmc pass per counter
# → gpurun_out/r03/pmc_part.txt.  Run on the GPU box.
set -uo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/r03"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$OUT/pmc_part.txt"
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf /tmp/pmc_part
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_part -o p -- python3 "$ROOT/tools/groupby_bench.py" sf10 by_partkey > /dev/null 2>&1
  f="$(find /tmp/pmc_part -name '*counter_collection.csv' | head -1)"
  [ -n "$f" ] && python3 - "$f" >> "$OUT/pmc_part.txt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "llkv_jit_a" in r["Kernel_Name"] or "part_reduce" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(k, c, "launches", len(v), "avg_KB", sum(v) / len(v), "bytes_corrected", (2 if c == "FETCH_SIZE" else 1) * 1024 * sum(v) / len(v))
PY
done
cat "$OUT/pmc_part.txt"
