#!/usr/bin/env bash
# LDS / wait counters of the shared-image GROUP BY kernel (tools/groupby_bench.py by_shipdate), one rocprofv3 --pmc pass per
# counter group → gpurun_out/r02/pmc_image.txt
set -uo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/r02"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$OUT/pmc_image.txt"
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_INSTS_VALU"; do
  rm -rf /tmp/pmc_img
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_img -o p -- python3 "$ROOT/tools/groupby_bench.py" sf10 image_only > /dev/null 2>&1
  f="$(find /tmp/pmc_img -name '*counter_collection.csv' | head -1)"
  [ -n "$f" ] && python3 - "$f" >> "$OUT/pmc_image.txt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "llkv_jit_a" in r["Kernel_Name"] or "fused_scan" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(k, c, "launches", len(v), "avg", sum(v) / len(v))
PY
done
cat "$OUT/pmc_image.txt"
