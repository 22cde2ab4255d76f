#!/usr/bin/env bash
# Wait / LDS counters of the partitioned GROUP BY's scatter (llkv_jit_a) and reduce kernels (tools/groupby_bench.py sf10 by_partkey),
# one rocprofv3 --pmc pass per counter group → gpurun_out/r03/pmc_part_sq.txt.  Run on the GPU box.
set -uo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/r03"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$OUT/pmc_part_sq.txt"
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  rm -rf /tmp/pmc_psq
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_psq -o p -- python3 "$ROOT/tools/groupby_bench.py" sf10 by_partkey > /dev/null 2>&1
  f="$(find /tmp/pmc_psq -name '*counter_collection.csv' | head -1)"
  if [ -n "$f" ]; then python3 - "$f" >> "$OUT/pmc_part_sq.txt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "llkv_jit_a" in r["Kernel_Name"] or "part_reduce" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:32], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(k, c, "launches", len(v), "avg", sum(v) / len(v))
PY
  else echo "no counters for: $c" >> "$OUT/pmc_part_sq.txt"; fi
done
cat "$OUT/pmc_part_sq.txt"
