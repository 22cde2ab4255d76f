#!/usr/bin/env bash
# SQ counters of the Q1 / Q6 scan kernels (bench.py, 10 steps), one rocprofv3 --pmc pass per counter pair → gpurun_out/r02/pmc_q1.txt
set -uo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/r02"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$OUT/pmc_q1.txt"
for w in q1_sf10 q6_sf10; do
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/pmc_q1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_q1 -o p -- python3 "$ROOT/bench.py" --workload $w --steps 10 --warmup 2 --also "" --no-cpu-baseline > /dev/null 2>&1
  f="$(find /tmp/pmc_q1 -name '*counter_collection.csv' | head -1)"
  [ -n "$f" ] && python3 - "$f" "$w" >> "$OUT/pmc_q1.txt" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "fused_scan_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(acc.items()):
    v = v[2:] or v
    print(sys.argv[2], c, "launches", len(v), "avg", sum(v) / len(v))
PY
done
done
cat "$OUT/pmc_q1.txt"
