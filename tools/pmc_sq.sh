#!/usr/bin/env bash
# Where the waves of the Q3 pipeline's kernels spend their cycles: SQ counters per kernel (rocprofv3 --pmc, its own run).
# WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ≈ WAVE_CYCLES (MI355X_MICROARCH.md).
# usage (GPU box): bash tools/pmc_sq.sh [extra env assignments…]   → gpurun_out/pmc_sq.txt
#   PMC_MEM=1: the memory pipeline's counters; PMC_SETS="A B;C D": these sets; PMC_CMD="script.py args" PMC_BY_NAME=1: another workload, per kernel name
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_sq"
rm -rf "$OUT"; mkdir -p "$OUT"
for sw in "$@"; do export "$sw"; done
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR")
# PMC_MEM=1: the memory pipeline instead — texture addresser, L1 (TCP), L2 (TCC) and its fabric side
if [ -n "${PMC_MEM:-}" ]; then
  # (one or two counters per block and pass: four TA counters at once "exceed the capabilities of the hardware" and rocprofv3 aborts)
  SETS=("TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
        "TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_32B_sum" "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_avr TCP_READ_TAGCONFLICT_STALL_CYCLES_sum")
fi
if [ -n "${PMC_SETS:-}" ]; then IFS=';' read -r -a SETS <<< "$PMC_SETS"; fi # e.g. PMC_SETS="TA_BUSY_avr GRBM_GUI_ACTIVE;SQ_WAIT_ANY SQ_WAVE_CYCLES"
for set in "${SETS[@]}"; do
  tag="$(echo "$set" | tr ' ' '_')"
  echo "[pmc] $set"
  timeout -k 5 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/$tag" -o pmc -- python3 ${PMC_CMD:-"$ROOT/tools/q3_bench.py" sf10} > "$OUT/$tag.log" 2>&1 || echo "pass $tag failed"
done
python3 - "$OUT" <<'PY' | tee "$ROOT/gpurun_out/pmc_sq.txt"
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # name kernels by position in the pipeline that starts at hj_fill_zero_ranges_kernel; skip the first (cold) pipeline
    pipes, cur = [], None
    by_dispatch = collections.OrderedDict()
    for r in rows: by_dispatch.setdefault(int(r["Dispatch_Id"]), []).append(r)
    for d, rs in by_dispatch.items():
        n = rs[0]["Kernel_Name"].split("(")[0].replace("llkv::", "")
        if n.startswith("hj_fill_zero_ranges"): cur = []; pipes.append(cur)
        if cur is not None: cur.append((n, {r["Counter_Name"]: float(r["Counter_Value"]) for r in rs}))
    if os.environ.get("PMC_BY_NAME"):  # any command: average per kernel name (the first dispatch of a name is dropped: cold)
        seen = set()
        for d, rs in by_dispatch.items():
            n = rs[0]["Kernel_Name"].split("(")[0].replace("llkv::", "")[:60]
            if n not in seen: seen.add(n); continue
            for r in rs: acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        continue
    if not pipes: continue
    pipes = [p for p in pipes if len(p) == len(pipes[-1])][1:]
    for p in pipes:
        for i, (n, cs) in enumerate(p):
            for c, v in cs.items(): acc[f"{i}:{n}"][c].append(v)
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:24s} {sum(v)/len(v):16.0f}")
PY
