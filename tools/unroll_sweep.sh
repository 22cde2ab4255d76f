#!/usr/bin/env bash
# GROUP BY shapes of tools/groupby_bench.py under LLKV_HIP_UNROLL (steps of 512 / 2 048 rows a thread block keeps in flight): the
# narrow-row shared-image plans and the wide-state dense plan.  usage (GPU box): bash tools/unroll_sweep.sh r04
set -euo pipefail
ROUND="${1:-r04}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/$ROUND"; mkdir -p "$OUT"
cd "$ROOT"
CASES="by_shipdate,by_shipdate_count_only,by_flag_status_shipdate,q1_wide_state"
for u in default 1 4 8; do
  if [ "$u" = default ]; then unset LLKV_HIP_UNROLL; else export LLKV_HIP_UNROLL=$u; fi
  echo "== LLKV_HIP_UNROLL=$u"
  python3 tools/groupby_bench.py sf10 "$CASES" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items():
    if isinstance(v,dict) and 'kernel_ms' in v: print(f\"  {k:28s} kernel {v['kernel_ms']*1e3:8.1f} us  frac {v['frac_of_8TBs']:.3f}  end-to-end {v['seconds_best']*1e3:.3f} ms\")
"
done | tee "$OUT/unroll_sweep.txt"
