#!/usr/bin/env bash
# Per-kernel timelines of Q3 (tools/q3_bench.py sf10) under the switches given as arguments — each an environment assignment, "-" for
# none — on one box, without the tests tools/q3_check.sh runs first.  Usage (GPU box): bash tools/q3_timelines.sh - "LLKV_HIP_JIT_DEFINES=-DX=1"
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/${ROUND:-r04}"
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
for sw in "$@"; do
  [ "$sw" = "-" ] && sw="LLKV_NONE=1"
  echo "== $sw"
  rm -rf /tmp/prof_q3db
  export "$sw"   # rocprofv3 must start the interpreter itself: the switch travels in this shell's environment
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof_q3db -o q3 -- python3 "$ROOT/tools/q3_bench.py" sf10 > "$OUT/q3_tl.log" 2>&1)
  db="$(find /tmp/prof_q3db -name "*_results.db" | head -1)"
  python3 "$ROOT/tools/rocprof_timeline.py" "$db" hj_fill_zero_ranges_kernel
  unset "${sw%%=*}"
done
