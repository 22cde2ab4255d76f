#!/usr/bin/env bash
# Q3 on one GPU: the pipeline's tests, then per-kernel timelines and the bench under the switches given as arguments
# (each argument an environment assignment or "-" for none), all on one box.  Usage (on the GPU box): bash tools/q3_check.sh - LLKV_HIP_JOIN_NO_LATE=1
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/${ROUND:-r04}"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 700 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 -k "q3 or join_groupby or join_pipeline" > "$OUT/q3_tests.log" 2>&1 || { tail -20 "$OUT/q3_tests.log"; exit 1; }
tail -3 "$OUT/q3_tests.log"
for sw in "$@"; do
  [ "$sw" = "-" ] && sw="LLKV_NONE=1"
  echo "== $sw"
  export TMPDIR=/tmp
  rm -rf /tmp/prof_q3db
  export "$sw"   # rocprofv3 must start the interpreter itself: the switch travels in this shell's environment
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/prof_q3db -o q3 -- python3 "$ROOT/tools/q3_bench.py" sf10 > "$OUT/q3_tl.log" 2>&1)
  db="$(find /tmp/prof_q3db -name "*_results.db" | head -1)"
  python3 "$ROOT/tools/rocprof_timeline.py" "$db" hj_fill_zero_ranges_kernel | tee "$OUT/q3_timeline_${sw%%=*}.txt"
  python tools/q3_bench.py sf10 2>/dev/null | cut -c1-330
  unset "${sw%%=*}"
done
