#!/usr/bin/env bash
# rocprofv3 kernel + memory-copy trace of `bench.py --steps 40` (main workload only) and the steady-state timeline of its last 20
# launches → gpurun_out/<round>/bench_timeline.txt (+ the kernel stats csv).  usage (GPU box): bash tools/bench_timeline.sh r04 [bench args…]
set -euo pipefail
ROUND="${1:-r04}"; shift || true
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/$ROUND"
mkdir -p "$OUT"
export TMPDIR=/tmp
rm -rf /tmp/prof_bench_tl
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --stats -d /tmp/prof_bench_tl -o bench -- python3 "$ROOT/bench.py" --steps 40 --warmup 5 --no-cpu-baseline --also "" "$@" > "$OUT/bench_timeline_run.json" 2> "$OUT/bench_timeline_run.err")
db="$(find /tmp/prof_bench_tl -name '*_results.db' | head -1)"
python3 "$ROOT/tools/bench_timeline.py" "$db" fused_scan_kernel 20 | tee "$OUT/bench_timeline.txt"
f="$(find /tmp/prof_bench_tl -name '*kernel_stats.csv' | head -1)"; [ -n "$f" ] && cp "$f" "$OUT/bench_timeline_kernel_stats.csv" || true
