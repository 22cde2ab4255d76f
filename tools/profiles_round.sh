#!/usr/bin/env bash
# Collects the judged measurements of a round into gpurun_out/<round>/profiles/ (copy them to profiles/<round>/ afterwards):
#   tools/profiles_round.sh r02 [part]     part: a = bench line + its kernel stats + PMC traffic, b = Q3 / GROUP BY / scan / join
# rocprofv3 is given the python interpreter directly (no wrapper that re-execs).
set -uo pipefail
ROUND="${1:-r02}"; PART="${2:-ab}"
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/$ROUND/profiles"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
stats() { # name, command…  → $OUT/<name>_kernel_stats.csv
  local name="$1"; shift
  rm -rf "/tmp/prof_$name"
  rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/prof_$name" -o p -- "$@" > "$OUT/${name}_rocprof.log" 2>&1 || echo "rocprofv3 $name failed"
  local f; f="$(find "/tmp/prof_$name" -name '*kernel_stats.csv' | head -1)"
  [ -n "$f" ] && cp "$f" "$OUT/${name}_kernel_stats.csv"
  echo "[$name] done"
}
if [[ "$PART" == *a* ]]; then
  python3 "$ROOT/bench.py" > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"; echo "[bench] rc=$?"
  stats bench python3 "$ROOT/bench.py" --no-cpu-baseline
  bash "$ROOT/tools/pmc_traffic.sh" "$ROUND" > "$OUT/pmc_traffic.log" 2>&1 && cp "$ROOT/gpurun_out/pmc_traffic.json" "$OUT/pmc_traffic.json"
  for f in "$ROOT"/gpurun_out/pmc/q1_sf10_FETCH_SIZE "$ROOT"/gpurun_out/pmc/q1_sf10_WRITE_SIZE; do
    c="$(find "$f" -name '*counter_collection.csv' | head -1)"; [ -n "$c" ] && cp "$c" "$OUT/pmc_$(basename "$f")_counter_collection.csv"
  done
  echo "[pmc] done"
  ROUND="$ROUND" bash "$ROOT/tools/bench_timeline.sh" "$ROUND" > "$OUT/bench_timeline.log" 2>&1 && cp "$ROOT/gpurun_out/$ROUND/bench_timeline.txt" "$OUT/bench_timeline.txt"; echo "[timeline] rc=$?"
  bash "$ROOT/tools/steps_overhead.sh" > "$OUT/steps_overhead.txt" 2>&1; echo "[steps] rc=$?"
fi
if [[ "$PART" == *b* ]]; then
  python3 "$ROOT/tools/q3_bench.py" sf10 --general > "$OUT/q3_bench.json" 2>/dev/null; echo "[q3] rc=$?"
  stats q3 python3 "$ROOT/tools/q3_bench.py" sf10
  # one iteration of the pipeline kernel by kernel (start offset, duration): the busy time against the host's time
  rm -rf /tmp/prof_q3db; rocprofv3 --kernel-trace -d /tmp/prof_q3db -o q3 -- python3 "$ROOT/tools/q3_bench.py" sf10 > "$OUT/q3_timeline_rocprof.log" 2>&1
  db="$(find /tmp/prof_q3db -name '*_results.db' | head -1)"; [ -n "$db" ] && python3 "$ROOT/tools/rocprof_timeline.py" "$db" hj_fill_zero_ranges_kernel > "$OUT/q3_timeline.txt"
  bash "$ROOT/tools/pmc_q3.sh" > "$OUT/pmc_q3.log" 2>&1 && cp "$ROOT/gpurun_out/pmc_q3.json" "$OUT/pmc_q3.json"; echo "[pmc q3] rc=$?"
  python3 "$ROOT/tools/groupby_bench.py" sf10 > "$OUT/groupby_bench.txt" 2>/dev/null; echo "[groupby] rc=$?"
  stats groupby python3 "$ROOT/tools/groupby_bench.py" sf10
  python3 "$ROOT/tools/scan_bench.py" > "$OUT/scan_bench.json" 2>/dev/null; echo "[scan] rc=$?"
  python3 "$ROOT/tools/join_bench.py" > "$OUT/join_bench.json" 2>/dev/null; echo "[join] rc=$?"
fi
ls -la "$OUT"
