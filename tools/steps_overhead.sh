#!/usr/bin/env bash
# ms per step of the default bench workload at 20 and at 200 timed steps: what a step costs beyond its kernel in steady state, and what
# the start and the drain of the timed region add when the region is short.  usage (GPU box): bash tools/steps_overhead.sh
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
for k in 20 20 200 200; do
  python3 bench.py --steps $k --warmup 5 --no-cpu-baseline --also "" 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('steps', d['steps'], 'ms_per_step', round(d['ms_per_step'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'over the kernel (us)', round((d['ms_per_step']-d['roofline']['kernel_ms'])*1e3,1))"
done
