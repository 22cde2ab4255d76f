#!/usr/bin/env python3
"""A/B on ONE device: side-stream overlap on/off, interleaved rounds (methodology rule 24)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rnd in range(3):
    for mode in ("1", "0"):
        for wl in ("q1_sf10", "q6_sf10", "q6_sf1"):
            e = dict(os.environ); e["LLKV_HIP_SIDE_STREAM"] = mode
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "10", "--no-cpu-baseline", "--also", "", "--workload", wl],
                                 env=e, capture_output=True, text=True, timeout=300)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print("ERR", out.stderr[-300:]); continue
            j = json.loads(line[-1])
            print(f"round {rnd} overlap={mode} {wl}: step {j['ms_per_step']*1e3:.1f} us kernel {j['roofline']['kernel_ms']*1e3:.1f} us value {j['value']/1e9:.1f} Grows/s", flush=True)
