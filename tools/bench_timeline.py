#!/usr/bin/env python3
"""Steady-state timeline of bench.py's timed loop from a rocprofv3 results database (--kernel-trace [--memory-copy-trace]):
the last N launches of the dominant kernel — start offset, duration, the gap to the previous launch and what ran in the gap
(other kernels, memory copies).  usage: bench_timeline.py results.db [kernel-substring] [n]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
name = sys.argv[2] if len(sys.argv) > 2 else "fused_scan_kernel"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
tables = {r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")}
kern = db.execute("select name,start,end from kernels order by start").fetchall()
copies = []
for t in ("memory_copies", "memory_copy"):
    if t in tables:
        try:
            copies = db.execute(f"select name,start,end from {t} order by start").fetchall()
        except sqlite3.Error:
            pass
        break
idx = [i for i, r in enumerate(kern) if name in r[0]]
# the main workload's launches: the longest run of launches with the same duration class — take the first `steps` after warm-up
# by looking for the densest block; simply: the first block of consecutive launches of the kernel whose median duration is largest
blocks, cur = [], []
for k, i in enumerate(idx):
    if cur and kern[i][1] - kern[cur[-1]][2] > 5_000_000:  # > 5 ms apart: another workload / phase
        blocks.append(cur); cur = []
    cur.append(i)
if cur:
    blocks.append(cur)
blocks = [b for b in blocks if len(b) >= n]
if not blocks:
    sys.exit("no block of %d launches of %s" % (n, name))
main = max(blocks, key=lambda b: sorted(kern[i][2] - kern[i][1] for i in b)[len(b) // 2])
sel = main[-n:]
t0 = kern[sel[0]][1]
gaps, durs = [], []
print(f"# last {n} launches of {name} in the block of {len(main)} (µs): start, duration, gap before, in the gap")
for a, i in enumerate(sel):
    s, e = kern[i][1], kern[i][2]
    prev_end = kern[sel[a - 1]][2] if a else None
    between = []
    if a:
        between = [f"{r[0][:40]}:{(r[2] - r[1]) / 1000:.1f}" for r in kern if prev_end <= r[1] < s and name not in r[0]]
        between += [f"copy {r[0][:24]}:{(r[2] - r[1]) / 1000:.1f}" for r in copies if prev_end <= r[1] < s]
        gaps.append((s - prev_end) / 1000)
    durs.append((e - s) / 1000)
    print(f"{(s - t0) / 1000:10.1f} {(e - s) / 1000:8.1f} {'' if not a else f'{gaps[-1]:7.1f}'}  {' '.join(between)}")
span = (kern[sel[-1]][2] - t0) / 1000
print(f"# span {span:.1f} µs for {n} launches = {span / n:.2f} µs per step; kernel avg {sum(durs) / n:.2f} µs; gap avg {sum(gaps) / max(1, len(gaps)):.2f} µs (min {min(gaps):.1f}, max {max(gaps):.1f})")
