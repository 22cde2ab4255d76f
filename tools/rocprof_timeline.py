#!/usr/bin/env python3
"""Kernel timeline of the LAST iteration recorded in a rocprofv3 results database (…_results.db):
start offset, duration (µs), name.  usage: rocprof_timeline.py results.db [first-kernel-substring] [n]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
first = sys.argv[2] if len(sys.argv) > 2 else None
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
if first:
    idx = [i for i, r in enumerate(rows) if first in r[0]]
    starts = [i for k, i in enumerate(idx) if k == 0 or idx[k - 1] != i - 1]
    rows = rows[starts[-1]:]
else:
    rows = rows[-n:]
t0 = rows[0][1]
busy = 0
for name, s, e in rows:
    busy += e - s
    print(f"{(s - t0) / 1000:9.1f} {(e - s) / 1000:8.1f}  {name[:100]}")
print(f"span {(rows[-1][2] - t0) / 1000:.1f} us, busy {busy / 1000:.1f} us, {len(rows)} kernels")
