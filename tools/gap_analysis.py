#!/usr/bin/env python3
"""Idle gaps between consecutive kernels of a rocprofv3 kernel trace (rocpd database).
usage: gap_analysis.py results.db [last_fraction]   - analyses the last fraction (default 0.25) of the dispatches"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
cols = [d[1] for d in db.execute("pragma table_info(kernels)")]
rows = list(db.execute("select name, start, end from kernels order by start"))
rows = rows[int(len(rows) * (1 - frac)):]
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
gaps = defaultdict(lambda: [0, 0])
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    g = s1 - e0
    if g > 0:
        k = (n0.split("(")[0][-40:], n1.split("(")[0][-40:])
        gaps[k][0] += 1; gaps[k][1] += g
print(f"dispatches {len(rows)} span {span/1e6:.2f} ms busy {busy/1e6:.2f} ms idle {100*(span-busy)/span:.1f}%")
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"  {t/1e6:8.3f} ms in {c:5d} gaps (avg {t/c/1e3:7.1f} us)  {k[0]} -> {k[1]}")
