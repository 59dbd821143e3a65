#!/usr/bin/env python3
"""rocprofv3 rocpd database -> per-kernel summary CSV (the view `top_kernels`; durations in microseconds).
usage: kernel_stats.py results.db > summary.csv"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("kernel,calls,total_us,avg_us,percent")
for n, c, t, a, p in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f'"{n}",{c},{t:.3f},{a:.3f},{p:.4f}')
