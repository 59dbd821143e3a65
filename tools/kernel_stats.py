#!/usr/bin/env python3
"""rocprofv3 rocpd database -> per-kernel summary CSV (the view `top_kernels`; durations in microseconds), followed by
the same rows added up per kernel FUNCTION (template arguments stripped: `k_rb_spmv<false, 1>` and `<false, 9>` are
one function `k_rb_spmv`, which is the granularity of bench.py's roofline.kernel).
usage: kernel_stats.py results.db > summary.csv"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("kernel,calls,total_us,avg_us,percent")
agg = {}
for n, c, t, a, p in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f'"{n}",{c},{t:.3f},{a:.3f},{p:.4f}')
    m = re.search(r"(k_[a-z0-9_]+)", n)
    base = m.group(1) if m else n.split("(")[0]
    e = agg.setdefault(base, [0, 0.0, 0.0])
    e[0] += c; e[1] += t; e[2] += p
print("# per kernel function (all template instantiations added up)")
print("function,calls,total_us,avg_us,percent")
for base, (c, t, p) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'"{base}",{c},{t:.3f},{t / max(c, 1):.3f},{p:.4f}')
