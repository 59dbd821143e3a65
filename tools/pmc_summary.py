#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database: per kernel, per counter: launches, sum, mean per launch.
usage: pmc_summary.py results.db [kernel-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = f"%{sys.argv[2]}%" if len(sys.argv) > 2 else "%"
cur = db.cursor()
rows = cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                   "where kernel_name like ? group by kernel_name, counter_name order by kernel_name, counter_name", (pat,))
print("kernel,counter,launches,sum,mean_per_launch")
for k, c, n, s in rows:
    print(f'"{k[:90]}",{c},{n},{s:.6g},{s / n:.6g}')
