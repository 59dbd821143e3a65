#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into the per-launch HBM
traffic of one kernel and record it in profiles/pmc_traffic.json.

usage: pmc_traffic.py fetch.db write.db <kernel-substring> <key e.g. k_rb_spmv@464> [out.json]

Units / corrections (MI355X_MICROARCH.md, "HBM"): rocprofv3 reports FETCH_SIZE and WRITE_SIZE in
KiB-like units of 1024 B; on gfx950 FETCH_SIZE tallies 128-B fabric requests at 64 B, so it is
doubled; WRITE_SIZE is taken as is."""
import json
import os
import sqlite3
import sys


def per_launch(db, counter, pat):
    cur = sqlite3.connect(db).cursor()
    n, s = cur.execute("select count(*), sum(value) from counters_collection where counter_name=? and kernel_name like ?",
                       (counter, f"%{pat}%")).fetchone()
    return n, (s or 0.0) / max(n, 1)


fetch_db, write_db, pat, key = sys.argv[1:5]
nf, f = per_launch(fetch_db, "FETCH_SIZE", pat)
nw, w = per_launch(write_db, "WRITE_SIZE", pat)
bytes_per_launch = 2.0 * f * 1024.0 + w * 1024.0
out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
try:
    d = json.load(open(out))
except Exception:
    d = {}
d[key] = {"bytes_per_launch": bytes_per_launch, "fetch_size_raw_kb_per_launch": f, "write_size_raw_kb_per_launch": w,
          "launches": [nf, nw],
          "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes of bench.py), FETCH_SIZE doubled (gfx950)"}
json.dump(d, open(out, "w"), indent=1, sort_keys=True)
print(key, d[key])
