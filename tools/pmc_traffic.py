#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into the per-launch HBM traffic of one
kernel and record it in profiles/pmc_traffic.json.

usage: pmc_traffic.py fetch_summary.csv write_summary.csv <kernel-substring> <key e.g. k_rb_spmv@464> <commit> [out.json]
(the summaries are what tools/pmc_run.sh writes: kernel,counter,launches,sum,mean_per_launch; all template
instantiations whose name contains the substring are added up)

Units / corrections (MI355X_MICROARCH.md, "HBM"): rocprofv3 reports FETCH_SIZE and WRITE_SIZE in units of 1024 B; on
gfx950 FETCH_SIZE tallies 128-B fabric requests at 64 B, so it is doubled; WRITE_SIZE is taken as is.  Both
corrections were checked on this kernel's own access widths with tools/pmc_calibrate.py (a banded operator of known
traffic): 2 x FETCH_SIZE x 1024 = 3.128e9 B against 3.120e9 B read, WRITE_SIZE x 1024 = 32.05e6 B against 32.00e6 B
written (profiles/r02_pmc_calibration.txt)."""
import csv
import json
import os
import sys


def total(path, counter, pat):
    n = 0; s = 0.0; names = []
    for row in csv.DictReader(open(path)):
        if row["counter"] == counter and pat in row["kernel"]:
            n += int(row["launches"]); s += float(row["sum"]); names.append(row["kernel"])
    return n, s, names


fetch_csv, write_csv, pat, key, commit = sys.argv[1:6]
nf, f, names = total(fetch_csv, "FETCH_SIZE", pat)
nw, w, _ = total(write_csv, "WRITE_SIZE", pat)
bytes_per_launch = 2.0 * f * 1024.0 / max(nf, 1) + w * 1024.0 / max(nw, 1)
out = sys.argv[6] if len(sys.argv) > 6 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
try:
    d = json.load(open(out))
except Exception:
    d = {}
d[key] = {"bytes_per_launch": bytes_per_launch, "fetch_size_raw_kb_per_launch": f / max(nf, 1),
          "write_size_raw_kb_per_launch": w / max(nw, 1), "launches": [nf, nw], "commit": commit, "kernels": sorted(set(names)),
          "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes of bench.py, tools/pmc_run.sh), "
                    "FETCH_SIZE doubled (gfx950), calibrated with tools/pmc_calibrate.py; taken at commit " + commit}
json.dump(d, open(out, "w"), indent=1, sort_keys=True)
print(key, d[key])
