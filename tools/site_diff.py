#!/usr/bin/env python3
"""Per-call-site comparison of two `bench.py --verbose` logs: tools/site_diff.py base.log new.log"""
import re
import sys


def parse(f):
    d = {}
    for l in open(f):
        m = re.match(r"\s+(.+?)\s+L(\d+)\s+calls=\s*(\d+) avg=\s*([\d.]+) ms\s+([\d.]+) GB/s", l)
        if m:
            d[(m.group(1), int(m.group(2)))] = (float(m.group(4)), float(m.group(5)), int(m.group(3)))
    return d


a, b = parse(sys.argv[1]), parse(sys.argv[2])
ncyc = min(v[2] for v in a.values())
ta = tb = 0.0
cls = {}
for k in sorted(a, key=lambda k: (k[0], k[1])):
    if k in b:
        per = a[k][2] / ncyc
        ta += a[k][0] * per; tb += b[k][0] * per
        c = cls.setdefault(k[0], [0.0, 0.0]); c[0] += a[k][0] * per; c[1] += b[k][0] * per
        print(f"{k[0]:28s} L{k[1]:<3d} {a[k][0]:.3f} -> {b[k][0]:.3f} ms   {a[k][1]:.0f} -> {b[k][1]:.0f} GB/s  ({(b[k][1] / a[k][1] - 1) * 100:+.1f}%)")
print()
for k, (x, y) in cls.items():
    print(f"{k:28s} {x:7.3f} -> {y:7.3f} ms per cycle ({(y / x - 1) * 100:+.1f}%)")
print(f"{'all sites':28s} {ta:7.3f} -> {tb:7.3f} ms per cycle ({(tb / ta - 1) * 100:+.1f}%)")
