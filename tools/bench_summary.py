#!/usr/bin/env python3
"""one line per bench log: tools/bench_summary.py log [log ...]"""
import json
import sys
for f in sys.argv[1:]:
    for l in open(f):
        if l.startswith('{"metric"'):
            d = json.loads(l); r = d["roofline"]; o = r.get("other_kernels", {}).get("k_rb_fc_first", {})
            print(f"{f:50s} {d['value']:7.2f} it/s  k_rb_spmv {r['achieved']:.0f} GB/s ({r['frac']:.4f})  fc_first {o.get('achieved', 0):.0f}  "
                  f"vcycle {r['vcycle_kernel_ms']:.2f} ms  setup {d['config']['setup_s']:.1f} s")
