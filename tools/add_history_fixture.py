#!/usr/bin/env python3
"""Adds the residual history of a single-GPU `bench.py --verbose` run to tests/golden/bench_history_n1.json (the fixture
the N > 1 bench lines compare their own history with: config.history_check).

usage: python tools/add_history_fixture.py <size> <recipe> <stderr log of bench.py --verbose> [source note]"""
import json
import os
import sys

size, recipe, log = sys.argv[1], sys.argv[2], sys.argv[3]
note = sys.argv[4] if len(sys.argv) > 4 else ""
line = [l for l in open(log) if l.startswith("[bench] residual history of the converged solve: ")][-1]
hist = json.loads(line.split(": ", 1)[1])
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "bench_history_n1.json")
try:
    fx = json.load(open(path))
except (OSError, ValueError):
    fx = {}
fx[size] = {"recipe": recipe, "iterations": len(hist) - 1, "history": hist,
            "source": note or f"bench.py --verbose --size {size} on one MI355X: residual norms of the converged right-preconditioned GMRES(30) solve, b = 0, x0 = 1"}
json.dump(fx, open(path, "w"), indent=1)
print(f"{size}: {len(hist) - 1} iterations")
