#!/usr/bin/env python3
"""BASELINE config 2 (2D advection 1000 x 1000 = 1e6 DoF, docs/gpus.md:34-38 recipe) on one MI355X: PCApply time with and
without the cycle captured in a hipGraph (launch-bound: ~35 levels of kernels of a few microseconds each), and the 3D 96^3 /
128^3 cases for the crossover.   usage: python tools/config2_timing.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pflare_amd as P

P.lib().airhip_set_device(0)
RECIPE = dict(coarsest_inverse_type="newton", coarsest_matrix_free_polys=True, coarsest_poly_order=10)
CASES = [("2D 1000^2 (config 2)", dict(dim=2, M=1000, N=1000), dict(**RECIPE)),
         ("2D 1000^2 truncated", dict(dim=2, M=1000, N=1000), dict(auto_truncate_start_level=1, auto_truncate_tol=1e-1, **RECIPE)),
         ("3D 96^3 bench recipe", dict(dim=3, M=96, N=96, P=96), dict(a_lump=True, auto_truncate_start_level=1, auto_truncate_tol=1e-1, **RECIPE)),
         ("3D 160^3 bench recipe", dict(dim=3, M=160, N=160, P=160), dict(a_lump=True, auto_truncate_start_level=1, auto_truncate_tol=1e-1, **RECIPE)),
         ("3D 256^3 bench recipe", dict(dim=3, M=256, N=256, P=256), dict(a_lump=True, auto_truncate_start_level=1, auto_truncate_tol=1e-1, **RECIPE))]
for name, gen, opts in CASES:
    A = P.HostMat.adv_diff_fd(**gen)
    n = A.shape[0]
    pc = P.PCAIR(**opts).setup(A)
    if os.environ.get("WAIT_MIRROR", "1") == "1":
        pc.host_mirror_wait()
    x = torch.ones(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    out = []
    for g in (False, True):
        pc.use_graph(g)
        for _ in range(3):
            pc.apply(x, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            pc.apply(x, y)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 20 * 1e3)
    b = torch.zeros(n, dtype=torch.float64, device="cuda")
    its = []
    for g in (False, True):
        pc.use_graph(g)
        sol = torch.ones_like(b)
        r = P.KSP("gmres", pc_side="right", rtol=1e-10, atol=1e-50, max_it=60).solve(b, sol, pc=pc)
        its.append((r["its"], round(r["solve_ms"], 2)))
    print(f"{name}: n={n} levels={pc.num_levels} PCApply {out[0]:.3f} ms plain, {out[1]:.3f} ms in a graph; GMRES (its, ms) plain {its[0]}, graph {its[1]}", flush=True)
    del pc
