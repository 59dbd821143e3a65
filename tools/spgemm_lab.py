#!/usr/bin/env python3
"""SpGEMM lab (GPU box): the device SpGEMM of the setup on operands with the structure of the deep AIR levels, where
the 464^3 setup spends its time - Z-like products A_cf * M_ff and, as a stand-in for R * (A P), A_cf * (A_ff * A_ff)
(rows of B with a few hundred entries, 5-20 thousand candidates per output row).  Run with AIRHIP_SPGEMM_TRACE=1 for
the per-phase host timers, under rocprofv3 --kernel-trace --stats for the kernel times.
usage: python tools/spgemm_lab.py [grid size, default 112] [first level, default 5] [levels, default 3] [repeats]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def device_spgemm(P, A, B, drop, tol, lump, negate):
    lib = P.lib()
    fn = lib.airhip_spgemm_filtered
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    a, b = P.as_csr(A), P.as_csr(B)
    h = C.c_void_p()
    stats = (C.c_int64 * 5)()
    t0 = time.perf_counter()
    rc = fn(a.ref(), b.ref(), int(drop), float(tol), int(lump), int(negate), 1, C.byref(h), stats)
    dt = time.perf_counter() - t0
    assert rc == 0, lib.airhip_last_error().decode()
    out = P.HostMat(h)
    return out, [int(s) for s in stats], dt


def to_scipy(c):
    return sp.csr_matrix((c.val, c.col, c.rowptr), shape=c.shape)


if __name__ == "__main__":
    import pflare_amd as P
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 112
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    A = P.HostMat.adv_diff_fd(dim=3, M=size, N=size, P=size)
    pc = P.PCAIR(a_lump=True, coarsest_inverse_type="newton", coarsest_matrix_free_polys=True, coarsest_poly_order=10)
    t0 = time.perf_counter()
    pc.setup(A)
    print(f"[lab] {size}^3: {pc.num_levels} levels, setup {time.perf_counter() - t0:.1f} s", flush=True)
    if len(sys.argv) > 5:          # experiment switches of a lab build of the kernels (not in the committed library)
        os.environ["AIRHIP_SPGEMM_LAB"] = sys.argv[5]
    for lv in range(first, min(first + count, pc.num_levels - 1)):
        s = pc.export_level(lv, copy=True)
        if "A_ff" not in s or "mat" not in s["inv_ff"]:
            continue
        Acf, Aff, M = to_scipy(s["A_cf"]), to_scipy(s["A_ff"]), to_scipy(s["inv_ff"]["mat"])
        B2 = (Aff @ Aff).tocsr(); B2.sort_indices()
        for name, X, Y, kw in (("A_cf*M_ff   drop 0   negate", Acf, M, dict(drop=True, tol=0.0, lump=False, negate=True)),
                               ("A_cf*A_ff^2 drop 1e-4 lump ", Acf, B2, dict(drop=True, tol=1e-4, lump=True, negate=False))):
            cand = int((X @ sp.csr_matrix((np.ones(Y.nnz), Y.indices, Y.indptr), shape=Y.shape)).sum()) if X.nnz < 5e7 else -1
            for r in range(reps):
                out, stats, dt = device_spgemm(P, X, Y, **kw)
            print(f"[lab] level {lv} {name}: m={X.shape[0]} nnzA={X.nnz} ({X.nnz / X.shape[0]:.0f}/row) nnzB={Y.nnz} "
                  f"({Y.nnz / Y.shape[0]:.0f}/row) nnzC={out.nnz} bins={stats} call {dt * 1e3:.1f} ms", flush=True)
