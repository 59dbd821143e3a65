#!/usr/bin/env python3
"""SpMV structure lab (GPU box): achieved algorithmic GB/s of airhip_mat_mult on synthetic operators whose
row length and column locality are controlled.  Separates what the kernel costs per row / per non-zero from what
the x-gather locality of the real AIR operators costs.
usage: python tools/spmv_lab.py [nnz_millions]"""
import sys
import time
import os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def csr_banded(m, L, spread=1, jitter=0, seed=0):
    """row i: L columns i + spread*(0..L-1) - centred (rows near the ends are left out); jitter replaces that
    fraction of the entries by random far columns"""
    rng = np.random.default_rng(seed)
    lo = spread * (L // 2 + 1); hi = m - spread * (L + 1)
    rows = np.arange(lo, hi, dtype=np.int32)
    cols = rows[:, None] + (spread * (np.arange(L, dtype=np.int32) - L // 2))[None, :]
    if jitter > 0:
        far = rng.random(cols.shape) < jitter
        cols = np.where(far, rng.integers(0, m, size=cols.shape, dtype=np.int32), cols)
        cols.sort(axis=1)
        ok = (np.diff(cols, axis=1) > 0).all(axis=1) if L > 1 else np.ones(cols.shape[0], dtype=bool)
        cols = cols[ok]
    mm = cols.shape[0]
    rp = np.arange(mm + 1, dtype=np.int64) * L
    vals = rng.standard_normal(mm * L)
    return (rp, np.ascontiguousarray(cols.reshape(-1), dtype=np.int32), vals, (mm, m))


def run(name, A, reps=20):
    import torch
    import pflare_amd as P
    m, n = A[3]
    nnz = int(A[0][-1])
    dm = P.DeviceMat(A)
    x = torch.randn(n, dtype=torch.float64, device="cuda")
    y = torch.empty(m, dtype=torch.float64, device="cuda")
    for _ in range(3):
        dm.mult(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        dm.mult(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    alg = 12.0 * nnz + 4.0 * (m + 1) + 8.0 * m + 8.0 * n
    print(f"{name:44s} m={m:9d} nnz={nnz:10d}  {dt*1e3:7.3f} ms  {alg/dt/1e9:7.1f} GB/s alg  ({12.0*nnz/dt/1e9:7.1f} stream only)", flush=True)
    del dm


if __name__ == "__main__":
    import pflare_amd as P
    nnz_m = float(sys.argv[1]) if len(sys.argv) > 1 else 64.0
    if len(sys.argv) > 2:      # probe mode: the diagnostic kernels on a few structures
        for probe in (0, 1, 2):
            P.lib().airhip_set_tuning(b"spmv_probe", probe)
            for L in (4, 16, 64, 512):
                run(f"probe={probe} banded L={L}", csr_banded(int(nnz_m * 1e6 / L), L))
            run(f"probe={probe} L=16 10% random far", csr_banded(int(nnz_m * 1e6 / 16), 16, jitter=0.1))
        P.lib().airhip_set_tuning(b"spmv_probe", 0)
        sys.exit(0)
    for L in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
        m = int(nnz_m * 1e6 / L)
        run(f"banded L={L} (perfect locality)", csr_banded(m, L))
    for L in (4, 16, 32):
        m = int(nnz_m * 1e6 / L)
        run(f"L={L} spread=1000 (strided far columns)", csr_banded(m, L, spread=1000))
        run(f"L={L} 10% random far columns", csr_banded(m, L, jitter=0.1))
        run(f"L={L} 50% random far columns", csr_banded(m, L, jitter=0.5))
