#!/usr/bin/env python3
"""The reference's python/ex2.py (PETSc tutorial ex2: 2-D five-point Laplacian, exact solution u = 1, PCAIR chosen and
configured through the pflare option API) on this package: scipy assembles the operator, `pflare_amd` owns the
preconditioner and the Krylov solve on the GPU.

usage: python examples/ex2.py [-m 8] [-n 7]"""
import argparse
import os
import sys

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pflare_amd as P           # noqa: E402
from pflare_amd import pflare     # noqa: E402


def laplace_2d(m, n):
    ii, jj = np.meshgrid(np.arange(m), np.arange(n), indexing="ij")
    row = (ii * n + jj).ravel()
    rows, cols, vals = [row], [row], [np.full(row.size, 4.0)]
    for mask, off in ((ii > 0, -n), (ii < m - 1, n), (jj > 0, -1), (jj < n - 1, 1)):
        r = row[mask.ravel()]
        rows.append(r); cols.append(r + off); vals.append(np.full(r.size, -1.0))
    A = sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(m * n, m * n))
    A.sort_indices()
    return A


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-m", type=int, default=8)
    ap.add_argument("-n", type=int, default=7)
    args = ap.parse_args()
    import torch
    A = laplace_2d(args.m, args.n)
    u = np.ones(A.shape[0])
    b = A @ u
    # PCAIR through the reference's option API
    pc = P.PCAIR()
    pflare.pcair_set_z_type(pc, pflare.AIR_Z_PRODUCT)
    pflare.pcair_set_inverse_type(pc, pflare.PFLAREINV_POWER)
    pflare.pcair_set_a_drop(pc, 1e-3)
    pc.setup(A)
    ksp = P.KSP("gmres", pc_side="left", rtol=1e-7, max_it=50)
    bd = torch.from_numpy(b).cuda()
    xd = torch.zeros_like(bd)
    res = ksp.solve(bd, xd, pc=pc)
    err = float(np.linalg.norm(xd.cpu().numpy() - u))
    print(f"Norm of error {err:.3e} iterations {res['its']} levels {pflare.pcair_get_num_levels(pc)} "
          f"grid complexity {pflare.pcair_get_grid_complexity(pc):.3f}")
    return 0 if (res["reason"] > 0 and err < 1e-4) else 1


if __name__ == "__main__":
    sys.exit(main())
