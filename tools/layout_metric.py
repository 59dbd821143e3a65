#!/usr/bin/env python3
"""Host-side model of the x-gather locality of the row-block SpMV under different row/column orderings of A_ff.

For every level of a PCAIR hierarchy (built on the host) the F points are renumbered by (a) their natural order,
(b) a BxBxB brick key of the grid point they descend from (what bench.py passes as layout hint), (c) reverse
Cuthill-McKee of A_ff + A_ff^T (needs no geometry).  The operator is then cut into 512-non-zero row blocks as the
device format does, and for each block the columns are sorted: reported are the distinct 128-byte lines of x a
block touches (HBM/L2 traffic of the gather) and the mean number of lines per 64-lane gather instruction (what the
measurements of profiles/r01_spmv_experiments.md found the kernel time to follow).

usage: layout_metric.py [M=64] [brick=16]"""
import sys

import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import pflare_amd as P  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 64
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
BLK = 512


def block_metric(A):
    """A: CSR with rows/cols already in the order under test"""
    A = A.tocsr(); A.sort_indices()
    nnz = A.nnz
    if nnz == 0:
        return 0.0, 0.0, 0.0
    # row blocks of ~512 non-zeros (whole rows)
    starts = [0]
    acc = 0
    for r in range(A.shape[0]):
        rn = A.indptr[r + 1] - A.indptr[r]
        if acc and acc + rn > BLK:
            starts.append(r); acc = 0
        acc += rn
    starts.append(A.shape[0])
    lines_blk = []; lines_gather = []; dist = []
    for b in range(len(starts) - 1):
        cols = np.sort(A.indices[A.indptr[starts[b]]:A.indptr[starts[b + 1]]])
        if cols.size == 0:
            continue
        dist.append(np.unique(cols).size)
        lines_blk.append(np.unique(cols // 16).size)
        for g in range(0, cols.size, 64):
            lines_gather.append(np.unique(cols[g:g + 64] // 16).size)
    return float(np.mean(lines_blk)), float(np.mean(lines_gather)), float(np.sum(dist)) / max(1.0, float(np.sum(lines_blk)))


def main():
    hm = P.HostMat.adv_diff_fd(dim=3, M=M, N=M, P=M)
    pc = P.PCAIR(a_lump=True, auto_truncate_start_level=1, auto_truncate_tol=1e-1, coarsest_inverse_type="newton",
                 coarsest_matrix_free_polys=True, coarsest_poly_order=10).setup_host_only(hm)
    ax = np.arange(M)
    bi, li = ax // B, ax % B
    nb = (M + B - 1) // B
    key = (((bi[:, None, None] * nb + bi[None, :, None]) * nb + bi[None, None, :]) * B ** 3
           + (li[:, None, None] * B + li[None, :, None]) * B + li[None, None, :]).reshape(-1)
    print(f"{M}^3 advection, {pc.num_levels} levels; columns: lines per 512-nnz block | lines per 64-lane gather | distinct columns per line")
    print(f"{'level':>5} {'n_f':>9} {'nnz':>10} | {'natural':>22} | {'brick ' + str(B):>22} | {'RCM':>22}")
    for l in range(pc.num_levels - 1):
        e = pc.export_level(l)
        F = e["F"]; C = e["C"]
        kf = key[F]
        key = key[C]                       # coarse rows inherit the key of their parent C point
        if e.get("A_ff") is None:
            continue
        a = e["A_ff"]
        A = sp.csr_matrix((a.val, a.col, a.rowptr), shape=a.shape)
        if A.nnz < 20000:
            break
        out = []
        for name in ("natural", "brick", "rcm"):
            if name == "natural":
                p = np.arange(A.shape[0])
            elif name == "brick":
                p = np.argsort(kf, kind="stable")
            else:
                S = (A + A.T).tocsr()
                p = np.asarray(reverse_cuthill_mckee(S, symmetric_mode=True))
            Ap = A[p][:, p]
            out.append(block_metric(Ap))
        print(f"{l + 1:>5} {A.shape[0]:>9} {A.nnz:>10} | " + " | ".join(f"{x[0]:7.1f} {x[1]:6.2f} {x[2]:6.2f}" for x in out))


if __name__ == "__main__":
    main()
