"""Reader for PETSc binary viewer files (the format of the reference's tests/data/*: MatLoad /
VecLoad in tests/ex6.c:36-44, tests/ex12f.F90:48-55).  Big-endian; a file is a sequence of objects:

  Mat (AIJ): int32 MAT_FILE_CLASSID=1211216, M, N, nnz, int32 rowlen[M], int32 col[nnz], float64 val[nnz]
  Vec      : int32 VEC_FILE_CLASSID=1211214, n, float64 val[n]
"""
from __future__ import annotations

import numpy as np

MAT_FILE_CLASSID = 1211216
VEC_FILE_CLASSID = 1211214


def read_petsc_binary(path):
    """Returns the list of objects in the file: ('mat', (rowptr int64, col int32, val float64, shape)) or
    ('vec', float64 array)."""
    raw = open(path, "rb").read()
    off, out = 0, []
    while off < len(raw):
        cid = int(np.frombuffer(raw, ">i4", 1, off)[0]); off += 4
        if cid == MAT_FILE_CLASSID:
            M, N, nz = (int(v) for v in np.frombuffer(raw, ">i4", 3, off)); off += 12
            if nz < 0:
                raise ValueError("dense / special Mat storage is not supported")
            rl = np.frombuffer(raw, ">i4", M, off).astype(np.int64); off += 4 * M
            col = np.frombuffer(raw, ">i4", nz, off).astype(np.int32); off += 4 * nz
            val = np.frombuffer(raw, ">f8", nz, off).astype(np.float64); off += 8 * nz
            rp = np.zeros(M + 1, np.int64); np.cumsum(rl, out=rp[1:])
            if rp[-1] != nz:
                raise ValueError("corrupt Mat: row lengths do not add up")
            # AIJ keeps columns ascending inside a row; sort defensively (stable)
            for i in range(M):
                a, b = rp[i], rp[i + 1]
                if b - a > 1 and np.any(np.diff(col[a:b]) < 0):
                    o = np.argsort(col[a:b], kind="stable"); col[a:b] = col[a:b][o]; val[a:b] = val[a:b][o]
            out.append(("mat", (rp, col, val, (M, N))))
        elif cid == VEC_FILE_CLASSID:
            n = int(np.frombuffer(raw, ">i4", 1, off)[0]); off += 4
            out.append(("vec", np.frombuffer(raw, ">f8", n, off).astype(np.float64))); off += 8 * n
        else:
            raise ValueError(f"unknown PETSc class id {cid} at byte {off - 4}")
    return out


def load_mat_vec(path):
    """(scipy CSR matrix, rhs or None): what tests/ex6.c / ex12f.F90 read from one file."""
    import scipy.sparse as sp
    objs = read_petsc_binary(path)
    rp, col, val, shape = next(o for k, o in objs if k == "mat")
    b = next((o for k, o in objs if k == "vec"), None)
    return sp.csr_matrix((val, col, rp), shape=shape), b
