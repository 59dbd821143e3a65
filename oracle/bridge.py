"""TEST INFRASTRUCTURE ONLY: converts between the product's exported hierarchy
(pflare_amd.PCAIR.export_level / export_coarse) and the oracle's `air_ref.Hierarchy`, so the
oracle's CPU V-cycle can be run on exactly the operands the HIP path uses, and the other way
round (oracle-built hierarchy injected through the bring-your-own C ABI)."""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from . import air_ref as R


def to_scipy(c):
    return sp.csr_matrix((c.val, c.col, c.rowptr), shape=c.shape)


def _inv_from_dict(d, A):
    k = d["kind"]
    if k == 1:
        return R.InvOp("assembled", mat=to_scipy(d["mat"]))
    if k == 2:
        return R.InvOp("diag", d=np.asarray(d["d"]))
    diag = np.asarray(A.diagonal()) if d["diag_scale"] else None
    if k == 3:
        c = np.asarray(d["coeffs"])
        return R.InvOp("horner", A=A, coeffs=c, diag=diag, neumann=d["neumann"],
                       spmvs=int(np.count_nonzero(c[:-1])))
    if k == 4:
        op = R.InvOp("newton", A=A, re=np.asarray(d["re"]), im=np.asarray(d["im"]), diag=diag)
        op.spmvs = R.newton_spmv_count(op.re, op.im)
        return op
    raise ValueError(k)


def _inv_from_dict_scipy(d, A):
    """same as _inv_from_dict for dicts whose matrices are already scipy CSR"""
    if d["kind"] == 1:
        return R.InvOp("assembled", mat=d["mat"])
    return _inv_from_dict(d, A)


def hierarchy_from_product(pc, A_top) -> R.Hierarchy:
    H = R.Hierarchy(A=A_top)
    nl = pc.num_levels
    for l in range(nl - 1):
        e = pc.export_level(l)
        F = e["F"].astype(np.int64); C = e["C"].astype(np.int64)
        n = e["n"]
        A_ff = R.DiagMat(e["aff_diag"]) if e["aff_diag"] is not None and e.get("A_ff") is None else to_scipy(e["A_ff"])
        A_fc = to_scipy(e["A_fc"]); A_cf = to_scipy(e["A_cf"]); Z = to_scipy(e["Z"])
        if e["one_point"]:
            wc = e["wcol"]
            rows = np.flatnonzero(wc >= 0)
            W = sp.csr_matrix((np.ones(rows.size), (rows, wc[rows])), shape=(F.size, C.size))
        else:
            W = to_scipy(e["W"])
        A_full = to_scipy(e["A_full"]) if e.get("A_full") is not None else None
        inv_ff = _inv_from_dict(e["inv_ff"], A_full if A_full is not None else A_ff)
        if A_full is not None:
            H.full_smoothing = True
        A_cc = inv_cc = None
        if "A_cc" in e:
            A_cc = to_scipy(e["A_cc"])
            inv_cc = _inv_from_dict(e["inv_cc"], A_cc)
        so = [s for s in e["smooth_order"] if s != 0]
        H.levels.append(R.Level(n=n, F=F, C=C, A_ff=A_ff, A_fc=A_fc, A_cf=A_cf, A_cc=A_cc, inv_A_ff=inv_ff,
                                inv_A_cc=inv_cc, Z=Z, W=W, R=R.build_R(Z, F, C, n), P=R.build_P(W, F, C, n),
                                smooth_order=so, aff_diag=e["aff_diag"] is not None and e.get("A_ff") is None,
                                A_full=A_full))
        H.nnzs.append(e["nnz_A"])
    cA, cinv = pc.export_coarse()
    H.coarse_A = to_scipy(cA)
    H.coarse_inv = _inv_from_dict(cinv, H.coarse_A)
    return H


def _inv_to_dict(op: R.InvOp):
    if op.kind == "assembled":
        m = op.mat.tocsr(); m.sort_indices()
        return dict(kind=1, mat=m)
    if op.kind == "diag":
        return dict(kind=2, d=op.d)
    if op.kind == "horner":
        return dict(kind=3, coeffs=op.coeffs, diag_scale=op.diag is not None, neumann=op.neumann)
    if op.kind == "newton":
        return dict(kind=4, re=op.re, im=op.im, diag_scale=op.diag is not None)
    raise ValueError(op.kind)


def levels_for_product(H: R.Hierarchy):
    """air_ref.Hierarchy -> arguments of PCAIR.set_hierarchy"""
    levels = []
    for L in H.levels:
        d = dict(n=L.n, F=L.F, C=L.C, A_fc=L.A_fc, Z=L.Z, A_cf=L.A_cf, inv_ff=_inv_to_dict(L.inv_A_ff),
                 smooth_order=list(L.smooth_order))
        if isinstance(L.A_ff, R.DiagMat):
            d["aff_diag"] = L.A_ff.d
        else:
            d["A_ff"] = L.A_ff
        W = L.W.tocsr()
        if W.nnz <= W.shape[0] and np.all(W.data == 1.0) and np.all(np.diff(W.indptr) <= 1):
            wc = np.full(W.shape[0], -1, dtype=np.int32)
            rows = np.flatnonzero(np.diff(W.indptr) == 1)
            wc[rows] = W.indices[W.indptr[rows]]
            d["wcol"] = wc
        else:
            d["W"] = W
        if L.A_cc is not None:
            d["A_cc"] = L.A_cc
            d["inv_cc"] = _inv_to_dict(L.inv_A_cc)
        levels.append(d)
    return levels, H.coarse_A, _inv_to_dict(H.coarse_inv)
