"""Loader for tests/golden/*.npz (see tests/golden/make_golden.py)."""
import glob
import os

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(HERE, "golden", "*.npz")))


def _csr(d, key):
    return sp.csr_matrix((d[key + "_data"], d[key + "_indices"], d[key + "_indptr"]), shape=tuple(d[key + "_shape"]))


def _inv(d, key):
    kind = int(d[key + "_kind"])
    out = dict(kind=kind, diag_scale=bool(d[key + "_dscale"]), neumann=bool(d[key + "_neumann"]))
    if kind == 1:
        out["mat"] = _csr(d, key + "_mat")
    elif kind == 2:
        out["d"] = d[key + "_d"]
    elif kind == 3:
        out["coeffs"] = d[key + "_coeffs"]
    else:
        out["re"] = d[key + "_re"]; out["im"] = d[key + "_im"]
    return out


def load(name):
    """returns (A, levels, coarse_A, coarse_inv, coarse_rich, data) with `levels` in the format of
    pflare_amd.PCAIR.set_hierarchy."""
    d = np.load(os.path.join(HERE, "golden", name + ".npz"))
    A = _csr(d, "A")
    levels = []
    for l in range(int(d["nlevels"]) - 1):
        p = f"L{l}_"
        lv = dict(F=d[p + "F"], C=d[p + "C"], A_fc=_csr(d, p + "A_fc"), A_cf=_csr(d, p + "A_cf"), Z=_csr(d, p + "Z"),
                  inv_ff=_inv(d, p + "inv_ff"), smooth_order=[int(s) for s in d[p + "smooth"]])
        lv["n"] = lv["F"].size + lv["C"].size
        if p + "aff_diag" in d:
            lv["aff_diag"] = d[p + "aff_diag"]
        else:
            lv["A_ff"] = _csr(d, p + "A_ff")
        W = _csr(d, p + "W")
        if np.all(np.diff(W.indptr) <= 1) and np.all(W.data == 1.0):
            wc = np.full(W.shape[0], -1, dtype=np.int32)
            rows = np.flatnonzero(np.diff(W.indptr) == 1)
            wc[rows] = W.indices[W.indptr[rows]]
            lv["wcol"] = wc
        else:
            lv["W"] = W
        lv["W_mat"] = W
        if p + "A_cc_data" in d:
            lv["A_cc"] = _csr(d, p + "A_cc"); lv["inv_cc"] = _inv(d, p + "inv_cc")
        levels.append(lv)
    return A, levels, _csr(d, "coarse_A"), _inv(d, "coarse_inv"), int(d["coarse_rich"]), d


def oracle_hierarchy(name):
    """the fixture as an oracle (air_ref) Hierarchy"""
    from oracle import air_ref as R, bridge
    A, levels, cA, cinv, rich, d = load(name)
    H = R.Hierarchy(A=A, coarse_richardson_its=rich)
    for lv in levels:
        F = lv["F"].astype(np.int64); C = lv["C"].astype(np.int64); n = lv["n"]
        A_ff = R.DiagMat(lv["aff_diag"]) if "aff_diag" in lv else lv["A_ff"]
        A_cc = lv.get("A_cc")
        H.levels.append(R.Level(n=n, F=F, C=C, A_ff=A_ff, A_fc=lv["A_fc"], A_cf=lv["A_cf"], A_cc=A_cc,
                                inv_A_ff=bridge._inv_from_dict_scipy(lv["inv_ff"], A_ff),
                                inv_A_cc=bridge._inv_from_dict_scipy(lv["inv_cc"], A_cc) if A_cc is not None else None,
                                Z=lv["Z"], W=lv["W_mat"], R=R.build_R(lv["Z"], F, C, n), P=R.build_P(lv["W_mat"], F, C, n),
                                smooth_order=[s for s in lv["smooth_order"] if s != 0], aff_diag="aff_diag" in lv))
    H.coarse_A = cA
    H.coarse_inv = bridge._inv_from_dict_scipy(cinv, cA)
    return H, d
