"""Generates tests/golden/*.npz.

The reference (PFLARE on PETSc) cannot be run in this environment, so these fixtures are NOT
reference outputs: they are small complete hierarchies built by the CPU oracle
(oracle/air_ref.py, itself pinned by the reference's own iteration-bound tests) together with
the oracle's V-cycle output and Krylov histories on them.  They freeze the oracle so that
(a) later edits to the oracle, (b) the C oracle and (c) the HIP path through the bring-your-own
hierarchy ABI are all checked against the same committed data.

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import air_ref as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def lcg(n):
    i = np.arange(n, dtype=np.uint64)
    return ((np.uint64(1103515245) * (i + np.uint64(12345))) % np.uint64(2 ** 31)).astype(np.float64) / 2.0 ** 30 - 1.0


def put_csr(d, key, A):
    A = A.tocsr(); A.sort_indices()
    d[key + "_indptr"] = A.indptr.astype(np.int64)
    d[key + "_indices"] = A.indices.astype(np.int32)
    d[key + "_data"] = A.data.astype(np.float64)
    d[key + "_shape"] = np.array(A.shape, dtype=np.int64)


def put_inv(d, key, op):
    kinds = {"assembled": 1, "diag": 2, "horner": 3, "newton": 4}
    d[key + "_kind"] = np.array(kinds[op.kind])
    d[key + "_dscale"] = np.array(int(op.diag is not None))
    d[key + "_neumann"] = np.array(int(op.neumann))
    if op.kind == "assembled":
        put_csr(d, key + "_mat", op.mat)
    elif op.kind == "diag":
        d[key + "_d"] = op.d
    elif op.kind == "horner":
        d[key + "_coeffs"] = op.coeffs
    else:
        d[key + "_re"] = op.re; d[key + "_im"] = op.im


def dump(name, A, opts, ksp):
    H = R.setup_air(A, opts)
    d = {}
    put_csr(d, "A", A)
    d["nlevels"] = np.array(H.no_levels)
    for l, L in enumerate(H.levels):
        p = f"L{l}_"
        d[p + "F"] = L.F.astype(np.int32); d[p + "C"] = L.C.astype(np.int32)
        if isinstance(L.A_ff, R.DiagMat):
            d[p + "aff_diag"] = L.A_ff.d
        else:
            put_csr(d, p + "A_ff", L.A_ff)
        put_csr(d, p + "A_fc", L.A_fc); put_csr(d, p + "A_cf", L.A_cf)
        put_csr(d, p + "Z", L.Z); put_csr(d, p + "W", L.W)
        put_inv(d, p + "inv_ff", L.inv_A_ff)
        if L.A_cc is not None:
            put_csr(d, p + "A_cc", L.A_cc); put_inv(d, p + "inv_cc", L.inv_A_cc)
        so = list(L.smooth_order) + [0] * 10
        d[p + "smooth"] = np.array(so[:10], dtype=np.int32)
    put_csr(d, "coarse_A", H.coarse_A)
    put_inv(d, "coarse_inv", H.coarse_inv)
    d["coarse_rich"] = np.array(H.coarse_richardson_its)
    n = A.shape[0]
    x = lcg(n)
    d["vcycle_in"] = x
    d["vcycle_out"] = R.vcycle(H, x)
    pc = lambda r: R.vcycle(H, r)
    hist = []
    if ksp["type"] == "gmres":
        xs, its, reason, rn = R.ksp_gmres(A, pc, np.zeros(n), np.ones(n), rtol=ksp["rtol"], max_it=ksp["max_it"],
                                          pc_side=ksp["side"], history=hist)
    else:
        xs, its, reason, rn = R.ksp_richardson(A, pc, np.zeros(n), np.ones(n), rtol=ksp["rtol"], max_it=ksp["max_it"],
                                               norm_type=ksp["norm"], history=hist)
    d["ksp_type"] = np.array(0 if ksp["type"] == "richardson" else 1)
    d["ksp_side"] = np.array(1 if ksp.get("side") == "right" else 0)
    d["ksp_norm"] = np.array(1 if ksp.get("norm") == "unpreconditioned" else 0)
    d["ksp_rtol"] = np.array(ksp["rtol"]); d["ksp_max_it"] = np.array(ksp["max_it"])
    d["ksp_its"] = np.array(its); d["ksp_reason"] = np.array(reason); d["ksp_rnorm"] = np.array(rn)
    d["ksp_history"] = np.array(hist); d["ksp_x"] = xs
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(f"{name}: levels={H.no_levels} its={its} reason={reason} rnorm={rn:.3e} "
          f"size={os.path.getsize(os.path.join(HERE, name + '.npz'))/1024:.0f} KiB")


if __name__ == "__main__":
    dump("adv2d_14_default", R.gen_adv_diff_fd(2, 14, 14), R.AirOptions(),
         dict(type="gmres", side="left", rtol=1e-5, max_it=5))
    dump("adv3d_8_lump_right", R.gen_adv_diff_fd(3, 8, 8, 8), R.AirOptions(a_lump=True),
         dict(type="gmres", side="right", rtol=1e-10, max_it=4))
    dump("diff2d_8_neumann_mf", R.gen_adv_diff_fd(2, 8, 8, u=0, v=0, alpha=1.0),
         R.AirOptions(inverse_type=R.INV_NEUMANN, a_drop=1e-3, matrix_free_polys=True),
         dict(type="gmres", side="left", rtol=1e-5, max_it=5))
    dump("adv2d_8_fc", R.gen_adv_diff_fd(2, 8, 8), R.AirOptions(smooth_type="fc"),
         dict(type="gmres", side="left", rtol=1e-5, max_it=3))
    dump("adv1d_200_newton_coarse", R.gen_adv_1d(200),
         R.AirOptions(coarsest_inverse_type=R.INV_NEWTON, coarsest_poly_order=10, coarsest_matrix_free_polys=True,
                      a_drop=1e-3, inverse_type=R.INV_POWER),
         dict(type="gmres", side="right", rtol=1e-10, max_it=4))
    dump("diff2d_8_exact_jacobi", R.gen_adv_diff_fd(2, 8, 8, u=0, v=0, alpha=1.0),
         R.AirOptions(strong_threshold=0.0, a_drop=0.0, r_drop=0.0, inverse_type=R.INV_JACOBI,
                      coarse_richardson_its=10),
         dict(type="richardson", norm="unpreconditioned", rtol=1e-5, max_it=1))
