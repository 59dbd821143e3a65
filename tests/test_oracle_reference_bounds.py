"""Pins the oracle with the reference's OWN known-answer tests: the reference
suite asserts convergence within -ksp_max_it N; every case below is one of its command lines
(tests/Makefile line cited) run through the oracle restatement.  The reference itself cannot
be built here (PETSc + Fortran), so this is the strongest pin available."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import air_ref as R

f32 = lambda v: float(np.float32(v))


def solve(A, o, ksp="gmres", side="left", rtol=1e-5, max_it=50, norm="preconditioned"):
    H = R.setup_air(A, o)
    n = A.shape[0]
    b = np.zeros(n); x0 = np.ones(n)       # tests/adv_diff_fd.c:179,325 ; tests/adv_1d.c:118,160
    pc = lambda r: R.vcycle(H, r)
    if ksp == "gmres":
        x, its, reason, rn = R.ksp_gmres(A, pc, b, x0, rtol=rtol, max_it=max_it, pc_side=side)
    else:
        x, its, reason, rn = R.ksp_richardson(A, pc, b, x0, rtol=rtol, max_it=max_it, norm_type=norm)
    return its, reason, x, H


DIFF8 = dict(dim=2, M=8, N=8, u=0, v=0, alpha=1.0)
NEWTON_COARSE = dict(coarsest_inverse_type=R.INV_NEWTON, coarsest_poly_order=10, coarsest_matrix_free_polys=True)

CASES = [
    # (label, matrix, options, ksp kwargs, bound)
    ("Makefile:421-424 AIRG as an exact solver", R.gen_adv_diff_fd(**DIFF8),
     R.AirOptions(strong_threshold=0.0, a_drop=0.0, r_drop=0.0, inverse_type=R.INV_JACOBI, coarse_richardson_its=10),
     dict(ksp="richardson", norm="unpreconditioned", max_it=1), 1),
    ("Makefile:537-540 adv_1d n=1000", R.gen_adv_1d(1000),
     R.AirOptions(a_drop=1e-3, inverse_type=R.INV_POWER, **NEWTON_COARSE),
     dict(side="right", rtol=1e-10, max_it=2), 2),
    ("Makefile:1323 adv 25x25 defaults", R.gen_adv_diff_fd(2, 25, 25), R.AirOptions(), dict(max_it=5), 5),
    ("Makefile:300 adv 8x8 fc smoothing", R.gen_adv_diff_fd(2, 8, 8), R.AirOptions(smooth_type="fc"), dict(max_it=3), 3),
    ("Makefile:543-545 3D 10^3 a_lump right rtol 1e-10", R.gen_adv_diff_fd(3, 10, 10, 10), R.AirOptions(a_lump=True),
     dict(side="right", rtol=1e-10, max_it=4), 4),
    ("Makefile:388-389 diffusion 8x8 defaults", R.gen_adv_diff_fd(**DIFF8), R.AirOptions(), dict(max_it=5), 5),
    ("Makefile:390-392 diffusion arnoldi a_drop 1e-3", R.gen_adv_diff_fd(**DIFF8),
     R.AirOptions(inverse_type=R.INV_ARNOLDI, coarsest_inverse_type=R.INV_ARNOLDI, a_drop=1e-3), dict(max_it=5), 5),
    ("Makefile:398-399 diffusion neumann", R.gen_adv_diff_fd(**DIFF8),
     R.AirOptions(inverse_type=R.INV_NEUMANN, a_drop=1e-3), dict(max_it=5), 5),
    ("Makefile:400-402 diffusion neumann matrix-free", R.gen_adv_diff_fd(**DIFF8),
     R.AirOptions(inverse_type=R.INV_NEUMANN, a_drop=1e-3, matrix_free_polys=True), dict(max_it=5), 5),
    ("Makefile:413-414 diffusion weighted jacobi", R.gen_adv_diff_fd(**DIFF8),
     R.AirOptions(inverse_type=R.INV_WJACOBI, a_drop=1e-3), dict(max_it=8), 8),
    ("Makefile:417-418 diffusion jacobi", R.gen_adv_diff_fd(**DIFF8),
     R.AirOptions(inverse_type=R.INV_JACOBI, a_drop=1e-3), dict(max_it=5), 5),
    ("Makefile:405-406 diffusion SAI", R.gen_adv_diff_fd(**DIFF8),
     R.AirOptions(inverse_type=R.INV_SAI, a_drop=1e-3), dict(max_it=5), 5),
    ("Makefile:409-410 diffusion ISAI", R.gen_adv_diff_fd(**DIFF8),
     R.AirOptions(inverse_type=R.INV_ISAI, a_drop=1e-3), dict(max_it=5), 5),
    ("Makefile:638-642 lAIR 2D advection 50x50 ffc wjacobi", R.gen_adv_diff_fd(2, 50, 50),
     R.AirOptions(a_lump=True, a_drop=1e-4, smooth_type="ffc", z_type="lair", inverse_type=R.INV_WJACOBI),
     dict(side="right", rtol=1e-10, max_it=10), 10),
    ("Makefile:698-699 improve Z 3 its", R.gen_adv_diff_fd(2, 50, 50),
     R.AirOptions(improve_z_its=3, a_drop=1e-3, inverse_type=R.INV_POWER), dict(max_it=3), 3),
    ("Makefile:700-701 improve Z 1 it", R.gen_adv_diff_fd(2, 50, 50),
     R.AirOptions(improve_z_its=1, a_drop=1e-3, inverse_type=R.INV_POWER), dict(max_it=3), 3),
    ("Makefile:704-706 improve W 3 its", R.gen_adv_diff_fd(2, 50, 50),
     R.AirOptions(improve_w_its=3, a_drop=1e-3, inverse_type=R.INV_POWER, one_point_classical_prolong=False), dict(max_it=4), 4),
    ("Makefile:707-709 improve W 1 it", R.gen_adv_diff_fd(2, 50, 50),
     R.AirOptions(improve_w_its=1, a_drop=1e-3, inverse_type=R.INV_POWER, one_point_classical_prolong=False), dict(max_it=4), 4),
    ("Makefile:556-560 auto truncation + newton coarse", R.gen_adv_diff_fd(2, 10, 10),
     R.AirOptions(auto_truncate_start_level=1, auto_truncate_tol=1e-2, a_drop=1e-3, inverse_type=R.INV_POWER,
                  **NEWTON_COARSE), dict(ksp="richardson", max_it=3), 3),
    ("Makefile:1121-1133 adv 100x100 a_lump a_drop 1e-5 strong 0.99", R.gen_adv_diff_fd(2, 100, 100),
     R.AirOptions(a_lump=True, a_drop=1e-5, strong_threshold=0.99), dict(side="right", rtol=1e-10, max_it=6), 6),
]


@pytest.mark.parametrize("label,A,opts,kw,bound", CASES, ids=[c[0] for c in CASES])
def test_reference_iteration_bounds(label, A, opts, kw, bound):
    its, reason, x, H = solve(A, opts, **kw)
    assert reason > 0, f"{label}: did not converge (reason {reason}) within {bound}"
    assert its <= bound
    if "exact solver" in label:
        assert np.linalg.norm(x) < 1e-7        # exact solution is 0 (b = 0)


@pytest.mark.parametrize("itype,mf", [(R.INV_NEWTON, True), (R.INV_ARNOLDI, False), (R.INV_POWER, False)])
@pytest.mark.parametrize("diag", ["ones", "1.5", "two_values"])
def test_mat_diag_polynomials_exact(itype, mf, diag):
    """tests/mat_diag.F90 + tests/Makefile:577-582: GMRES polynomials are exact on constant-diagonal
    matrices (1 iteration with the default order-6 polynomial)."""
    n = 100
    d = {"ones": np.ones(n), "1.5": np.full(n, 1.5), "two_values": np.r_[2.5, np.full(n - 1, 1.5)]}[diag]
    A = sp.diags(d).tocsr()
    op = R.pcpflareinv_setup(A, itype, 6, 1, mf)
    b = np.random.default_rng(0).random(n)
    x, its, reason, rn = R.ksp_gmres(A, lambda r: R.apply_inv(op, r), b, np.zeros(n), max_it=1)
    assert reason > 0 and its <= 1
    assert np.linalg.norm(x - b / d) < 1e-12


@pytest.mark.parametrize("M,order", [(5, 16), (10, 50)])
def test_high_order_newton_exact_on_small_systems(M, order):
    """tests/Makefile:548-553: matrix-free Newton polynomial of order 16 / 50 solves the 5x5 / 10x10
    advection problem in one iteration."""
    A = R.gen_adv_diff_fd(2, M, M)
    n = A.shape[0]
    op = R.pcpflareinv_setup(A, R.INV_NEWTON, order, 1, True)
    x, its, reason, rn = R.ksp_gmres(A, lambda r: R.apply_inv(op, r), np.zeros(n), np.ones(n), max_it=1)
    assert reason > 0 and its <= 1


def test_pflareinv_power_bound():
    """tests/Makefile:394-395: single-level power-basis GMRES polynomial, <= 8 iterations."""
    A = R.gen_adv_diff_fd(**DIFF8)
    n = A.shape[0]
    op = R.pcpflareinv_setup(A, R.INV_POWER, 6, 1, False)
    x, its, reason, rn = R.ksp_gmres(A, lambda r: R.apply_inv(op, r), np.zeros(n), np.ones(n), max_it=8)
    assert reason > 0 and its <= 8


def test_level_sizes_of_config_table():
    """SURVEY.md 8 level-1 sizes computed from the generators."""
    A = R.gen_adv_1d(100000)
    assert A.shape[0] == 100000 and A.nnz == 199999
    A = R.gen_adv_diff_fd(2, 100, 100)
    assert A.nnz == 3 * 99 * 99 + (100 * 100 - 99 * 99)
