"""The Krylov solver's tuning keys that are NOT bitwise neutral (include/airhip.h, airhip_set_tuning): what each may change
and by how much.  Reference semantics: PETSc KSPGMRES with right preconditioning (tests/Makefile:543-545, 1123-1124) -
classical Gram-Schmidt, normalised basis, solution x += M (V y) (one PCApply after every restart cycle).

  ksp_keep_preconditioned  1 (default): z_j = M v_j is kept and x += Z y.  Same Arnoldi process: iteration count, reason
                           and EVERY residual norm of the history bit for bit; pc_applies = iterations instead of
                           iterations + restart cycles; the solution agrees with PETSc's form to 1e-12 of the correction
                           and with the C oracle's (PETSc's form on the host) to the parity tolerance.
  ksp_unnormalised         1 (default): basis vectors keep their norms.  Same iteration count, history within 1e-7 relative.
  krylov_blocks            grid of the dot / axpy kernels = summation order.  Same iteration count, history within 1e-9.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NEWTON_COARSE = dict(coarsest_inverse_type="newton", coarsest_matrix_free_polys=True, coarsest_poly_order=10)
DEFAULTS = {"ksp_keep_preconditioned": 1, "ksp_unnormalised": 1, "krylov_blocks": 512}


def _lcg(n):
    i = np.arange(n, dtype=np.uint64)
    return (((np.uint64(1103515245) * (i + np.uint64(12345))) % np.uint64(2 ** 31)).astype(np.float64) / float(2 ** 30) - 1.0)


@pytest.fixture(scope="module")
def problem(gpu):
    import torch
    import pflare_amd as P
    A = P.HostMat.adv_diff_fd(dim=3, M=40, N=40, P=40)
    pc = P.PCAIR(a_lump=True, auto_truncate_start_level=1, auto_truncate_tol=1e-1, **NEWTON_COARSE).setup(A)
    yield torch, P, A, pc
    for k, v in DEFAULTS.items():
        P.lib().airhip_set_tuning(k.encode(), v)


def _solve(torch, P, A, pc, tuning, b=None, **kw):
    for k, v in DEFAULTS.items():
        assert P.lib().airhip_set_tuning(k.encode(), tuning.get(k, v)) == 0
    n = A.shape[0]
    bd = torch.zeros(n, dtype=torch.float64, device="cuda") if b is None else torch.from_numpy(b).cuda()
    x = torch.ones(n, dtype=torch.float64, device="cuda")
    opts = dict(ksp_type="gmres", pc_side="right", rtol=1e-10, atol=1e-50, max_it=60)
    opts.update(kw)
    res = P.KSP(**opts).solve(bd, x, pc=pc)
    return res, x.cpu().numpy()


@pytest.mark.parametrize("restart", [30, 4], ids=["one-cycle", "restarts"])
@pytest.mark.parametrize("rhs", ["zero", "lcg"])
def test_kept_preconditioned_basis_is_the_same_arnoldi_process(problem, restart, rhs):
    torch, P, A, pc = problem
    n = A.shape[0]
    b = None if rhs == "zero" else _lcg(n)
    r1, x1 = _solve(torch, P, A, pc, {"ksp_keep_preconditioned": 1}, b=b, restart=restart)
    r0, x0 = _solve(torch, P, A, pc, {"ksp_keep_preconditioned": 0}, b=b, restart=restart)
    assert r1["reason"] > 0 and r1["reason"] == r0["reason"] and r1["its"] == r0["its"], (r1, r0)
    # within one restart cycle the Arnoldi process is the same bit for bit; after a restart the new cycle starts from a
    # solution that differs in its last rounding, so the later entries agree to rounding only
    k = min(restart, r1["its"]) + 1
    assert r1["history"][:k] == r0["history"][:k]
    assert len(r1["history"]) == len(r0["history"])
    h1, h0 = np.array(r1["history"]), np.array(r0["history"])
    assert np.all(np.abs(h1 - h0) <= 1e-9 * h0[0] + 1e-6 * h0), (h1, h0)
    cycles = -(-r1["its"] // restart)
    assert r1["pc_applies"] == r1["its"], r1
    assert r0["pc_applies"] == r0["its"] + cycles, (r0, cycles)
    corr = np.linalg.norm(x0 - 1.0)
    assert np.linalg.norm(x1 - x0) <= 1e-11 * corr, (np.linalg.norm(x1 - x0), corr)
    # the true residual of both solutions meets the tolerance the estimate promised
    import scipy.sparse as sp
    a = A.csr()
    S = sp.csr_matrix((a.val, a.col, a.rowptr), shape=a.shape)
    bb = np.zeros(n) if b is None else b
    r_init = np.linalg.norm(bb - S @ np.ones(n))
    for x in (x1, x0):
        assert np.linalg.norm(bb - S @ x) <= 2e-10 * r_init


def test_kept_basis_against_the_c_oracle(problem):
    """the oracle runs PETSc's form on the host: identical iteration count and reason, residual norms within the parity
    tolerance, same solution"""
    torch, P, A, pc = problem
    from oracle import c_oracle
    n = A.shape[0]
    b = _lcg(n)
    res, x = _solve(torch, P, A, pc, {}, b=b)
    ch = c_oracle.CHierarchy(pc)
    xo, its, reason, rn, hist = ch.ksp(A, b, np.ones(n), "gmres", "right", rtol=1e-10, atol=1e-50, max_it=60)
    assert res["its"] == its and res["reason"] == reason, (res, its, reason)
    assert len(hist) == len(res["history"])
    assert max(abs(a - c) / c for a, c in zip(res["history"], hist) if c > 1e-12 * hist[0]) <= 1e-6
    assert np.linalg.norm(x - xo) <= 1e-9 * np.linalg.norm(xo)


def test_left_preconditioning_and_richardson_are_untouched(problem):
    torch, P, A, pc = problem
    for kw in (dict(pc_side="left"), dict(ksp_type="richardson", pc_side="left", rtol=1e-8)):
        r1, x1 = _solve(torch, P, A, pc, {"ksp_keep_preconditioned": 1}, **kw)
        r0, x0 = _solve(torch, P, A, pc, {"ksp_keep_preconditioned": 0}, **kw)
        assert r1["its"] == r0["its"] and r1["history"] == r0["history"] and r1["pc_applies"] == r0["pc_applies"], (kw, r1, r0)
        assert np.array_equal(x1, x0)


def test_unnormalised_basis_keeps_the_iteration_count(problem):
    """ksp_unnormalised changes the rounding of every Hessenberg entry - same iteration counts, histories
    within a stated tolerance"""
    torch, P, A, pc = problem
    for b in (None, _lcg(A.shape[0])):
        r1, _ = _solve(torch, P, A, pc, {"ksp_unnormalised": 1}, b=b)
        r0, _ = _solve(torch, P, A, pc, {"ksp_unnormalised": 0}, b=b)
        assert r1["its"] == r0["its"] and r1["reason"] == r0["reason"], (r1, r0)
        h1, h0 = np.array(r1["history"]), np.array(r0["history"])
        big = h0 > 1e-11 * h0[0]
        assert np.max(np.abs(h1[big] - h0[big]) / h0[big]) <= 1e-7


def test_grid_of_the_krylov_kernels_only_moves_the_last_digits(problem):
    torch, P, A, pc = problem
    ref, _ = _solve(torch, P, A, pc, {})
    for blocks in (64, 2048):
        r, _ = _solve(torch, P, A, pc, {"krylov_blocks": blocks})
        assert r["its"] == ref["its"] and r["reason"] == ref["reason"], (blocks, r, ref)
        h1, h0 = np.array(r["history"]), np.array(ref["history"])
        big = h0 > 1e-11 * h0[0]
        assert np.max(np.abs(h1[big] - h0[big]) / h0[big]) <= 1e-9
