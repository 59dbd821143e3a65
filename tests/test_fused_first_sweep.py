"""The operator of the fused first F sweep, G = A_fc + A_ff W (one-point W): host builder against scipy (CPU), device builder
bit for bit against the host builder, and the fused cycle against the reference's sequence of products and against the oracle
(GPU).  Reference: the F-point relaxation of src/FC_Smooth.F90:499-568 after the prolongation (src/AIR_MG_Setup.F90:1005):
x0 = W x_c, t4 = b_f - A_fc x_c, x_{k+1} = x_k + M (t4 - A_ff x_k).  Fused: r0 = b_f - G x_c; d = M r_k, x += d,
r_{k+1} = r_k - A_ff d."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp


def _fused(P, A_ff, A_fc, wcol, on_device):
    lib = P.lib()
    fn = lib.airhip_fused_first_operator
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    a, b = P.as_csr(A_ff), P.as_csr(A_fc)
    w = np.ascontiguousarray(wcol, dtype=np.int32)
    h = C.c_void_p()
    rc = fn(a.ref(), b.ref(), w.ctypes.data_as(C.c_void_p), int(on_device), C.byref(h))
    assert rc == 0, lib.airhip_last_error().decode()
    return P.HostMat(h).csr()


def _blocks(nf, nc, per_ff, per_fc, seed):
    rng = np.random.default_rng(seed)

    def rand(m, n, k):
        rows = np.repeat(np.arange(m), k)
        cols = rng.integers(0, n, size=m * k)
        vals = rng.standard_normal(m * k) * 10.0 ** rng.integers(-3, 2, size=m * k)
        A = sp.csr_matrix((vals, (rows, cols)), shape=(m, n))
        A.sum_duplicates(); A.sort_indices()
        return A.tocsr()
    A_ff = (rand(nf, nf, per_ff) + sp.eye(nf, format="csr") * 4.0).tocsr()
    A_ff.sort_indices()
    A_fc = rand(nf, nc, per_fc)
    wcol = rng.integers(-1, nc, size=nf).astype(np.int32)       # some F points without a W entry
    return A_ff, A_fc, wcol


def _scipy_G(A_ff, A_fc, wcol):
    nf, nc = A_fc.shape
    has = wcol >= 0
    W = sp.csr_matrix((np.ones(int(has.sum())), (np.flatnonzero(has), wcol[has])), shape=(nf, nc))
    G = (A_fc + A_ff @ W).tocsr()
    G.sort_indices()
    return G


@pytest.mark.parametrize("nf,nc,kff,kfc", [(300, 500, 4, 3), (1, 1, 1, 1), (2000, 700, 30, 45), (50, 4000, 2, 300)])
def test_host_builder_is_a_fc_plus_a_ff_w(nf, nc, kff, kfc):
    import pflare_amd as P
    A_ff, A_fc, wcol = _blocks(nf, nc, min(kff, nf), min(kfc, nc), 3)
    g = _fused(P, A_ff, A_fc, wcol, on_device=False)
    G = sp.csr_matrix((g.val, g.col, g.rowptr), shape=g.shape)
    R = _scipy_G(A_ff, A_fc, wcol)
    # pattern: every structural entry of A_fc and of A_ff W (scipy drops nothing either); values to rounding
    assert G.shape == R.shape and G.nnz >= R.nnz
    D = (G - R).tocsr()
    assert D.nnz == 0 or np.max(np.abs(D.data)) <= 1e-12 * max(1.0, np.max(np.abs(R.data)))
    assert np.all(np.diff(g.rowptr) >= np.diff(A_fc.indptr))
    for i in range(min(nf, 50)):
        c = g.col[g.rowptr[i]:g.rowptr[i + 1]]
        assert np.all(np.diff(c) > 0)               # sorted, no duplicates


def test_fused_operator_on_a_real_hierarchy_is_little_larger_than_a_fc():
    """G = A_fc + A_ff W holds 1.0-1.3 x the non-zeros of A_fc on the levels of a 3D advection hierarchy (the W images of
    a row's F neighbours mostly are its C neighbours), far fewer than nnz(A_fc) + nnz(A_ff) - which is why one product with G
    replaces two; and it is A_fc + A_ff W on every level (host builder against scipy)"""
    import pflare_amd as P
    A = P.HostMat.adv_diff_fd(dim=3, M=40, N=40, P=40)
    pc = P.PCAIR(a_lump=True, auto_truncate_start_level=1, auto_truncate_tol=1e-1, coarsest_inverse_type="newton",
                 coarsest_matrix_free_polys=True, coarsest_poly_order=10)
    pc.setup_host_only(A)
    seen = 0
    for l in range(pc.num_levels - 1):
        lv = pc.export_level(l)
        if "A_ff" not in lv:
            continue                               # diagonal A_ff: that level's sweep is one launch already
        a_ff = sp.csr_matrix((lv["A_ff"].val, lv["A_ff"].col, lv["A_ff"].rowptr), shape=lv["A_ff"].shape)
        a_fc = sp.csr_matrix((lv["A_fc"].val, lv["A_fc"].col, lv["A_fc"].rowptr), shape=lv["A_fc"].shape)
        wcol = np.asarray(lv["wcol"], dtype=np.int32)
        g = _fused(P, a_ff, a_fc, wcol, on_device=False)
        G = sp.csr_matrix((g.val, g.col, g.rowptr), shape=g.shape)
        D = (G - _scipy_G(a_ff, a_fc, wcol)).tocsr()
        assert D.nnz == 0 or np.max(np.abs(D.data)) <= 1e-12 * np.max(np.abs(G.data))
        assert a_fc.nnz <= G.nnz <= 1.35 * a_fc.nnz, (l, a_fc.nnz, a_ff.nnz, G.nnz)
        assert G.nnz <= 0.9 * (a_fc.nnz + a_ff.nnz), (l, a_fc.nnz, a_ff.nnz, G.nnz)
        seen += 1
    assert seen >= 3


@pytest.mark.gpu
@pytest.mark.parametrize("nf,nc,kff,kfc", [(300, 500, 4, 3), (1, 1, 1, 1), (5000, 3000, 30, 45), (200, 9000, 2, 700), (40000, 30000, 9, 12)])
def test_device_builder_gives_the_host_builders_bits(gpu, nf, nc, kff, kfc):
    import pflare_amd as P
    A_ff, A_fc, wcol = _blocks(nf, nc, min(kff, nf), min(kfc, nc), 5)
    h = _fused(P, A_ff, A_fc, wcol, on_device=False)
    d = _fused(P, A_ff, A_fc, wcol, on_device=True)
    assert h.shape == d.shape and np.array_equal(h.rowptr, d.rowptr) and np.array_equal(h.col, d.col)
    assert np.array_equal(h.val, d.val)              # bitwise


def _lcg(n):
    i = np.arange(n, dtype=np.uint64)
    return (((np.uint64(1103515245) * (i + np.uint64(12345))) % np.uint64(2 ** 31)).astype(np.float64) / float(2 ** 30) - 1.0)


CYCLE_CASES = {
    "adv3d_24_ff": (dict(dim=3, M=24, N=24, P=24), dict(a_lump=True)),
    "adv2d_60_fff": (dict(dim=2, M=60, N=60), dict(smooth_type="fff")),
    "adv2d_60_f": (dict(dim=2, M=60, N=60), dict(smooth_type="f")),
    "adv2d_40_fcf": (dict(dim=2, M=40, N=40), dict(smooth_type="fcf")),          # the second F group needs A_fc itself
    "adv2d_40_fc": (dict(dim=2, M=40, N=40), dict(smooth_type="fc")),
    "diff2d_30_diagscale": (dict(dim=2, M=30, N=30, u=0.0, v=0.0, alpha=1.0), dict(diag_scale_polys=True)),
    # matrix-free polynomial inverses (Horner / Newton / Neumann): d = M r is the polynomial, one A_ff product less per level
    "adv2d_50_mf": (dict(dim=2, M=50, N=50), dict(matrix_free_polys=True)),
    "adv2d_50_mf_fff": (dict(dim=2, M=50, N=50), dict(matrix_free_polys=True, smooth_type="fff")),
    "adv2d_40_newton_mf": (dict(dim=2, M=40, N=40), dict(inverse_type="newton", matrix_free_polys=True)),
    "diff2d_30_neumann_mf": (dict(dim=2, M=30, N=30, u=0.0, v=0.0, alpha=1.0), dict(inverse_type="neumann", matrix_free_polys=True)),
    "adv3d_20_trunc": (dict(dim=3, M=20, N=20, P=20), dict(a_lump=True, auto_truncate_start_level=1, auto_truncate_tol=1e-1,
                                                           coarsest_inverse_type="newton", coarsest_matrix_free_polys=True, coarsest_poly_order=10)),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CYCLE_CASES))
def test_fused_cycle_is_the_reference_cycle_to_rounding(gpu, name):
    """one hierarchy uploaded with fuse_first_sweep = 2 (G and A_fc both on the device): the cycle with and without the fused
    sweep, and the C oracle's cycle (the reference's sequence on the host)"""
    import torch
    import pflare_amd as P
    from oracle import c_oracle
    gen, opts = CYCLE_CASES[name]
    lib = P.lib()
    try:
        assert lib.airhip_set_tuning(b"fuse_first_sweep", 2) == 0
        A = P.HostMat.adv_diff_fd(**gen)
        n = A.shape[0]
        pc = P.PCAIR(**opts).setup(A)
        x = _lcg(n)
        xd = torch.from_numpy(x).cuda()
        z1 = torch.empty_like(xd); z0 = torch.empty_like(xd)
        assert lib.airhip_set_tuning(b"fuse_first_sweep_run", 1) == 0
        pc.apply(xd, z1)
        assert lib.airhip_set_tuning(b"fuse_first_sweep_run", 0) == 0
        pc.apply(xd, z0)
        z1, z0 = z1.cpu().numpy(), z0.cpu().numpy()
        assert np.linalg.norm(z1 - z0) <= 1e-13 * np.linalg.norm(z0), np.linalg.norm(z1 - z0) / np.linalg.norm(z0)
        zr = c_oracle.CHierarchy(pc).vcycle(x)
        assert np.linalg.norm(z1 - zr) <= 1e-10 * np.linalg.norm(zr)
        # the profile names the fused launches where the sweep is fused
        assert lib.airhip_set_tuning(b"fuse_first_sweep_run", 1) == 0
        pc.profile(True); pc.profile_reset()
        pc.apply(xd, torch.empty_like(xd)); pc.sync()
        names = {s["name"] for s in pc.profile_stats() if s["calls"] > 0}
        pc.profile(False)
        assert any(s.startswith("fc_first(G=") for s in names), names
    finally:
        lib.airhip_set_tuning(b"fuse_first_sweep", 1)
        lib.airhip_set_tuning(b"fuse_first_sweep_run", 1)


@pytest.mark.gpu
def test_default_upload_drops_a_fc_where_nothing_reads_it_and_solves_the_same(gpu):
    """default (fuse_first_sweep = 1) against the reference's sequence (0) on separate uploads: same iteration count, same
    residual history to rounding, less device memory for the F-C blocks"""
    import torch
    import pflare_amd as P
    lib = P.lib()
    A = P.HostMat.adv_diff_fd(dim=3, M=32, N=32, P=32)
    n = A.shape[0]
    out = {}
    try:
        for mode in (1, 0):
            assert lib.airhip_set_tuning(b"fuse_first_sweep", mode) == 0
            pc = P.PCAIR(a_lump=True).setup(A)
            b = torch.zeros(n, dtype=torch.float64, device="cuda"); x = torch.ones_like(b)
            out[mode] = P.KSP("gmres", pc_side="right", rtol=1e-10, atol=1e-50, max_it=30).solve(b, x, pc=pc)
    finally:
        lib.airhip_set_tuning(b"fuse_first_sweep", 1)
    assert out[1]["its"] == out[0]["its"] and out[1]["reason"] == out[0]["reason"], out
    h1, h0 = np.array(out[1]["history"]), np.array(out[0]["history"])
    assert np.max(np.abs(h1 - h0) / h0[0]) <= 1e-10
