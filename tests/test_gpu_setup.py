"""GPU parity of the setup products: the device SpGEMM with fused drop / lump (csrc/device/
spgemm.hip) must be BIT-IDENTICAL to the host SpGEMM (csrc/host/spops.cpp), and a hierarchy
built with the products on the device must be bit-identical to the host-only setup (which
tests/test_host_setup.py checks against the oracle).  Exactness matters: the CF splitting
of the next level flips on 1-ulp differences of the coarse operator."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _spgemm(P, A, B, drop=False, tol=0.0, lump=False, negate=False, on_device=True):
    lib = P.lib()
    fn = lib.airhip_spgemm_filtered
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    a, b = P.as_csr(A), P.as_csr(B)
    h = C.c_void_p()
    stats = (C.c_int64 * 5)()
    rc = fn(a.ref(), b.ref(), int(drop), float(tol), int(lump), int(negate), int(on_device), C.byref(h), stats)
    assert rc == 0, P.lib().airhip_last_error().decode()
    out = P.HostMat(h).csr()
    return out, [int(s) for s in stats]


def _rand(m, n, per_row, seed, diag=False):
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(m), per_row)
    cols = rng.integers(0, n, size=m * per_row)
    vals = rng.standard_normal(m * per_row) * 10.0 ** rng.integers(-4, 2, size=m * per_row)
    A = sp.csr_matrix((vals, (rows, cols)), shape=(m, n))
    if diag:
        A = A + sp.eye(m, n, format="csr") * 3.0
    A.sum_duplicates()
    A.sort_indices()
    return A.tocsr()


def _same(c, d, what=""):
    assert c.shape == d.shape, what
    assert np.array_equal(c.rowptr, d.rowptr), what
    assert np.array_equal(c.col, d.col), what
    bad = np.flatnonzero(c.val != d.val)          # bitwise
    assert bad.size == 0, (what, bad.size, c.val.size, bad[:5], c.val[bad[:5]], d.val[bad[:5]],
                           float(np.max(np.abs(c.val[bad] - d.val[bad]))))


SPGEMM_CASES = [
    # m, k, n, nnz/row A, nnz/row B, accumulator kind that must see rows (0 registers, 1-3 LDS 1k / 2k / 4k, 4 global table)
    (500, 400, 600, 5, 6, 0),
    (700, 90, 60, 8, 8, 0),             # 64 candidates on 60 columns: runs of equal columns in the register sort
    (400, 400, 500, 12, 14, 1),
    (300, 300, 300, 40, 40, 1),         # 1600 candidates on 300 columns: the sampled ratio sends the rows to the 1k tables
    (300, 3000, 40000, 40, 40, 2),      # 1600 candidates, nearly all distinct
    (200, 3000, 40000, 70, 40, 3),      # 2800
    (64, 3000, 40000, 300, 30, 4),      # 9000
    (64, 4000, 200000, 128, 30, 3),     # ~3800 distinct columns: the 4k table nearly full (no room for the sample sort's
                                        # scratch words -> bitonic network, or too many probe rounds -> global table)
]


@pytest.mark.parametrize("m,k,n,pa,pb,path", SPGEMM_CASES)
@pytest.mark.parametrize("mode", ["plain", "drop", "drop_lump", "drop_negate"])
def test_device_spgemm_bitwise(gpu, m, k, n, pa, pb, path, mode):
    import pflare_amd as P
    A = _rand(m, k, pa, 1)
    B = _rand(k, n, pb, 2, diag=(k == n))
    kw = dict(plain={}, drop=dict(drop=True, tol=1e-2), drop_lump=dict(drop=True, tol=5e-2, lump=True),
              drop_negate=dict(drop=True, tol=1e-3, negate=True))[mode]
    href, _ = _spgemm(P, A, B, on_device=False, **kw)
    dev, stats = _spgemm(P, A, B, on_device=True, **kw)
    assert stats[path] > 0, stats
    _same(dev, href)
    if mode == "plain":
        ref = (A @ B).tocsr()
        ref.sort_indices()
        assert np.array_equal(ref.indptr, dev.rowptr) and np.array_equal(ref.indices, dev.col)
        np.testing.assert_allclose(dev.val, ref.data, rtol=1e-12, atol=1e-12)


def test_device_spgemm_many_short_rows_of_b(gpu):
    """rows of A with more than 64 entries whose candidates still number <= 64 (rows of B with 0 or 1 entries) take the
    hash path; rows with exactly 64 candidates the register path"""
    import pflare_amd as P
    rng = np.random.default_rng(5)
    k = 500
    B = sp.random(k, 300, density=0.002, random_state=3, format="csr", data_rvs=rng.standard_normal)   # 0.6 per row
    B.sort_indices()
    A = _rand(200, k, 90, 7)
    for kw in ({}, dict(drop=True, tol=1e-1, lump=True)):
        href, _ = _spgemm(P, A, B, on_device=False, **kw)
        dev, stats = _spgemm(P, A, B, on_device=True, **kw)
        assert stats[1] > 0, stats
        _same(dev, href)
    A64 = _rand(300, 64, 8, 11)
    B8 = _rand(64, 1000, 8, 12)
    assert int((A64 @ abs(B8)).getnnz(axis=1).max()) <= 64
    href, _ = _spgemm(P, A64, B8, on_device=False, drop=True, tol=1e-2)
    dev, stats = _spgemm(P, A64, B8, on_device=True, drop=True, tol=1e-2)
    assert stats[0] == 300, stats
    _same(dev, href)


def test_device_spgemm_rows_that_overflow_the_planned_table_move_up(gpu):
    """the accumulator is chosen from the ratio distinct columns / candidates of a SAMPLE of rows; rows with far more
    distinct columns than the sample promised overflow the 1k table and are redone with a 2k table, those that
    overflow that with a 4k table (device lists)"""
    import pflare_amd as P
    rng = np.random.default_rng(21)
    k, n, m = 2000, 50000, 1200
    # rows 0..999 of B live on 300 columns, rows 1000..1999 anywhere
    rows = np.repeat(np.arange(k), 40)
    cols = np.where(rows < 1000, rng.integers(0, 300, size=rows.size), rng.integers(0, n, size=rows.size))
    B = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(k, n)); B.sum_duplicates(); B.sort_indices()
    # rows of A: 40 entries on the clustered rows of B; every 40th row 40 entries, every other 40th row 70 entries on the spread rows
    length = np.where(np.arange(m) % 40 == 13, 70, 40)
    arows = np.repeat(np.arange(m), length)
    spread = np.isin(arows % 40, (7, 13))
    acols = np.where(spread, rng.integers(1000, 2000, size=arows.size), rng.integers(0, 1000, size=arows.size))
    A = sp.csr_matrix((rng.standard_normal(arows.size), (arows, acols)), shape=(m, k)); A.sum_duplicates(); A.sort_indices()
    for kw in ({}, dict(drop=True, tol=1e-2, lump=True)):
        href, _ = _spgemm(P, A, B, on_device=False, **kw)
        dev, stats = _spgemm(P, A, B, on_device=True, **kw)
        # every row tried the 1k table, both kinds of spread rows moved to 2k, the long ones on to 4k
        assert stats[1] == m and stats[2] == 2 * (m // 40) and stats[3] == m // 40 and stats[4] == 0, stats
        _same(dev, href)


def test_device_spgemm_pool_exhaustion_takes_more_rounds(gpu):
    """the output pool is a bump allocator; rows that find it full are redone with a fresh pool (airhip_set_tuning
    "spgemm_pool_entries" shrinks it so that the case arises at test size), rows that overflow the 4k LDS table go to the
    global table on the way"""
    import pflare_amd as P
    lib = P.lib()
    cases = [(_rand(3000, 400, 5, 1), _rand(400, 600, 6, 2)), (_rand(600, 300, 40, 3), _rand(300, 300, 40, 4)),
             (_rand(300, 3000, 40, 7), _rand(3000, 40000, 40, 8)), (_rand(64, 3000, 300, 5), _rand(3000, 40000, 30, 6))]
    try:
        for A, B in cases:
            for kw in ({}, dict(drop=True, tol=5e-2, lump=True)):
                assert lib.airhip_set_tuning(b"spgemm_pool_entries", 0) == 0
                href, _ = _spgemm(P, A, B, on_device=False, **kw)
                assert lib.airhip_set_tuning(b"spgemm_pool_entries", max(1, href.val.size // 3)) == 0
                dev, _ = _spgemm(P, A, B, on_device=True, **kw)
                _same(dev, href)
    finally:
        lib.airhip_set_tuning(b"spgemm_pool_entries", 0)


def test_device_spgemm_single_entry_rows_of_b(gpu):
    """A * P with a one-point prolongator: no row of B has more than one entry, many entries of a row of A land on
    the same column.  The kernel then takes 64 entries of A per step and adds lane by lane (the Gustavson order)."""
    import pflare_amd as P
    rng = np.random.default_rng(17)
    for (m, k, n, per_row) in ((300, 5000, 40, 70), (200, 6000, 900, 300), (100, 9000, 3000, 2000)):
        A = _rand(m, k, per_row, 3)
        rows = np.flatnonzero(rng.random(k) < 0.9)                       # a tenth of the rows of B are empty
        B = sp.csr_matrix((rng.standard_normal(rows.size), (rows, rng.integers(0, n, size=rows.size))), shape=(k, n))
        assert int(B.getnnz(axis=1).max()) == 1
        for kw in ({}, dict(drop=True, tol=1e-2, lump=True), dict(drop=True, tol=1e-3, negate=True)):
            href, _ = _spgemm(P, A, B, on_device=False, **kw)
            dev, stats = _spgemm(P, A, B, on_device=True, **kw)
            assert sum(stats[1:4]) >= m, stats                           # the table kernels, not the register path
            _same(dev, href)


def test_device_spgemm_chunked_pool(gpu):
    """on large products a wave takes 2048 pool entries at a time and places its rows itself; forced here at test size
    (chunks of 100 entries: rows longer than a chunk, rows that straddle the end of one)"""
    import pflare_amd as P
    lib = P.lib()
    cases = [(_rand(3000, 400, 5, 1), _rand(400, 600, 6, 2)), (_rand(600, 300, 40, 3), _rand(300, 300, 40, 4)),
             (_rand(300, 3000, 40, 7), _rand(3000, 40000, 40, 8))]
    try:
        assert lib.airhip_set_tuning(b"spgemm_pool_chunk", 100) == 0
        for A, B in cases:
            for kw in ({}, dict(drop=True, tol=5e-2, lump=True)):
                href, _ = _spgemm(P, A, B, on_device=False, **kw)
                dev, _ = _spgemm(P, A, B, on_device=True, **kw)
                _same(dev, href)
    finally:
        lib.airhip_set_tuning(b"spgemm_pool_chunk", -1)


def test_device_spgemm_empty_rows_and_diagonal_creation(gpu):
    import pflare_amd as P
    # rows without entries, and rows whose diagonal only exists through lumping
    A = sp.csr_matrix(np.array([[0, 0, 0, 0], [1e-6, 0, 2.0, 0], [0, 0, 0, 0], [1.0, 1e-9, 0, 0.5]]))
    B = sp.identity(4, format="csr")
    for kw in (dict(drop=True, tol=1e-3, lump=True), dict(drop=True, tol=1e-3), {}):
        href, _ = _spgemm(P, A, B, on_device=False, **kw)
        dev, _ = _spgemm(P, A, B, on_device=True, **kw)
        _same(dev, href)


SETUP_CASES = {
    "adv2d_60": (dict(dim=2, M=60, N=60), {}),
    # options whose host routines need the level operator itself: it stays on the device after R*A*P and must come
    # over on demand (SetupAccel::materialize) - full smoothing, the strong-R threshold, lAIR, a host-only CF splitting
    "adv2d_40_full": (dict(dim=2, M=40, N=40), dict(full_smoothing_up_and_down=True)),
    "advdiff2d_40_strong_r": (dict(dim=2, M=40, N=40, alpha=0.05), dict(strong_r_threshold=0.05)),
    "adv3d_12_lair_rthr": (dict(dim=3, M=12, N=12, P=12), dict(z_type="lair", strong_r_threshold=0.01, inverse_type="power")),
    "diff2d_30_diag_dom": (dict(dim=2, M=30, N=30, u=0, v=0, alpha=1.0), dict(cf_splitting_type="diag_dom", strong_threshold=0.6)),
    "adv3d_20_lump": (dict(dim=3, M=20, N=20, P=20), dict(a_lump=True)),
    "diff2d_40": (dict(dim=2, M=40, N=40, u=0, v=0, alpha=1.0), {}),
    "advdiff2d_50_order4": (dict(dim=2, M=50, N=50, alpha=0.1), dict(poly_order=4, inverse_sparsity_order=2)),
    "adv2d_40_fcf": (dict(dim=2, M=40, N=40), dict(smooth_type="fcf", a_lump=True, a_drop=1e-5)),
    # auto truncation: the test runs on the device (Arnoldi + Newton apply), decisions and the kept solver must equal the host's
    "adv3d_24_trunc": (dict(dim=3, M=24, N=24, P=24), dict(a_lump=True, auto_truncate_start_level=1, auto_truncate_tol=1e-1,
                                                         coarsest_inverse_type="newton", coarsest_matrix_free_polys=True,
                                                         coarsest_poly_order=10)),
    "adv2d_80_trunc_1e-2": (dict(dim=2, M=80, N=80), dict(auto_truncate_start_level=2, auto_truncate_tol=1e-2,
                                                          coarsest_inverse_type="newton_no_extra", coarsest_matrix_free_polys=True,
                                                          coarsest_poly_order=8)),
}


@pytest.mark.parametrize("case", sorted(SETUP_CASES))
def test_device_setup_equals_host_setup(gpu, case):
    import pflare_amd as P
    gen, opts = SETUP_CASES[case]
    A = P.HostMat.adv_diff_fd(**gen)
    pd = P.PCAIR(**opts)
    pd.set_option("setup_device_min_nnz", 0)
    pd.setup(A)
    ph = P.PCAIR(**opts)
    ph.setup_host_only(A)
    assert pd.num_levels == ph.num_levels
    for lv in range(pd.num_levels - 1):
        d, h = pd.export_level(lv), ph.export_level(lv)
        assert np.array_equal(d["F"], h["F"]) and np.array_equal(d["C"], h["C"])
        # (A_ff, A_fc, A_cf and - with C smoothing - A_cc come from the device extraction of the resident level operator)
        for k in ("Z", "W", "A_fc", "A_cf", "A_ff", "A_cc"):
            assert (k in d) == (k in h), k
            if k in h:
                _same(d[k], h[k], f"level {lv} {k}")
    (Ad, invd), (Ah, invh) = pd.export_coarse(), ph.export_coarse()
    _same(Ad, Ah, "coarse")
    assert invd["kind"] == invh["kind"]
    for k in ("re", "im", "coeffs"):
        if k in invh:
            assert np.array_equal(invd[k], invh[k]), k


# ---------------------------------------------------------------------------------------------------------------
# CF splitting on the device (csrc/device/cfsplit.hip): markers identical to the host code, which tests/test_host_setup.py
# checks against the oracle's restatement of src/PMISR_Module.F90 / src/DDC_Module.F90
CF_CASES = [
    ("adv2d_70", dict(dim=2, M=70, N=70), {}),
    ("adv3d_24", dict(dim=3, M=24, N=24, P=24), {}),
    ("diff2d_50", dict(dim=2, M=50, N=50, u=0, v=0, alpha=1.0), {}),
    ("advdiff2d_60_theta0.25", dict(dim=2, M=60, N=60, alpha=0.05), dict(strong_threshold=0.25)),
    ("adv2d_50_symmetric", dict(dim=2, M=50, N=50), dict(symmetric=True)),               # directed graph, no S + S^T
    ("adv2d_50_theta0", dict(dim=2, M=50, N=50), dict(strong_threshold=0.0)),            # exact independent set, no DDC
    ("diff2d_50_ddc2", dict(dim=2, M=50, N=50, u=0, v=0, alpha=1.0), dict(ddc_its=2, fraction_swap=0.3)),
    ("diff2d_50_ddc_threshold", dict(dim=2, M=50, N=50, u=0, v=0, alpha=1.0), dict(fraction_swap=-0.6)),
    ("adv3d_20_luby2", dict(dim=3, M=20, N=20, P=20), dict(max_luby_steps=2)),
]


@pytest.mark.parametrize("name,gen,kw", CF_CASES, ids=[c[0] for c in CF_CASES])
def test_device_cf_splitting_equals_host(gpu, name, gen, kw):
    import pflare_amd as P
    A = P.HostMat.adv_diff_fd(**gen)
    Fh, Ch = P.compute_cf_splitting(A, **kw)
    Fd, Cd = P.compute_cf_splitting(A, on_device=True, **kw)
    assert np.array_equal(Fh, Fd) and np.array_equal(Ch, Cd)
    assert Fh.size > 0 and Ch.size > 0


def test_device_cf_splitting_on_the_coarse_operators_of_a_hierarchy_and_on_reference_data(gpu):
    """denser rows, general values: every level operator of a 3D hierarchy, and the reference's own test matrix"""
    import os
    import pflare_amd as P
    from pflare_amd.petscio import load_mat_vec
    from oracle import bridge
    A = P.HostMat.adv_diff_fd(dim=3, M=18, N=18, P=18)
    pc = P.PCAIR(a_lump=True, max_levels=6).setup_host_only(A)
    H = bridge.hierarchy_from_product(pc, bridge.to_scipy(A.csr()))
    mats = [bridge.to_scipy(A.csr())]
    cur = mats[0]
    for L in H.levels:                      # R A P of every level, recomputed with scipy (only the pattern / values matter)
        cur = (L.R @ (cur @ L.P)).tocsr(); cur.sort_indices()
        mats.append(cur)
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_data", "mat_stream_2364")
    mats.append(load_mat_vec(root)[0])
    for M in mats:
        if M.shape[0] < 8:
            continue
        Fh, Ch = P.compute_cf_splitting(M)
        Fd, Cd = P.compute_cf_splitting(M, on_device=True)
        assert np.array_equal(Fh, Fd) and np.array_equal(Ch, Cd), M.shape


# ---------------------------------------------------------------------------------------------------------------
# fixed-sparsity polynomial assembly on the device (csrc/device/polyasm.hip): bit-identical to host/poly.cpp, which
# tests/test_host_setup.py checks against the oracle's restatement of src/Gmres_Poly.F90:920-1337
POLY_CASES = [
    ("adv2d_60_arnoldi6", dict(dim=2, M=60, N=60), dict(type="arnoldi", poly_order=6)),
    ("advdiff2d_50_power4", dict(dim=2, M=50, N=50, alpha=0.1), dict(type="power", poly_order=4)),
    ("diff3d_14_arnoldi6", dict(dim=3, M=14, N=14, P=14, u=0, v=0, w=0, alpha=1.0), dict(type="arnoldi", poly_order=6)),
    ("diff2d_40_order2", dict(dim=2, M=40, N=40, u=0, v=0, alpha=1.0), dict(type="arnoldi", poly_order=2)),
    ("diff2d_40_diag_scale", dict(dim=2, M=40, N=40, u=0, v=0, alpha=1.0), dict(type="arnoldi", poly_order=6, diag_scale=True)),
]


@pytest.mark.parametrize("name,gen,kw", POLY_CASES, ids=[c[0] for c in POLY_CASES])
def test_device_polynomial_assembly_bitwise(gpu, name, gen, kw):
    import pflare_amd as P
    A = P.HostMat.adv_diff_fd(**gen)
    n = A.shape[0]
    pd = P.PCPFLAREINV(setup_device_min_nnz=0, **kw).setup(A)
    ph = P.PCPFLAREINV(**kw).setup_host_only(A)
    d, h = pd.get_inverse(n), ph.get_inverse(n)
    assert d["kind"] == h["kind"] == 1          # assembled
    _same(d["mat"], h["mat"], name)


def test_device_polynomial_assembly_on_denser_operators(gpu):
    """rows of a few hundred entries (the large bin), rows without a stored diagonal, empty rows"""
    import pflare_amd as P
    rng = np.random.default_rng(11)
    n = 3000
    A = (_rand(n, n, 30, 5) + sp.eye(n, format="csr") * 40.0).tolil()
    for i in (3, 77, 500):                      # rows without a diagonal entry
        A[i, i] = 0.0
    A[1234, :] = 0.0                            # an empty row
    A = A.tocsr(); A.eliminate_zeros(); A.sort_indices()
    dense = sp.csr_matrix(rng.standard_normal((1, n)) * (rng.random((1, n)) < 0.12))
    A = sp.vstack([A[:10], dense, A[11:]]).tocsr(); A.sort_indices()        # one row of ~360 entries
    pd = P.PCPFLAREINV(type="power", poly_order=3, setup_device_min_nnz=0).setup(A)
    ph = P.PCPFLAREINV(type="power", poly_order=3).setup_host_only(A)
    _same(pd.get_inverse(n)["mat"], ph.get_inverse(n)["mat"], "dense rows")


def test_device_setup_gives_the_host_codes_hierarchy_and_two_objects_coexist(gpu):
    """the default device-memory path of the setup (block cache + hipMalloc; the virtual-memory arenas of round 3 are opt-in
    since round 4, see test_setup_fallback_paths_in_a_process_of_their_own): the hierarchy must be the host code's, bit for
    bit, and a second object set up while the first one is alive must give the same solve."""
    import torch
    import pflare_amd as P
    A = P.HostMat.adv_diff_fd(dim=3, M=44, N=44, P=44)
    opts = dict(a_lump=True)
    pd = P.PCAIR(**opts); pd.set_option("setup_device_min_nnz", 0); pd.setup(A)
    ph = P.PCAIR(**opts); ph.setup_host_only(A)
    assert pd.num_levels == ph.num_levels
    for lv in range(pd.num_levels - 1):
        d, h = pd.export_level(lv), ph.export_level(lv)
        assert np.array_equal(d["F"], h["F"])
        for k in ("Z", "A_fc", "A_ff"):
            if k in h:
                _same(d[k], h[k], f"level {lv} {k}")
    n = A.shape[0]
    b = torch.zeros(n, dtype=torch.float64, device="cuda"); x = torch.ones_like(b)
    r = P.KSP("gmres", pc_side="right", rtol=1e-10, atol=1e-50, max_it=30).solve(b, x, pc=pd)
    assert r["reason"] > 0 and r["its"] <= 9, r
    # a second object set up while the first one still holds arena blocks: its setup takes the old path, same result
    p2 = P.PCAIR(**opts); p2.set_option("setup_device_min_nnz", 0); p2.setup(A)
    x2 = torch.ones_like(b)
    r2 = P.KSP("gmres", pc_side="right", rtol=1e-10, atol=1e-50, max_it=30).solve(b, x2, pc=p2)
    assert r2["its"] == r["its"] and float((x2 - x).abs().max()) == 0.0
    del pd, p2


@pytest.mark.parametrize("defer", ["1", "0"], ids=["deferred-downloads", "eager-downloads"])
def test_host_mirror_of_a_device_setup_is_the_host_codes_hierarchy(gpu, defer, monkeypatch):
    """capi.cpp mirror thread / device/setup_accel.cpp defer_downloads: in the plain recipe A_cf, the assembled A_ff inverse and Z
    stay on the device during the setup and a thread of the library copies them to the host afterwards; whoever reads the host
    hierarchy waits for it.  Every operator must be the host code's, bit for bit, with the downloads deferred and with
    AIRHIP_DEFER_DOWNLOADS=0 (everything copied during the setup); the solve on the device hierarchy - converted from the copies the
    setup kept on the device - must be the same in both."""
    import torch
    import pflare_amd as P
    monkeypatch.setenv("AIRHIP_DEFER_DOWNLOADS", defer)
    A = P.HostMat.adv_diff_fd(dim=3, M=40, N=40, P=40)
    opts = dict(a_lump=True)
    pd = P.PCAIR(**opts); pd.set_option("setup_device_min_nnz", 0); pd.setup(A)
    n = A.shape[0]
    b = torch.zeros(n, dtype=torch.float64, device="cuda"); x = torch.ones_like(b)
    r = P.KSP("gmres", pc_side="right", rtol=1e-10, atol=1e-50, max_it=30).solve(b, x, pc=pd)     # beside the mirror thread
    t = pd.host_mirror_wait()
    assert t >= 0.0 and (defer == "1") == (t > 0.0), t
    ph = P.PCAIR(**opts); ph.setup_host_only(A)
    assert pd.num_levels == ph.num_levels and pd.complexities()["storage"] == pytest.approx(ph.complexities()["storage"], rel=1e-14)
    for lv in range(pd.num_levels - 1):
        d, h = pd.export_level(lv), ph.export_level(lv)
        assert np.array_equal(d["F"], h["F"])
        for k in ("Z", "A_fc", "A_ff", "A_cf"):
            if k in h:
                _same(d[k], h[k], f"level {lv} {k}")
        if h["inv_ff"].get("mat") is not None:
            _same(d["inv_ff"]["mat"], h["inv_ff"]["mat"], f"level {lv} M_ff")
        if h.get("wcol") is not None:          # the one-point prolongator, read off A_fc on the device
            assert np.array_equal(d["wcol"], h["wcol"]), f"level {lv} wcol"
    assert r["reason"] > 0 and r["its"] <= 9, r
    # a second setup on the same object while the first hierarchy's mirror may still be running
    pd.setup(A)
    x2 = torch.ones_like(b)
    r2 = P.KSP("gmres", pc_side="right", rtol=1e-10, atol=1e-50, max_it=30).solve(b, x2, pc=pd)
    assert r2["its"] == r["its"] and float((x2 - x).abs().max()) == 0.0


FALLBACK_SCRIPT = r"""
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pflare_amd as P
P.lib().airhip_set_device(0)
M = int(sys.argv[1])
A = P.HostMat.adv_diff_fd(dim=3, M=M, N=M, P=M)
n = A.shape[0]
pc = P.PCAIR(a_lump=True).setup(A)
b = torch.zeros(n, dtype=torch.float64, device="cuda"); x = torch.ones_like(b)
r = P.KSP("gmres", pc_side="right", rtol=1e-10, atol=1e-50, max_it=40).solve(b, x, pc=pc)
t = pc.host_mirror_wait()
# a second setup of the same object with host allocations in between and the first hierarchy's arrays just released
junk = [np.ones(1 << 22) for _ in range(8)]
pc.setup(A)
x2 = torch.ones_like(b)
r2 = P.KSP("gmres", pc_side="right", rtol=1e-10, atol=1e-50, max_it=40).solve(b, x2, pc=pc)
pc.host_mirror_wait()
sizes = [pc.level_sizes(l)["nnz_A"] for l in range(pc.num_levels - 1)]
print("RESULT " + json.dumps(dict(levels=pc.num_levels, its=r["its"], its2=r2["its"], rnorm=r["rnorm"], rnorm2=r2["rnorm"],
                                  same=float((x2 - x).abs().max()), nnz=sizes, cx=pc.complexities()["storage"])))
"""


def _run_fallback(env_extra, M=96):
    import json
    import os
    import sys
    from proc_util import run
    env = dict(os.environ); env.update(env_extra)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = run([sys.executable, "-c", FALLBACK_SCRIPT, str(M)], timeout=420, cwd=root, env=env)
    assert r.returncode == 0, (env_extra, r.stdout[-2000:], r.stderr[-4000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_setup_fallback_paths_in_a_process_of_their_own(gpu):
    """The switches that stay are read once per process, so each runs in a child: R / P assembled on the host
    (AIRHIP_DEVICE_RP=0), no operator kept on the device for the upload (AIRHIP_KEEP_DEVICE_COPIES=0), downloads not deferred
    to the mirror thread (AIRHIP_DEFER_DOWNLOADS=0).  (The virtual-memory arenas of rounds 3-4 are gone: round 5 removed them.)  3D advection 96^3 (3.5 M non-zeros: every device path of the setup is taken,
    the mirror thread runs), two setups back to back in each process with the host mirror alive and host allocations in
    between.  Every variant must build the same hierarchy and solve it the same way as the default path."""
    base = _run_fallback({})
    assert base["its"] == base["its2"] and base["same"] == 0.0 and base["its"] <= 9, base
    for env in ({"AIRHIP_DEVICE_RP": "0"}, {"AIRHIP_KEEP_DEVICE_COPIES": "0"}, {"AIRHIP_DEFER_DOWNLOADS": "0"}):
        v = _run_fallback(env)
        assert v["levels"] == base["levels"] and v["nnz"] == base["nnz"] and v["cx"] == base["cx"], (env, v, base)
        assert v["its"] == base["its"] == v["its2"] and v["same"] == 0.0, (env, v, base)
        assert v["rnorm"] == base["rnorm"], (env, v["rnorm"], base["rnorm"])


def test_two_setups_back_to_back_at_128_with_the_mirror_thread_alive(gpu):
    """two setups back to back with the host-mirror thread alive and host allocations in between at
    about 128^3, on the default device-memory path (no address-space covers, no handle recycling)."""
    v = _run_fallback({}, M=128)
    assert v["its"] == v["its2"] and v["same"] == 0.0 and v["its"] <= 10, v


def test_graph_capture_right_after_a_setup_waits_for_the_mirror_thread(gpu):
    """found in round 4: a PCApply recorded into a hipGraph while the library's host-mirror thread was still copying and
    freeing device memory failed ("operation failed due to a previous error during capture").  The capture now waits for that
    thread; the captured cycle must give the plain cycle's result bit for bit."""
    import torch
    import pflare_amd as P
    A = P.HostMat.adv_diff_fd(dim=3, M=72, N=72, P=72)
    n = A.shape[0]
    pc = P.PCAIR(a_lump=True).setup(A)
    pc.use_graph(True)                         # (no host_mirror_wait: the mirror thread of the 72^3 setup is still busy)
    x = torch.ones(n, dtype=torch.float64, device="cuda")
    yg = torch.empty_like(x); pc.apply(x, yg); pc.apply(x, yg)
    pc.use_graph(False)
    yp = torch.empty_like(x); pc.apply(x, yp)
    assert torch.equal(yg, yp)
    assert pc.host_mirror_wait() >= 0.0
