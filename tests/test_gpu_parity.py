"""GPU parity: the HIP solve phase (through the C ABI) against the CPU oracle on the SAME
hierarchy (exported from the product setup, or built by the oracle and injected through the
bring-your-own ABI).  Tolerances follow the reference's CPU-vs-device differential test
(PFLARE_TOL_MATFREE_13 = 1e-13 for data movement / single SpMVs, 4e-11 for accumulated fp;
src/Pflare_Parameters.F90:209-210)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_SPMV = 1e-13
TOL_CYCLE = 1e-10


def lcg_vector(n):
    """SURVEY 8(d) microbenchmark vector: x_i = ((1103515245*(i+12345)) mod 2^31)/2^30 - 1"""
    i = np.arange(n, dtype=np.uint64)
    return (((np.uint64(1103515245) * (i + np.uint64(12345))) % np.uint64(2 ** 31)).astype(np.float64)
            / float(2 ** 30) - 1.0)


def relerr(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


@pytest.fixture(scope="module")
def torch_mod(gpu):
    import torch
    return torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


CASES = {
    "adv2d_25": (dict(dim=2, M=25, N=25), {}),
    "adv2d_40_lump": (dict(dim=2, M=40, N=40), dict(a_lump=True, a_drop=1e-5, strong_threshold=0.99)),
    "adv3d_12_lump": (dict(dim=3, M=12, N=12, P=12), dict(a_lump=True)),
    "diff2d_16": (dict(dim=2, M=16, N=16, u=0, v=0, alpha=1.0), {}),
    "diff2d_16_neumann": (dict(dim=2, M=16, N=16, u=0, v=0, alpha=1.0), dict(inverse_type="neumann", a_drop=1e-3)),
    "diff2d_16_neumann_mf": (dict(dim=2, M=16, N=16, u=0, v=0, alpha=1.0),
                             dict(inverse_type="neumann", a_drop=1e-3, matrix_free_polys=True)),
    "adv2d_30_mf": (dict(dim=2, M=30, N=30), dict(matrix_free_polys=True)),
    # BASELINE config 5 (SURVEY 8d substitute): anisotropic 3D FD diffusion, Neumann-polynomial smoother
    "aniso3d_14_neumann": (dict(dim=3, M=14, N=14, P=14, u=0, v=0, w=0, alpha=1.0, L_z=0.01),
                           dict(inverse_type="neumann", a_drop=1e-3)),
    "aniso3d_14_neumann_mf": (dict(dim=3, M=14, N=14, P=14, u=0, v=0, w=0, alpha=1.0, L_z=0.01),
                              dict(inverse_type="neumann", a_drop=1e-3, matrix_free_polys=True)),
    "adv2d_30_newton_coarse": (dict(dim=2, M=30, N=30),
                               dict(coarsest_inverse_type="newton", coarsest_poly_order=10,
                                    coarsest_matrix_free_polys=True, inverse_type="power", a_drop=1e-3,
                                    max_levels=4)),
    # lAIR restrictor with C smoothing (tests/Makefile:638-642), SAI smoother (tests/Makefile:405-406), Newton smoothers
    "adv2d_30_lair_ffc": (dict(dim=2, M=30, N=30), dict(a_lump=True, smooth_type="ffc", z_type="lair", inverse_type="wjacobi")),
    "adv3d_10_lair_rthr": (dict(dim=3, M=10, N=10, P=10), dict(z_type="lair", strong_r_threshold=0.01, inverse_type="power")),
    "diff2d_16_sai": (dict(dim=2, M=16, N=16, u=0, v=0, alpha=1.0), dict(inverse_type="sai", a_drop=1e-3)),
    "adv2d_24_newton": (dict(dim=2, M=24, N=24), dict(inverse_type="newton", poly_order=3)),
    # -pc_air_full_smoothing_up_and_down: multiplicative V(1,1) cycle on all points (src/AIR_MG_Setup.F90:1014-1079)
    "adv2d_24_full": (dict(dim=2, M=24, N=24), dict(full_smoothing_up_and_down=True)),
    "adv3d_10_full_mf": (dict(dim=3, M=10, N=10, P=10), dict(full_smoothing_up_and_down=True, matrix_free_polys=True, a_lump=True)),
    "diff2d_16_full_ideal_W": (dict(dim=2, M=16, N=16, u=0, v=0, alpha=1.0),
                               dict(full_smoothing_up_and_down=True, one_point_classical_prolong=False, inverse_type="neumann")),
    "adv2d_16_ideal_W": (dict(dim=2, M=16, N=16), dict(one_point_classical_prolong=False)),
    "adv2d_16_fc": (dict(dim=2, M=16, N=16), dict(smooth_type="fc")),
    "adv2d_16_fcf": (dict(dim=2, M=16, N=16), dict(smooth_type="fcf")),
    "diff2d_12_diagscale": (dict(dim=2, M=12, N=12, u=0, v=0, alpha=1.0), dict(diag_scale_polys=True)),
    "diff2d_12_jacobi_exact": (dict(dim=2, M=8, N=8, u=0, v=0, alpha=1.0),
                               dict(strong_threshold=0.0, a_drop=0.0, r_drop=0.0, inverse_type="jacobi",
                                    mg_coarse_ksp_max_it=10)),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_spmv_and_vcycle_match_oracle(torch_mod, case):
    torch = torch_mod
    import pflare_amd as P
    from oracle import air_ref as R, bridge
    gen, opts = CASES[case]
    hm = P.HostMat.adv_diff_fd(**gen)
    A = bridge.to_scipy(hm.csr())
    n = A.shape[0]
    x = lcg_vector(n)

    # K1: device SpMV vs oracle CSR SpMV
    dm = P.DeviceMat(A)
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    dm.mult(dev(torch, x), y)
    assert relerr(y.cpu().numpy(), A @ x) <= TOL_SPMV

    pc = P.PCAIR(**opts)
    pc.setup(A)
    H = bridge.hierarchy_from_product(pc, A)
    H.coarse_richardson_its = int(opts.get("mg_coarse_ksp_max_it", 0))
    z = torch.empty(n, dtype=torch.float64, device="cuda")
    pc.apply(dev(torch, x), z)
    z_ref = R.vcycle(H, x)
    assert relerr(z.cpu().numpy(), z_ref) <= TOL_CYCLE, case
    # second application must give the same answer (no state carried between applies)
    z2 = torch.empty_like(z)
    pc.apply(dev(torch, x), z2)
    assert torch.equal(z, z2)
    # graph replay gives bitwise the same as eager launches
    pc.use_graph(True)
    z3 = torch.empty_like(z)
    pc.apply(dev(torch, x), z3)
    pc.apply(dev(torch, x), z3)
    assert torch.equal(z, z3)


KSP_CASES = [
    ("adv2d_25", "gmres", "left", "preconditioned", 1e-5, 5),          # tests/Makefile:1323  (<= 5)
    ("adv3d_12_lump", "gmres", "right", "preconditioned", 1e-10, 6),   # tests/Makefile:543-545 shape
    ("adv2d_40_lump", "gmres", "right", "preconditioned", 1e-10, 6),   # tests/Makefile:1121-1133
    ("diff2d_16_neumann", "gmres", "left", "preconditioned", 1e-5, 6),
    ("diff2d_16_neumann_mf", "gmres", "left", "preconditioned", 1e-5, 6),
    ("adv2d_30_newton_coarse", "richardson", "left", "preconditioned", 1e-5, 8),
    ("diff2d_12_jacobi_exact", "richardson", "left", "unpreconditioned", 1e-5, 1),  # tests/Makefile:421-424
    ("adv2d_16_fc", "gmres", "left", "preconditioned", 1e-5, 3),       # tests/Makefile:300 (<= 3)
    ("adv2d_30_lair_ffc", "gmres", "right", "preconditioned", 1e-10, 10),   # tests/Makefile:638-642 shape
    ("diff2d_16_sai", "gmres", "left", "preconditioned", 1e-5, 6),
    ("adv3d_10_full_mf", "gmres", "right", "preconditioned", 1e-10, 12),
    ("diff2d_16_full_ideal_W", "gmres", "left", "preconditioned", 1e-5, 8),
]


@pytest.mark.parametrize("case,ktype,side,norm,rtol,bound", KSP_CASES)
def test_ksp_iteration_counts_match_oracle(torch_mod, case, ktype, side, norm, rtol, bound):
    torch = torch_mod
    import pflare_amd as P
    from oracle import air_ref as R, bridge
    gen, opts = CASES[case]
    A = bridge.to_scipy(P.HostMat.adv_diff_fd(**gen).csr())
    n = A.shape[0]
    pc = P.PCAIR(**opts).setup(A)
    H = bridge.hierarchy_from_product(pc, A)
    H.coarse_richardson_its = int(opts.get("mg_coarse_ksp_max_it", 0))
    b = np.zeros(n); x0 = np.ones(n)            # the reference drivers' problem: b = 0, x0 = 1
    pcf = lambda r: R.vcycle(H, r)
    if ktype == "gmres":
        xr, its_r, reason_r, rn_r = R.ksp_gmres(A, pcf, b, x0, rtol=rtol, max_it=bound, pc_side=side)
    else:
        xr, its_r, reason_r, rn_r = R.ksp_richardson(A, pcf, b, x0, rtol=rtol, max_it=bound, norm_type=norm)
    ksp = P.KSP(ksp_type=ktype, pc_side=side, norm_type=norm, rtol=rtol, max_it=bound)
    xd = dev(torch, x0)
    res = ksp.solve(dev(torch, b), xd, pc=pc)
    assert res["reason"] > 0, (case, res)
    assert res["its"] <= bound
    assert res["its"] == its_r and res["reason"] == reason_r, (res, its_r, reason_r)
    # final residual norms agree to 2 significant digits (SURVEY 8d parity gate)
    assert abs(res["rnorm"] - rn_r) <= 0.05 * max(rn_r, 1e-300) or max(res["rnorm"], rn_r) < 1e-13
    assert np.linalg.norm(xd.cpu().numpy() - xr) <= 1e-8 * max(1.0, np.linalg.norm(x0))


def test_oracle_built_hierarchy_through_byo_abi(torch_mod):
    """Hierarchy built by the ORACLE, injected through airhip_pcair_set_level: the HIP V-cycle
    must reproduce the oracle's V-cycle on it."""
    torch = torch_mod
    import pflare_amd as P
    from oracle import air_ref as R, bridge
    A = R.gen_adv_diff_fd(2, 20, 20)
    H = R.setup_air(A, R.AirOptions())
    levels, cA, cinv = bridge.levels_for_product(H)
    pc = P.PCAIR()
    pc.set_hierarchy(A, levels, cA, cinv)
    x = lcg_vector(A.shape[0])
    z = torch.empty(A.shape[0], dtype=torch.float64, device="cuda")
    pc.apply(dev(torch, x), z)
    assert relerr(z.cpu().numpy(), R.vcycle(H, x)) <= TOL_CYCLE


@pytest.mark.parametrize("itype,mf,order", [("arnoldi", False, 6), ("power", False, 6), ("neumann", False, 6),
                                            ("jacobi", False, 6), ("newton", True, 10), ("arnoldi", True, 6)])
def test_pcpflareinv_apply(torch_mod, itype, mf, order):
    torch = torch_mod
    import pflare_amd as P
    from oracle import air_ref as R, bridge
    A = R.gen_adv_diff_fd(2, 12, 12, u=0, v=0, alpha=1.0)
    n = A.shape[0]
    pc = P.PCPFLAREINV(type=itype, poly_order=order, matrix_free=mf).setup(A)
    d = pc.get_inverse(n)
    op = bridge._inv_from_dict(d, A)
    x = lcg_vector(n)
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    pc.apply(dev(torch, x), y)
    assert relerr(y.cpu().numpy(), R.apply_inv(op, x)) <= 4e-11


@pytest.mark.parametrize("case", ["adv3d_12_lump", "adv2d_30_newton_coarse", "adv2d_16_fcf", "diff2d_12_diagscale"])
def test_ordering_hint_changes_layout_not_results(torch_mod, case):
    """airhip_pcair_set_ordering_hint: any key vector (here a random shuffle, the worst case) must leave
    PCApply and the Krylov iteration counts unchanged - it only moves rows around inside the device."""
    torch = torch_mod
    import pflare_amd as P
    gen, opts = CASES[case]
    A = P.HostMat.adv_diff_fd(**gen)
    n = A.shape[0]
    x = dev(torch, lcg_vector(n))
    pc0 = P.PCAIR(**opts).setup(A)
    z0 = torch.empty_like(x); pc0.apply(x, z0)
    pc1 = P.PCAIR(**opts)
    pc1.set_ordering_hint(np.random.default_rng(5).permutation(n))
    pc1.setup(A)
    z1 = torch.empty_like(x); pc1.apply(x, z1)
    assert relerr(z1.cpu().numpy(), z0.cpu().numpy()) <= 1e-12
    ksp = P.KSP("gmres", pc_side="right", rtol=1e-10, atol=1e-50)
    b = torch.zeros_like(x)
    x0 = torch.ones_like(x); r0 = ksp.solve(b, x0, pc=pc0)
    x1 = torch.ones_like(x); r1 = ksp.solve(b, x1, pc=pc1)
    assert r0["its"] == r1["its"] and r0["reason"] == r1["reason"]


@pytest.mark.parametrize("case", ["adv3d_12_lump", "adv2d_16_fcf", "adv2d_24_full", "adv3d_10_lair_rthr", "diff2d_16_neumann_mf"])
@pytest.mark.parametrize("hint", [False, True])
def test_spatial_block_order_is_bitwise_neutral(torch_mod, case, hint):
    """Operators whose rows run through several sections of a hierarchical vector (R = [Z I], the outer operator,
    A_cf / A_cc, the level operator of full smoothing) are swept in the spatial order of their row blocks
    (DevMat::desc_ord; only from 4096 blocks on by default - forced here).  The order of the blocks must not change a
    bit of PCApply, with and without a layout hint, and the V-cycle must still match the oracle."""
    torch = torch_mod
    import pflare_amd as P
    from oracle import air_ref as R, bridge
    gen, opts = CASES[case]
    hm = P.HostMat.adv_diff_fd(**gen)
    A = bridge.to_scipy(hm.csr())
    n = A.shape[0]
    x = lcg_vector(n)
    key = np.random.default_rng(9).permutation(n) if hint else None
    out = []
    lib = P.lib()
    try:
        for min_blocks in (1 << 30, 1):              # never / always
            assert lib.airhip_set_tuning(b"block_order_min_blocks", min_blocks) == 0
            pc = P.PCAIR(**opts)
            if key is not None:
                pc.set_ordering_hint(key)
            pc.setup(A)
            z = torch.empty(n, dtype=torch.float64, device="cuda")
            pc.apply(dev(torch, x), z)
            b = torch.zeros(n, dtype=torch.float64, device="cuda"); sol = torch.ones_like(b)
            res = P.KSP("gmres", pc_side="right", rtol=1e-10, atol=1e-50, max_it=40).solve(b, sol, pc=pc)
            out.append((z, res["its"], res["rnorm"], pc))
    finally:
        lib.airhip_set_tuning(b"block_order_min_blocks", 4096)
    assert torch.equal(out[0][0], out[1][0])
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]
    H = bridge.hierarchy_from_product(out[1][3], A)
    assert relerr(out[1][0].cpu().numpy(), R.vcycle(H, x)) <= TOL_CYCLE


@pytest.mark.parametrize("case", ["adv3d_12_lump", "adv2d_16_fcf", "adv2d_30_lair_ffc", "adv2d_16_ideal_W", "diff2d_16_neumann_mf"])
@pytest.mark.parametrize("hint", [False, True])
def test_spatial_copy_of_x_c_is_bitwise_neutral(torch_mod, case, hint):
    """A_fc and the one-point prolongator gather from a copy of x_c in the spatial order of the level's C points
    (DevLevel::xs, tuning key fc_spatial_x: 0 off, 1 levels of >= 4096 C points - the default -, 2 every level).  Only the
    numbering of A_fc's columns changes, not the order of the entries inside a row: PCApply and the Krylov solve must not
    change by a bit, with and without a layout hint, for one-point and general W and for cycles with C smoothing."""
    torch = torch_mod
    import pflare_amd as P
    from oracle import air_ref as R, bridge
    gen, opts = CASES[case]
    hm = P.HostMat.adv_diff_fd(**gen)
    A = bridge.to_scipy(hm.csr())
    n = A.shape[0]
    x = lcg_vector(n)
    key = np.random.default_rng(9).permutation(n) if hint else None
    out = []
    lib = P.lib()
    try:
        for mode in (0, 2):
            assert lib.airhip_set_tuning(b"fc_spatial_x", mode) == 0
            pc = P.PCAIR(**opts)
            if key is not None:
                pc.set_ordering_hint(key)
            pc.setup(A)
            z = torch.empty(n, dtype=torch.float64, device="cuda")
            pc.apply(dev(torch, x), z)
            b = torch.zeros(n, dtype=torch.float64, device="cuda"); sol = torch.ones_like(b)
            res = P.KSP("gmres", pc_side="right", rtol=1e-10, atol=1e-50, max_it=40).solve(b, sol, pc=pc)
            out.append((z, res["its"], res["rnorm"], pc))
    finally:
        lib.airhip_set_tuning(b"fc_spatial_x", 1)
    assert torch.equal(out[0][0], out[1][0])
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]
    H = bridge.hierarchy_from_product(out[1][3], A)
    assert relerr(out[1][0].cpu().numpy(), R.vcycle(H, x)) <= TOL_CYCLE


@pytest.mark.parametrize("restart,n_side", [(50, 48), (30, 48), (7, 30)])
def test_unpreconditioned_gmres_with_long_restarts_matches_oracle(torch_mod, restart, n_side):
    """the one-pass multi-dot / multi-axpy kernels over more basis vectors than one launch takes (32) and over odd basis
    sizes: GMRES(restart) without a preconditioner on a 2D advection-diffusion operator needs many iterations; the
    iteration count, the convergence reason and the residual history must be the oracle's (KSPGMRES semantics)"""
    torch = torch_mod
    import pflare_amd as P
    from oracle import air_ref as R, bridge
    hm = P.HostMat.adv_diff_fd(dim=2, M=n_side, N=n_side, alpha=0.05)
    A = bridge.to_scipy(hm.csr())
    n = A.shape[0]
    b = lcg_vector(n)
    x0 = np.zeros(n)
    hist = []
    xr, its_r, reason_r, rn_r = R.ksp_gmres(A, lambda v: v, b, x0, rtol=1e-8, max_it=400, restart=restart, pc_side="right", history=hist)
    dm = P.DeviceMat(A)
    xd = torch.zeros(n, dtype=torch.float64, device="cuda")
    res = P.KSP("gmres", pc_side="right", rtol=1e-8, atol=1e-50, max_it=400, restart=restart).solve(dev(torch, b), xd, A=dm)
    # identical iteration counts over hundreds of iterations and several restart cycles (measured: 218 / 233 / 124); the
    # residual ESTIMATES of classical Gram-Schmidt without refinement differ by up to 1e-4 relative where they plateau, the
    # solutions agree and the true residual meets the tolerance
    assert res["its"] == its_r and res["reason"] == reason_r, (res["its"], its_r, res["reason"], reason_r)
    assert its_r > restart
    xg = xd.cpu().numpy()
    assert np.linalg.norm(b - A @ xg) <= 1.2e-8 * np.linalg.norm(b)
    assert relerr(xd.cpu().numpy(), xr) <= 1e-5
