"""BASELINE.json configs 1 and 2 at their full sizes on the GPU (configs 3/4 are bench.py's workload):
size-independent properties the reference states for them."""
import pytest

pytestmark = pytest.mark.gpu

NEWTON_COARSE = dict(coarsest_inverse_type="newton", coarsest_matrix_free_polys=True, coarsest_poly_order=10)


def test_config1_adv_1d_1e5(gpu):
    """tests/adv_1d.c N=100000 with the options of tests/Makefile:537-540 (the reference runs n=1000 and
    requires -ksp_max_it 2): GMRES(30) right PC, rtol 1e-10, power basis, a_drop 1e-3, Newton coarse solver."""
    import torch
    import pflare_amd as P
    n = 100000
    A = P.HostMat.adv_1d(n)
    pc = P.PCAIR(a_drop=1e-3, inverse_type="power", **NEWTON_COARSE).setup(A)
    b = torch.zeros(n, dtype=torch.float64, device="cuda"); x = torch.ones_like(b)
    res = P.KSP("gmres", pc_side="right", rtol=1e-10, atol=1e-50, max_it=2).solve(b, x, pc=pc)
    assert res["reason"] > 0 and res["its"] <= 2, res
    assert float(x.abs().max()) < 1e-8          # exact solution is 0


def test_config2_adv2d_1000_truncation_keeps_the_iteration_count(gpu):
    """docs/gpus.md:34-40: 2D 1000x1000 advection, Richardson, 10th-order matrix-free Newton coarse solver; with
    -pc_air_auto_truncate_start_level 1 -pc_air_auto_truncate_tol 1e-1 the hierarchy is truncated and "this
    gives the same iteration count as without truncation"."""
    import torch
    import pflare_amd as P
    A = P.HostMat.adv_diff_fd(dim=2, M=1000, N=1000)
    n = A.shape[0]
    out = {}
    for name, extra in (("full", {}), ("truncated", dict(auto_truncate_start_level=1, auto_truncate_tol=1e-1))):
        pc = P.PCAIR(**NEWTON_COARSE, **extra).setup(A)
        b = torch.zeros(n, dtype=torch.float64, device="cuda"); x = torch.ones_like(b)
        res = P.KSP("richardson", rtol=1e-5, max_it=50).solve(b, x, pc=pc)
        assert res["reason"] > 0, (name, res)
        out[name] = (res["its"], pc.num_levels)
    assert out["full"][0] == out["truncated"][0], out
    assert out["truncated"][1] < out["full"][1], out
