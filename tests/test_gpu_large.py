"""A hierarchy whose upper levels are wider than a packed block's column range (2^19), stored in a deliberately bad
(random) row order, so that the V-cycle and GMRES run through RAW row blocks on the big levels and packed blocks on
the small ones - against oracle/air_oracle.c on the same hierarchy (the natural-order layout of the same problem is
what bench.py measures)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_vcycle_and_gmres_through_raw_blocks_3d_144(gpu):
    import torch
    import pflare_amd as P
    from oracle import c_oracle
    M = 144                                           # 2 985 984 rows
    A = P.HostMat.adv_diff_fd(dim=3, M=M, N=M, P=M)
    n = A.shape[0]
    opts = dict(a_lump=True, auto_truncate_start_level=1, auto_truncate_tol=1e-1, coarsest_inverse_type="newton",
                coarsest_matrix_free_polys=True, coarsest_poly_order=10)
    pc = P.PCAIR(**opts)
    pc.set_ordering_hint(np.random.default_rng(2).permutation(n))
    pc.setup(A)
    ch = c_oracle.CHierarchy(pc)
    i = np.arange(n, dtype=np.uint64)
    x = ((np.uint64(1103515245) * (i + np.uint64(12345))) % np.uint64(2 ** 31)).astype(np.float64) / 2.0 ** 30 - 1.0
    z = torch.empty(n, dtype=torch.float64, device="cuda")
    pc.apply(torch.from_numpy(x).cuda(), z)
    zr = ch.vcycle(x)
    assert np.linalg.norm(z.cpu().numpy() - zr) <= 1e-10 * np.linalg.norm(zr)
    b = torch.zeros(n, dtype=torch.float64, device="cuda"); sol = torch.ones_like(b)
    res = P.KSP("gmres", pc_side="right", rtol=1e-10, atol=1e-50, max_it=30).solve(b, sol, pc=pc)
    _, its, reason, rn, _ = ch.ksp(A, np.zeros(n), np.ones(n), "gmres", "right", rtol=1e-10, max_it=30)
    assert res["reason"] > 0 and res["its"] == its and res["reason"] == reason, (res, its, reason)
    # the layout changes nothing but the layout: same iteration count as the natural order
    pc2 = P.PCAIR(**opts).setup(A)
    sol2 = torch.ones_like(b)
    res2 = P.KSP("gmres", pc_side="right", rtol=1e-10, atol=1e-50, max_it=30).solve(b, sol2, pc=pc2)
    assert res2["its"] == res["its"]
    assert float((sol - sol2).abs().max()) <= 1e-9
