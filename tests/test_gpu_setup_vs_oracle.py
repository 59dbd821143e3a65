"""The device setup kernels next to the ORACLE, directly (not through the product's host code): CF markers from
csrc/device/cfsplit.hip, the fixed-sparsity GMRES polynomial from csrc/device/polyasm.hip, Z = -A_cf M and R (A P) with
drop / lump from csrc/device/spgemm.hip, the F / C blocks from csrc/device/extract.hip - each level of an oracle
hierarchy handed to the device setup in isolation (max_levels = 2, the trick of tests/test_host_setup.py: a whole hierarchy of
two implementations only agrees level by level because the CF splitting flips on 1-ulp ties).

What the oracle restates: src/PMISR_Module.F90:121-671, src/DDC_Module.F90:221-481, src/Gmres_Poly.F90:920-1337,
src/AIR_Operators_Setup.F90:786-795,863-865,1008-1069, src/PETSc_Helper.F90:336-399."""
import numpy as np
import pytest

import pflare_amd as P
from oracle import air_ref as R
from oracle import bridge

pytestmark = pytest.mark.gpu

CASES = {
    # name: (generator, product options, oracle options)
    "adv2d_48": (dict(dim=2, M=48, N=48), {}, R.AirOptions()),
    "adv3d_14_lump": (dict(dim=3, M=14, N=14, P=14), dict(a_lump=True), R.AirOptions(a_lump=True)),
    "advdiff2d_40_theta0.25": (dict(dim=2, M=40, N=40, alpha=0.05), dict(strong_threshold=0.25), R.AirOptions(strong_threshold=0.25)),
    "diff2d_36_order4_s2": (dict(dim=2, M=36, N=36, u=0, v=0, alpha=1.0), dict(poly_order=4, inverse_sparsity_order=2),
                            R.AirOptions(poly_order=4, inverse_sparsity_order=2)),
    "adv2d_40_power_adrop": (dict(dim=2, M=40, N=40), dict(inverse_type="power", a_drop=1e-3),
                             R.AirOptions(inverse_type=R.INV_POWER, a_drop=1e-3)),
}


def _same_pattern(a, b):
    a = a.tocsr(); b = b.tocsr(); a.sort_indices(); b.sort_indices()
    return a.shape == b.shape and np.array_equal(a.indptr, b.indptr) and np.array_equal(a.indices, b.indices)


def _max_rel(a, b):
    a = a.tocsr(); b = b.tocsr(); a.sort_indices(); b.sort_indices()
    den = np.maximum(np.abs(b.data), 1e-300)
    return float(np.max(np.abs(a.data - b.data) / den)) if b.nnz else 0.0


def _level_operators(A, o):
    H = R.setup_air(A, o)
    mats = [A]
    cur = A
    for L in H.levels:
        RAP = (L.R @ (cur @ L.P)).tocsr(); RAP.sort_indices()
        cur = R.remove_small(RAP, o.a_drop, rel=1, lump=o.a_lump)
        mats.append(cur)
    return H, mats


@pytest.mark.parametrize("case", sorted(CASES))
def test_device_setup_of_one_level_equals_the_oracle(gpu, case):
    gen, popts, oopts = CASES[case]
    A = R.gen_adv_diff_fd(**gen)
    H, mats = _level_operators(A, oopts)
    checked = 0
    for l, L in enumerate(H.levels[:4]):
        if mats[l].shape[0] < 64:
            break
        pc = P.PCAIR(max_levels=2, **popts)
        pc.set_option("setup_device_min_nnz", 0)          # every product, the CF splitting, the polynomial assembly and the
        pc.setup(mats[l])                                 # block extraction on the device, whatever the size
        if pc.num_levels != 2:
            continue
        e = pc.export_level(0)
        # CF splitting (device markers) = the oracle's PMISR + DDC
        assert np.array_equal(e["F"], L.F) and np.array_equal(e["C"], L.C), (case, l)
        Hp = bridge.hierarchy_from_product(pc, mats[l])
        Lp = Hp.levels[0]
        assert Lp.aff_diag == L.aff_diag
        # device extraction of the F / C blocks: exact
        assert _same_pattern(Lp.A_fc, L.A_fc) and _max_rel(Lp.A_fc, L.A_fc) == 0.0, (case, l)
        if not L.aff_diag:
            assert _same_pattern(Lp.A_ff, L.A_ff) and _max_rel(Lp.A_ff, L.A_ff) == 0.0, (case, l)
        # device polynomial assembly (M_ff), device Z = -A_cf M with r_drop, one-point W
        if L.inv_A_ff.kind == "assembled":
            assert _same_pattern(Lp.inv_A_ff.mat, L.inv_A_ff.mat) and _max_rel(Lp.inv_A_ff.mat, L.inv_A_ff.mat) <= 1e-10, (case, l)
        elif L.inv_A_ff.kind == "diag":
            assert np.allclose(Lp.inv_A_ff.d, L.inv_A_ff.d, rtol=1e-10, atol=0)
        assert _same_pattern(Lp.Z, L.Z) and _max_rel(Lp.Z, L.Z) <= 1e-10, (case, l)
        assert _same_pattern(Lp.W, L.W) and _max_rel(Lp.W, L.W) <= 1e-10, (case, l)
        # device R (A P) with a_drop / lump = the oracle's next level operator
        assert _same_pattern(Hp.coarse_A, mats[l + 1]) and _max_rel(Hp.coarse_A, mats[l + 1]) <= 1e-10, (case, l)
        checked += 1
    assert checked >= 2, (case, checked)
