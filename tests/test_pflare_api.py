"""The reference's Python option API (python/pflare.py, exercised by its python/ex_pcair_options.py and
python/ex_pcpflareinv_options.py): same function names, argument order and enum values on this package's PC
objects (pflare_amd/pflare.py).  The set / get round trips below follow the reference's example script."""
import numpy as np
import pytest

import pflare_amd as P
from pflare_amd import pflare
from oracle import air_ref as R


def test_enum_values_match_the_reference_header():
    # include/pflare.h / src/Pflare_Parameters.F90:16-32, src/CF_Splitting.F90:19-25
    assert [int(v) for v in pflare.PCPFLAREINVType] == list(range(9)) and pflare.PFLAREINV_JACOBI == 8
    assert pflare.AIR_Z_PRODUCT == 0 and pflare.AIR_Z_LAIR == 1 and pflare.AIR_Z_LAIR_SAI == 2
    assert pflare.CF_PMISR_DDC == 0 and pflare.CF_DIAG_DOM == 1 and pflare.CF_CR == 6


def test_pcair_option_round_trips_as_in_the_reference_example():
    pc = P.PCAIR()
    # defaults of the reference (src/AIR_Data_Type.F90:34-264)
    assert pflare.pcair_get_z_type(pc) == pflare.AIR_Z_PRODUCT and pflare.pcair_get_poly_order(pc) == 6
    assert pflare.pcair_get_smooth_type(pc) == "ff" and pflare.pcair_get_strong_threshold(pc) == 0.5
    pflare.pcair_set_z_type(pc, pflare.AIR_Z_LAIR)
    pflare.pcair_set_inverse_type(pc, pflare.PFLAREINV_WJACOBI)
    pflare.pcair_set_smooth_type(pc, "fcf")
    pflare.pcair_set_poly_order(pc, 4)
    pflare.pcair_set_strong_threshold(pc, 0.25)
    assert pflare.pcair_get_z_type(pc) == pflare.AIR_Z_LAIR
    assert pflare.pcair_get_inverse_type(pc) == pflare.PFLAREINV_WJACOBI
    assert pflare.pcair_get_smooth_type(pc) == "fcf" and pflare.pcair_get_poly_order(pc) == 4
    assert pflare.pcair_get_strong_threshold(pc) == 0.25
    for name, val in (("max_levels", 10), ("coarse_eq_limit", 12), ("ddc_its", 3), ("ddc_fraction", 0.2), ("r_drop", 0.05),
                      ("a_drop", 0.001), ("lair_distance", 1), ("coarsest_poly_order", 8), ("matrix_free_polys", True),
                      ("matrix_free_polys", False), ("coarsest_inverse_type", pflare.PFLAREINV_ARNOLDI)):
        getattr(pflare, f"pcair_set_{name}")(pc, val)
        assert getattr(pflare, f"pcair_get_{name}")(pc) == val, name
    # the options take effect: the hierarchy is the one the oracle builds with the same settings
    A = R.gen_adv_diff_fd(2, 14, 14)
    pc.setup_host_only(A)
    Ho = R.setup_air(A, R.AirOptions(z_type="lair", inverse_type=R.INV_WJACOBI, smooth_type="fcf", poly_order=4,
                                     strong_threshold=0.25, max_levels=10, coarse_eq_limit=12, ddc_its=3, ddc_fraction=0.2,
                                     r_drop=0.05, a_drop=0.001, lair_distance=1, coarsest_poly_order=8))
    assert pflare.pcair_get_num_levels(pc) == Ho.no_levels
    assert np.array_equal(pc.export_level(0)["F"], Ho.levels[0].F)
    assert pflare.pcair_get_grid_complexity(pc) > 1.0 and pflare.pcair_get_operator_complexity(pc) > 1.0
    assert pflare.pcair_get_cycle_complexity(pc) > 0.0 and pflare.pcair_get_storage_complexity(pc) > 0.0
    # options outside the scope are refused when switched on, accepted at their default
    pflare.pcair_set_reuse_sparsity(pc, False)
    with pytest.raises(P.AirhipError):
        pflare.pcair_set_reuse_sparsity(pc, True)
    with pytest.raises(P.AirhipError):
        pflare.pcair_set_cf_splitting_type(pc, pflare.CF_AGG)
    assert pflare.pcair_get_cf_splitting_type(pc) == pflare.CF_PMISR_DDC      # a refused value is not recorded


def test_pcpflareinv_option_round_trips():
    pc = P.PCPFLAREINV()
    assert pflare.pcpflareinv_get_type(pc) == pflare.PFLAREINV_ARNOLDI and pflare.pcpflareinv_get_poly_order(pc) == 6
    pflare.pcpflareinv_set_type(pc, pflare.PFLAREINV_POWER)
    pflare.pcpflareinv_set_poly_order(pc, 3)
    pflare.pcpflareinv_set_sparsity_order(pc, 2)
    pflare.pcpflareinv_set_matrix_free(pc, False)
    assert pflare.pcpflareinv_get_type(pc) == pflare.PFLAREINV_POWER and pflare.pcpflareinv_get_poly_order(pc) == 3
    assert pflare.pcpflareinv_get_sparsity_order(pc) == 2 and pflare.pcpflareinv_get_matrix_free(pc) is False
    A = R.gen_adv_diff_fd(2, 10, 10, u=0, v=0, alpha=1.0)
    pc.setup_host_only(A)
    M = pflare.pcpflareinv_get_inverse_mat(pc, A.shape[0])
    op, _ = R.build_inverse(A, R.INV_POWER, 3, 2, False, False)
    assert M.shape == A.shape and M.nnz == op.mat.nnz
