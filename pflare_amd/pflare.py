"""The reference's Python option API on this package's PC objects.

The reference exposes `pflare.pcair_set_<option>(pc, value)` / `pflare.pcair_get_<option>(pc)` and the same for
`pcpflareinv_*` on petsc4py PC objects (python/pflare.py:55-190), with integer enums whose values match
include/pflare.h.  This module offers the same function names, argument order and enum values on `pflare_amd.PCAIR`
and `pflare_amd.PCPFLAREINV`, so that a script written against the reference's module needs only its PC objects
replaced.  Options outside this build's scope raise `AirhipError` when set to a non-default value, exactly as
`PCAIR.set_option` does; getters return the value last set (or the reference's default, src/AIR_Data_Type.F90:34-264).
"""
from enum import IntEnum

from .pc import PCAIR, PCPFLAREINV, compute_cf_splitting, compute_diag_dom_submatrix  # noqa: F401


class CFSplittingType(IntEnum):
    CF_PMISR_DDC = 0
    CF_DIAG_DOM = 1
    CF_PMIS = 2
    CF_PMIS_DIST2 = 3
    CF_AGG = 4
    CF_PMIS_AGG = 5
    CF_CR = 6


class PCPFLAREINVType(IntEnum):
    PFLAREINV_POWER = 0
    PFLAREINV_ARNOLDI = 1
    PFLAREINV_NEWTON = 2
    PFLAREINV_NEWTON_NO_EXTRA = 3
    PFLAREINV_NEUMANN = 4
    PFLAREINV_SAI = 5
    PFLAREINV_ISAI = 6
    PFLAREINV_WJACOBI = 7
    PFLAREINV_JACOBI = 8


class PCAIRZType(IntEnum):
    AIR_Z_PRODUCT = 0
    AIR_Z_LAIR = 1
    AIR_Z_LAIR_SAI = 2


for _e in (CFSplittingType, PCPFLAREINVType, PCAIRZType):
    for _m in _e:
        globals()[_m.name] = _m
del _e, _m

_INV_NAME = {0: "power", 1: "arnoldi", 2: "newton", 3: "newton_no_extra", 4: "neumann", 5: "sai", 6: "isai",
             7: "wjacobi", 8: "jacobi"}
_CF_NAME = {0: "pmisr_ddc", 1: "diag_dom", 2: "pmis", 3: "pmis_dist2", 4: "agg", 5: "pmis_agg", 6: "cr"}
_Z_NAME = {0: "product", 1: "lair", 2: "lair_sai"}

# option -> (default of the reference, translation of the value handed to set_option)
_f32 = lambda v: float(__import__("numpy").float32(v))   # the reference's single-precision literals, widened
_AIR = {
    "print_stats_timings": (False, None), "max_levels": (300, None), "coarse_eq_limit": (6, None),
    "auto_truncate_start_level": (-1, None), "auto_truncate_tol": (_f32(1e-14), None),
    "processor_agglom": (True, None), "subcomm": (False, None), "coarsest_subcomm": (False, None),
    "strong_threshold": (0.5, None), "ddc_its": (1, None), "ddc_fraction": (_f32(0.1), None),
    "cf_splitting_type": (CFSplittingType.CF_PMISR_DDC, _CF_NAME), "max_luby_steps": (-1, None),
    "smooth_type": ("ff", None), "diag_scale_polys": (False, None), "matrix_free_polys": (False, None),
    "one_point_classical_prolong": (True, None), "full_smoothing_up_and_down": (False, None), "symmetric": (False, None),
    "constrain_w": (False, None), "constrain_z": (False, None), "improve_w_its": (0, None), "improve_z_its": (0, None),
    "strong_r_threshold": (0.0, None), "inverse_type": (PCPFLAREINVType.PFLAREINV_ARNOLDI, _INV_NAME),
    "c_inverse_type": (PCPFLAREINVType.PFLAREINV_ARNOLDI, _INV_NAME), "z_type": (PCAIRZType.AIR_Z_PRODUCT, _Z_NAME),
    "lair_distance": (2, None), "poly_order": (6, None), "inverse_sparsity_order": (1, None), "c_poly_order": (6, None),
    "c_inverse_sparsity_order": (1, None), "coarsest_inverse_type": (PCPFLAREINVType.PFLAREINV_ARNOLDI, _INV_NAME),
    "coarsest_poly_order": (6, None), "coarsest_inverse_sparsity_order": (1, None),
    "coarsest_matrix_free_polys": (False, None), "coarsest_diag_scale_polys": (False, None),
    "r_drop": (_f32(0.01), None), "a_drop": (_f32(1e-4), None), "a_lump": (False, None), "reuse_sparsity": (False, None),
}
_INV = {"poly_order": (6, None), "sparsity_order": (1, None), "type": (PCPFLAREINVType.PFLAREINV_ARNOLDI, _INV_NAME),
        "matrix_free": (False, None)}


def _make(prefix, table, cls):
    def setter(name, default, names):
        def f(pc, value):
            assert isinstance(pc, cls), f"{prefix}_set_{name} needs a {cls.__name__}"
            pc.set_option(name, names[int(value)] if names is not None else value)
            pc.__dict__.setdefault("_pflare_opts", {})[name] = type(default)(value) if not isinstance(default, IntEnum) \
                else type(default)(int(value))
        f.__name__ = f"{prefix}_set_{name}"
        return f

    def getter(name, default):
        def f(pc):
            return pc.__dict__.get("_pflare_opts", {}).get(name, default)
        f.__name__ = f"{prefix}_get_{name}"
        return f

    for name, (default, names) in table.items():
        globals()[f"{prefix}_set_{name}"] = setter(name, default, names)
        globals()[f"{prefix}_get_{name}"] = getter(name, default)


_make("pcair", _AIR, PCAIR)
_make("pcpflareinv", _INV, PCPFLAREINV)


def pcair_get_num_levels(pc):
    return pc.num_levels


def pcair_get_grid_complexity(pc):
    return pc.complexities()["grid"]


def pcair_get_operator_complexity(pc):
    return pc.complexities()["operator"]


def pcair_get_cycle_complexity(pc):
    return pc.complexities()["cycle"]


def pcair_get_storage_complexity(pc):
    return pc.complexities()["storage"]


def pcair_get_reuse_storage_complexity(pc):
    return 0.0          # nothing is kept for reuse (reuse_sparsity is out of scope)


def pcpflareinv_get_inverse_mat(pc, n=None):
    """the assembled approximate inverse as CsrArrays (PCPFLAREINVGetInverseMat: a borrowed Mat in the reference)"""
    d = pc.get_inverse(n if n is not None else 0)
    return d.get("mat")
