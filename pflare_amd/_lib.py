"""ctypes binding of libairhip.so (include/airhip.h).  Fails loudly when the native library
is missing - there is no Python or CPU fallback for the solve phase."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# AIRHIP_LIB: alternative build of the same library (e.g. a sanitizer build of the host code)
LIB_PATH = os.environ.get("AIRHIP_LIB") or os.path.join(_HERE, "libairhip.so")


class airhip_csr(C.Structure):
    _fields_ = [("m", C.c_int64), ("n", C.c_int64), ("rowptr", C.POINTER(C.c_int64)),
                ("col", C.POINTER(C.c_int32)), ("val", C.POINTER(C.c_double))]


class airhip_invop(C.Structure):
    _fields_ = [("kind", C.c_int), ("mat", airhip_csr), ("diag", C.POINTER(C.c_double)),
                ("ncoef", C.c_int), ("coef", C.POINTER(C.c_double)), ("coef_im", C.POINTER(C.c_double)),
                ("diag_scale", C.c_int), ("neumann", C.c_int)]


class airhip_ksp_options(C.Structure):
    _fields_ = [("type", C.c_int), ("pc_side", C.c_int), ("norm_type", C.c_int), ("rtol", C.c_double),
                ("atol", C.c_double), ("dtol", C.c_double), ("max_it", C.c_int), ("restart", C.c_int)]


class airhip_ksp_result(C.Structure):
    _fields_ = [("its", C.c_int), ("reason", C.c_int), ("rnorm", C.c_double), ("solve_ms", C.c_double),
                ("pc_applies", C.c_int), ("reserved", C.c_int)]


# every symbol include/airhip.h declares (tests check that the library exports all of them)
DECLARED_SYMBOLS = [
    "airhip_last_error", "airhip_version", "airhip_device_count", "airhip_set_device", "airhip_set_num_threads",
    "airhip_set_tuning", "airhip_pcair_setup_host_only", "airhip_pcinv_setup_host_only", "airhip_spgemm_filtered", "airhip_fused_first_operator",
    "airhip_gen_adv_1d", "airhip_gen_adv_diff_fd", "airhip_hostmat_from_csr", "airhip_hostmat_get",
    "airhip_hostmat_destroy",
    "airhip_pcair_create", "airhip_pcair_set_option", "airhip_pcair_set_ordering_hint", "airhip_pcair_setup", "airhip_pcair_begin_levels",
    "airhip_pcair_set_level", "airhip_pcair_set_coarse", "airhip_pcair_finalize", "airhip_pcair_apply", "airhip_pcair_apply_on_stream", "airhip_pcinv_apply_on_stream",
    "airhip_mat_mult_on_stream", "airhip_ksp_solve_on_stream",
    "airhip_pcair_sync", "airhip_pcair_destroy", "airhip_pcair_num_levels", "airhip_pcair_level_sizes",
    "airhip_pcair_level_get", "airhip_pcair_level_inv", "airhip_pcair_level_smooth_order",
    "airhip_pcair_coarse_get", "airhip_pcair_complexities", "airhip_pcair_cycle_alg_bytes", "airhip_pcair_setup_times",
    "airhip_pcair_release_host", "airhip_pcair_host_mirror_wait", "airhip_pcair_use_graph", "airhip_pcair_profile",
    "airhip_pcair_profile_reset", "airhip_pcair_profile_count", "airhip_pcair_profile_get", "airhip_pcair_profile_kernel",
    "airhip_pcinv_create", "airhip_pcinv_set_option", "airhip_pcinv_setup", "airhip_pcinv_apply",
    "airhip_pcinv_sync", "airhip_pcinv_get_inv", "airhip_pcinv_destroy",
    "airhip_mat_create", "airhip_mat_mult", "airhip_mat_destroy", "airhip_mat_blocks",
    "airhip_ksp_default_options", "airhip_ksp_solve",
    "airhip_partition_create", "airhip_partition_extract", "airhip_partition_destroy",
    "airhip_archive_create", "airhip_archive_add", "airhip_archive_count", "airhip_archive_item",
    "airhip_archive_destroy", "airhip_comm_rccl_unique_id", "airhip_comm_create_rccl",
    "airhip_comm_create_staged", "airhip_comm_stats", "airhip_comm_traffic", "airhip_pcinv_as_pcair", "airhip_comm_selftest", "airhip_compute_cf_splitting", "airhip_compute_cf_splitting_on_device", "airhip_compute_diag_dom_submatrix", "airhip_comm_destroy", "airhip_pcair_load_distributed",
]

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double),
                          C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int64))
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C pflare_amd/csrc`).  pflare_amd has no pure-Python fallback.")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    L.airhip_last_error.restype = C.c_char_p
    vp = C.c_void_p
    pvp = C.POINTER(C.c_void_p)
    pcsr = C.POINTER(airhip_csr)
    pinv = C.POINTER(airhip_invop)
    pd = C.POINTER(C.c_double)
    pi32 = C.POINTER(C.c_int32)
    pi64 = C.POINTER(C.c_int64)
    pint = C.POINTER(C.c_int)
    sig = {
        "airhip_device_count": [pint],
        "airhip_set_device": [C.c_int],
        "airhip_set_num_threads": [C.c_int, pint],
        "airhip_set_tuning": [C.c_char_p, C.c_int],
        "airhip_gen_adv_1d": [C.c_int64, pvp],
        "airhip_gen_adv_diff_fd": [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double,
                                   C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, pvp],
        "airhip_hostmat_from_csr": [pcsr, pvp],
        "airhip_hostmat_get": [vp, pcsr],
        "airhip_hostmat_destroy": [pvp],
        "airhip_pcair_create": [pvp],
        "airhip_pcair_set_option": [vp, C.c_char_p, C.c_char_p],
        "airhip_pcair_set_ordering_hint": [vp, pi64, C.c_int64],
        "airhip_pcair_setup": [vp, pcsr],
        "airhip_pcair_setup_host_only": [vp, pcsr],
        "airhip_pcair_begin_levels": [vp, C.c_int, pcsr],
        "airhip_pcair_set_level": [vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, pi32, pi32, pcsr, pd, pcsr, pinv,
                                   pcsr, pi32, pcsr, pcsr, pcsr, pinv, pint],
        "airhip_pcair_set_coarse": [vp, pcsr, pinv, C.c_int],
        "airhip_pcair_finalize": [vp],
        "airhip_pcair_apply": [vp, vp, vp],
        "airhip_pcair_apply_on_stream": [vp, vp, vp, vp],
        "airhip_pcair_sync": [vp],
        "airhip_pcair_destroy": [pvp],
        "airhip_pcair_num_levels": [vp, pint],
        "airhip_pcair_level_sizes": [vp, C.c_int, pi64, pi64, pi64, pint, pint, pi64],
        "airhip_pcair_level_get": [vp, C.c_int, C.c_char_p, pcsr, C.POINTER(pi32), C.POINTER(pd), pi64],
        "airhip_pcair_level_inv": [vp, C.c_int, C.c_int, pinv],
        "airhip_pcair_level_smooth_order": [vp, C.c_int, pint],
        "airhip_pcair_coarse_get": [vp, pcsr],
        "airhip_pcair_complexities": [vp, pd, pd, pd, pd, pd],
        "airhip_pcair_cycle_alg_bytes": [vp, pd],
        "airhip_pcair_setup_times": [vp, pd, pd],
        "airhip_pcair_release_host": [vp],
        "airhip_pcair_host_mirror_wait": [vp, pd],
        "airhip_pcair_use_graph": [vp, C.c_int],
        "airhip_pcair_profile": [vp, C.c_int],
        "airhip_pcair_profile_reset": [vp],
        "airhip_pcair_profile_count": [vp, pint],
        "airhip_pcair_profile_get": [vp, C.c_int, C.POINTER(C.c_char_p), pint, pd, pi64, pd],
        "airhip_pcair_profile_kernel": [vp, C.c_int, C.POINTER(C.c_char_p)],
        "airhip_pcinv_create": [pvp],
        "airhip_pcinv_set_option": [vp, C.c_char_p, C.c_char_p],
        "airhip_pcinv_setup": [vp, pcsr],
        "airhip_pcinv_setup_host_only": [vp, pcsr],
        "airhip_pcinv_apply": [vp, vp, vp],
        "airhip_pcinv_apply_on_stream": [vp, vp, vp, vp],
        "airhip_pcinv_sync": [vp],
        "airhip_pcinv_get_inv": [vp, pinv],
        "airhip_pcinv_destroy": [pvp],
        "airhip_mat_create": [pcsr, pvp],
        "airhip_mat_mult": [vp, vp, vp],
        "airhip_mat_mult_on_stream": [vp, vp, vp, vp],
        "airhip_mat_destroy": [pvp],
        "airhip_mat_blocks": [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), vp, vp, vp, vp],
        "airhip_ksp_default_options": [C.POINTER(airhip_ksp_options)],
        "airhip_ksp_solve": [C.POINTER(airhip_ksp_options), vp, vp, vp, vp, vp, C.POINTER(airhip_ksp_result), pd,
                             C.c_int, pint],
        "airhip_ksp_solve_on_stream": [C.POINTER(airhip_ksp_options), vp, vp, vp, vp, vp, C.POINTER(airhip_ksp_result), pd,
                                       C.c_int, pint, vp],
        "airhip_partition_create": [vp, pcsr, C.c_int, pi32, pvp],
        "airhip_partition_extract": [vp, C.c_int, pvp],
        "airhip_partition_destroy": [pvp],
        "airhip_archive_create": [pvp],
        "airhip_archive_add": [vp, C.c_char_p, C.c_int, vp, C.c_int64],
        "airhip_archive_count": [vp, pint],
        "airhip_archive_item": [vp, C.c_int, C.POINTER(C.c_char_p), pint, pvp, pi64],
        "airhip_archive_destroy": [pvp],
        "airhip_comm_rccl_unique_id": [vp],
        "airhip_comm_create_rccl": [C.c_int, C.c_int, vp, pvp],
        "airhip_comm_create_staged": [C.c_int, C.c_int, EXCHANGE_FN, ALLREDUCE_FN, vp, pvp],
        "airhip_comm_stats": [vp, pi64, pi64],
        "airhip_comm_traffic": [vp, pi64, pi64],
        "airhip_pcinv_as_pcair": [vp, C.POINTER(vp)],
        "airhip_comm_selftest": [vp, C.c_int64, pint],
        "airhip_compute_cf_splitting": [pcsr, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int32)],
        "airhip_compute_cf_splitting_on_device": [pcsr, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int32)],
        "airhip_compute_diag_dom_submatrix": [pcsr, C.c_double, pvp],
        "airhip_comm_destroy": [pvp],
        "airhip_pcair_load_distributed": [vp, vp, vp],
    }
    for name, args in sig.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = None if name == "airhip_ksp_default_options" else C.c_int
    _lib = L
    return L


class AirhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"airhip error {code}: {msg}")
        self.code = code


def check(code):
    if code != 0:
        raise AirhipError(code, lib().airhip_last_error().decode())
