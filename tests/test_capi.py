"""The C-ABI library loads on a machine without a GPU, exports every symbol include/airhip.h
declares, and fails loudly (no CPU fallback) when asked to run the solve phase without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import pflare_amd as P
from pflare_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "airhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(airhip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = P.lib()
    names = header_functions()
    assert len(names) >= 40
    for name in names:
        assert hasattr(L, name), f"libairhip.so does not export {name}"
    assert sorted(names) == sorted(_lib.DECLARED_SYMBOLS)


def test_error_convention_and_messages():
    L = P.lib()
    assert L.airhip_version() >= 100
    h = C.c_void_p()
    assert L.airhip_gen_adv_diff_fd(2, 1, 5, 5, 1.0, 1.0, 1.0, 1, 0.0, 1.0, 1.0, 1.0, -1, C.byref(h)) == -1
    assert b"at least 2 points" in L.airhip_last_error()
    pc = C.c_void_p()
    assert L.airhip_pcair_create(C.byref(pc)) == 0
    assert L.airhip_pcair_apply(pc, None, None) == -2                      # not set up
    assert L.airhip_pcair_set_option(pc, b"poly_order", b"4") == 0
    assert L.airhip_pcair_set_option(pc, b"-pc_air_a_drop", b"1e-3") == 0   # full option names accepted
    assert L.airhip_pcair_destroy(C.byref(pc)) == 0 and not pc.value


def _no_gpu():
    n = C.c_int()
    P.lib().airhip_device_count(C.byref(n))
    return n.value == 0


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is visible")
def test_solve_phase_fails_loudly_without_a_device():
    A = P.HostMat.adv_diff_fd(dim=2, M=8, N=8)
    with pytest.raises(P.AirhipError) as ei:
        P.PCAIR().setup(A)
    assert ei.value.code == -3 and "no CPU fallback" in str(ei.value)
    with pytest.raises(P.AirhipError):
        P.PCPFLAREINV().setup(A)
    with pytest.raises(P.AirhipError):
        P.DeviceMat(A)
    # the host-side setup itself works without a device
    pc = P.PCAIR().setup_host_only(A)
    assert pc.num_levels >= 2


def test_export_roundtrip_zero_copy_views():
    A = P.HostMat.adv_diff_fd(dim=2, M=12, N=12)
    pc = P.PCAIR().setup_host_only(A)
    a = pc.export_level(0, copy=True); b = pc.export_level(0, copy=False)
    assert np.array_equal(a["F"], b["F"]) and np.array_equal(a["Z"].val, b["Z"].val)
    cx = pc.complexities()
    assert cx["cycle"] > 0 and cx["cycle_alg_bytes"] == 0.0   # device bytes only after a device setup


def test_print_stats_timings_output_has_the_reference_labels():
    """-pc_air_print_stats_timings: the per-level block and the timer labels of the reference
    (src/AIR_MG_Setup.F90:324-325,1220-1225, src/Timers.F90:30-46) so that its tools/parse_pflare_output.py
    reads our output (checked by hand against that script; here: the same regular expressions)."""
    import re
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import pflare_amd as P\n"
            "A = P.HostMat.adv_diff_fd(dim=2, M=30, N=30)\n"
            "P.PCAIR(print_stats_timings=True).setup_host_only(A)\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True, timeout=120).stdout
    levels = re.findall(r"~~~~~~~~~~~~ Level\s+(\d+)", out)
    assert levels and levels[0] == "1"
    assert re.search(r"Global rows\s+900\s+Global F-points\s+(\d+)\s+Global C-points\s+(\d+)", out)
    for label in ("coarsen time", "extract time", "proc agglom time", "inverse time", "restrict time", "prolong time",
                  "constrain time", "rap time", "identity time", "drop time", "truncate time"):
        assert len(re.findall(r"\b" + re.escape(label) + r"\s*:\s*(\S+)", out)) == len(levels), label
    assert re.search(r"~~~~~~~~~~~~ Coarse grid\s+(\d+)", out)
    assert re.search(r"Total cumulative setup time\s*:\s*(\S+)", out)


def test_operators_are_validated_at_the_boundary():
    """anything the kernels would index out of range with is refused with AIRHIP_ERR_ARG, not passed on"""
    n = 6
    rp = np.arange(0, 2 * n + 1, 2, dtype=np.int64)
    good = np.array([[i, (i + 1) % n] if i < n - 1 else [0, i] for i in range(n)], dtype=np.int32)
    good.sort(axis=1)
    val = np.ones(2 * n)
    ok = P.PCAIR().setup_host_only((rp, good.ravel().copy(), val, (n, n)))
    assert ok.num_levels >= 1
    cases = {
        "column index out of range": good.ravel().copy(),
        "negative": good.ravel().copy(),
        "strictly ascending": good[:, ::-1].ravel().copy(),
    }
    cases["column index out of range"][3] = n
    cases["negative"][0] = -1
    for what, col in cases.items():
        with pytest.raises(P.AirhipError):
            P.PCAIR().setup_host_only((rp, col, val, (n, n)))
        with pytest.raises(P.AirhipError):
            P.PCPFLAREINV().setup_host_only((rp, col, val, (n, n)))
    bad_rp = rp.copy(); bad_rp[2] = bad_rp[3] + 1
    with pytest.raises(P.AirhipError):
        P.PCAIR().setup_host_only((bad_rp, good.ravel().copy(), val, (n, n)))
    with pytest.raises(P.AirhipError):
        P.PCAIR().setup_host_only((rp, good.ravel().copy(), val, (n, n + 1)))


def test_bring_your_own_hierarchy_is_validated_then_needs_a_device():
    """an oracle hierarchy passes the boundary checks; without a GPU the upload fails loudly (no CPU fallback);
    a corrupted operand is refused before that"""
    import torch
    from oracle import air_ref as R, bridge
    A = R.gen_adv_diff_fd(2, 12, 12)
    H = R.setup_air(A, R.AirOptions(smooth_type="fcf"))
    levels, cA, cinv = bridge.levels_for_product(H)
    if not torch.cuda.is_available():
        with pytest.raises(P.AirhipError, match="no HIP device"):
            P.PCAIR().set_hierarchy(A, levels, cA, cinv)
    bad = [dict(lv) for lv in levels]
    z = bad[0]["Z"].tocsr().copy(); z.sum_duplicates()
    z.indices = z.indices.copy(); z.indices[0] = z.shape[1] + 5
    bad[0]["Z"] = P.CsrArrays(z.indptr, z.indices, z.data, z.shape)
    with pytest.raises(P.AirhipError, match="Z: column index out of range"):
        P.PCAIR().set_hierarchy(A, bad, cA, cinv)
    bad = [dict(lv) for lv in levels]
    bad[0]["F"] = np.asarray(bad[0]["F"]).copy(); bad[0]["F"][0] = A.shape[0]
    with pytest.raises(P.AirhipError, match="F index out of range"):
        P.PCAIR().set_hierarchy(A, bad, cA, cinv)


def test_setup_can_be_repeated_on_the_same_object():
    """PCSetUp called again after the operator changed (the reference's -regen drivers): the second hierarchy must be
    the one of the second matrix, with nothing carried over"""
    from oracle import air_ref as R, bridge
    A1 = R.gen_adv_diff_fd(2, 12, 12)
    A2 = R.gen_adv_diff_fd(2, 14, 14, u=0, v=0, alpha=1.0)
    pc = P.PCAIR(a_drop=1e-3)
    pc.setup_host_only(A1)
    pc.setup_host_only(A2)
    fresh = P.PCAIR(a_drop=1e-3).setup_host_only(A2)
    assert pc.num_levels == fresh.num_levels
    for lv in range(pc.num_levels - 1):
        a, b = pc.export_level(lv), fresh.export_level(lv)
        assert np.array_equal(a["F"], b["F"]) and np.array_equal(a["Z"].val, b["Z"].val)


def test_inverse_type_constants_equal_the_reference_enum():
    """include/airhip.h's AIRHIP_INV_* = PCPFLAREINVType of the reference's include/pflare.h:36-46, value for value
    (fixture tests/golden/reference_enums.json; re-derived from the reference header when it is present)."""
    import json
    import importlib.util
    gold = os.path.join(ROOT, "tests", "golden")
    ref = json.load(open(os.path.join(gold, "reference_enums.json")))["enums"]
    if os.path.exists("/root/reference/include/pflare.h"):
        spec = importlib.util.spec_from_file_location("make_reference_enums", os.path.join(gold, "make_reference_enums.py"))
        mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
        assert mod.parse_enums("/root/reference/include/pflare.h") == ref
    txt = open(os.path.join(ROOT, "include", "airhip.h")).read()
    mine = {k: int(v) for k, v in re.findall(r"#define\s+AIRHIP_INV_(\w+)\s+(\d+)", txt)}
    want = {k[len("PFLAREINV_"):]: v for k, v in ref["PCPFLAREINVType"].items()}
    assert mine == want
    # the Python mirror of the plugin surface uses the same numbers
    from pflare_amd import pflare
    for k, v in ref["PCPFLAREINVType"].items():
        assert int(getattr(pflare.PCPFLAREINVType, k)) == v
