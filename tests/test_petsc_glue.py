"""pflare_amd/csrc/petsc_glue.c cannot be compiled here (no PETSc in the image), so what can be checked is checked:
it exports the reference's plugin symbols with the reference's vtable, calls nothing but what include/airhip.h
declares, forwards only options the library knows, and the Makefile target refuses to run without PETSC_DIR."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "pflare_amd", "csrc", "petsc_glue.c")).read()
HDR = open(os.path.join(ROOT, "include", "airhip.h")).read() + open(os.path.join(ROOT, "include", "airhip_mpi.h")).read()


def test_exports_the_references_plugin_symbols():
    # src/PCAIR.c:3601-3663, src/PCPFLAREINV.c:887-957
    for sig in (r"PETSC_EXTERN PetscErrorCode PCCreate_AIR\(PC pc\)", r"PETSC_EXTERN void PCRegister_AIR\(void\)",
                r"PETSC_EXTERN PetscErrorCode PCCreate_PFLAREINV\(PC pc\)", r"PETSC_EXTERN void PCRegister_PFLAREINV\(void\)",
                r"PETSC_EXTERN PetscErrorCode PetscDLLibraryRegister_petscpflare\(void\)"):
        assert re.search(sig, SRC), sig
    assert 'PCRegister("air", PCCreate_AIR)' in SRC and 'PCRegister("pflareinv", PCCreate_PFLAREINV)' in SRC
    for op in ("apply", "setup", "destroy", "view", "reset", "setfromoptions"):      # src/PCAIR.c:3636-3641
        assert re.search(r"pc->ops->%s\s*=" % op, SRC), op


def test_calls_only_declared_abi_functions():
    called = set(re.findall(r"\b(airhip_[a-z0-9_]+)\s*\(", SRC)) - {"airhip_push_option", "airhip_make_handle", "airhip_complexity"}
    declared = set(re.findall(r"\b(airhip_[a-z0-9_]+)\s*\(", HDR))
    assert called and called <= declared, called - declared
    # the apply path goes through the stream-ordered entry points (no host synchronisation per PCApply)
    assert "airhip_pcair_apply_on_stream" in called and "airhip_pcinv_apply_on_stream" in called


def test_forwarded_options_are_known_to_the_library():
    import pflare_amd as P
    air = re.search(r"air_options\[\] = \{(.*?)NULL\}", SRC, re.S).group(1)
    inv = re.search(r"inv_options\[\] = \{(.*?)NULL\}", SRC, re.S).group(1)
    sample = {"smooth_type": "ff", "inverse_type": "arnoldi", "c_inverse_type": "arnoldi", "coarsest_inverse_type": "arnoldi",
              "z_type": "product", "cf_splitting_type": "0", "type": "arnoldi"}
    pc = P.PCAIR()
    for name in re.findall(r'"([a-z_0-9]+)"', air):
        pc.set_option(name, sample.get(name, "1"))          # raises on an unknown name
    pi = P.PCPFLAREINV()
    for name in re.findall(r'"([a-z_0-9]+)"', inv):
        pi.set_option(name, sample.get(name, "1"))


def test_make_target_needs_petsc():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "pflare_amd", "csrc"), "petsc_glue", "PETSC_DIR="],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "PETSC_DIR" in (r.stdout + r.stderr)


def test_typed_entry_points_follow_the_reference_header():
    """every PCAIRSet* / PCAIRGet* / PCPFLAREINVSet* the glue defines exists in the reference's include/pflare.h with the
    same parameter list (names committed in tests/golden/reference_prototypes.json), and the option each setter stores
    is one the library accepts"""
    import json
    import pflare_amd as P
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_prototypes.json")))["prototypes"]
    setters = re.findall(r"AIRHIP_SETTER\((\w+), (\w+), \"(\w+)\"", SRC)
    enums = re.findall(r"AIRHIP_ENUM_SETTER\((\w+), \"(\w+)\"", SRC)
    assert len(setters) >= 30 and len(enums) == 5
    for fn, typ, opt in setters:
        assert fn in ref, fn
        assert ref[fn][0] == "PC" and len(ref[fn]) == 2, (fn, ref[fn])
        want = ref[fn][1]
        assert typ == want or (typ == "int" and want in ("CFSplittingType",)), (fn, typ, want)
    for fn, opt in enums:
        assert fn in ref and ref[fn][1] in ("PCPFLAREINVType", "PCAIRZType"), fn
    for fn in ("PCAIRGetNumLevels", "PCAIRGetGridComplexity", "PCAIRGetOperatorComplexity", "PCAIRGetCycleComplexity",
               "PCAIRGetStorageComplexity", "PCAIRSetSmoothType"):
        assert fn in ref and re.search(r"PETSC_EXTERN PetscErrorCode %s\(" % fn, SRC), fn
    sample = {"cf_splitting_type": "0"}
    pc, pi = P.PCAIR(), P.PCPFLAREINV()
    for fn, typ, opt in setters:
        (pi if fn.startswith("PCPFLAREINV") else pc).set_option(opt, sample.get(opt, "1"))
    # the enum tables are the reference's enumerators in order
    enums_ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_enums.json")))["enums"]
    names = re.search(r"airhip_inv_names\[\]\s*=\s*\{(.*?)\}", SRC, re.S).group(1)
    mine = re.findall(r'"(\w+)"', names)
    want = [k[len("PFLAREINV_"):].lower() for k, v in sorted(enums_ref["PCPFLAREINVType"].items(), key=lambda kv: kv[1])]
    assert mine == want


def test_mpi_path_is_wired():
    assert "airhip_mpi_pcair_setup" in SRC and "MatMPIAIJGetSeqAIJ" in SRC
    assert "PETSC_VERSION_GE(3, 23, 0)" in SRC          # PetscOptionItems by value only from 3.23 on
    assert "hipDeviceSynchronize" in SRC                 # no stream handle: synchronise instead of assuming the NULL stream


import pytest
import shutil


@pytest.mark.parametrize("defs", [["-DAIRHIP_WITH_MPI"], [], ["-DAIRHIP_WITH_MPI", "-DPETSC_VERSION_MINOR=22"], ["-DPETSC_VERSION_MINOR=19"],
                                  ["-DAIRHIP_PETSC_NO_STREAM_HANDLE"]],
                         ids=["mpi", "one-rank", "petsc-3.22-options-by-pointer", "petsc-3.19-no-stream-handle", "no-stream-handle"])
def test_glue_passes_the_c_compiler_against_a_mock_of_the_petsc_api(defs):
    """No PETSc in this image: the glue is compiled (-fsyntax-only -Wall -Wextra -Werror) against tests/petsc_mock, a minimal
    mock of the PETSc API subset it uses (declarations only, written from the manual pages).  This checks the C of the file -
    syntax, types, macro expansions, argument counts against include/airhip.h and include/airhip_mpi.h - in the
    configurations its version macros select; it proves nothing about a real PETSc."""
    if not shutil.which("gcc") or not os.path.exists("/opt/conda/include/mpi.h"):
        pytest.skip("needs gcc and an mpi.h")
    mock = os.path.join(ROOT, "tests", "petsc_mock")
    cmd = ["gcc", "-fsyntax-only", "-std=c11", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-D__HIP_PLATFORM_AMD__"] + defs + \
          ["-I" + mock, "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include", "-idirafter", "/opt/conda/include",
           os.path.join(ROOT, "pflare_amd", "csrc", "petsc_glue.c")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
