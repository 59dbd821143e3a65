"""PETSc command lines of the reference's tests, read as this package's options.

The reference's tests are command lines (tests/Makefile): `./adv_diff_fd -da_grid_x 100 -da_grid_y 100 -pc_type air
-ksp_pc_side right -ksp_rtol 1e-10 -pc_air_a_lump -pc_air_a_drop 1e-5 ...`.  `parse_petsc_args` splits such a line into
what the three objects of this package take - the generator (tests/adv_diff_fd.c:120-257 option handling), the
preconditioner (`-pc_air_*` / `-pc_pflareinv_*`, docs/options.md:22-106; the names travel to airhip_pcair_set_option
unchanged, without the prefix) and the Krylov solver (`-ksp_*`) - so that a parity test can quote the reference's line.
Options of PETSc objects this build does not have (`-dm_*`, logging, `-mg_*` other than the two coarse-grid options the
library takes) are returned in `other`; nothing is dropped silently.
"""
import shlex

__all__ = ["parse_petsc_args", "build", "matrix"]

_KSP = {"ksp_type": ("ksp_type", str), "ksp_pc_side": ("pc_side", str), "ksp_norm_type": ("norm_type", str),
        "ksp_rtol": ("rtol", float), "ksp_atol": ("atol", float), "ksp_divtol": ("divtol", float),
        "ksp_max_it": ("max_it", int), "ksp_gmres_restart": ("restart", int)}
# tests/adv_diff_fd.c:120-257 (PetscOptionsGet*): grid, velocity, diffusion coefficient, domain lengths
_GEN = {"da_grid_x": ("M", int), "da_grid_y": ("N", int), "da_grid_z": ("P", int), "dim": ("dim", int), "u": ("u", float),
        "v": ("v", float), "w": ("w", float), "alpha": ("alpha", float), "L_x": ("L_x", float), "L_y": ("L_y", float),
        "L_z": ("L_z", float), "adv_nondim": ("adv_nondim", "flag"), "n": ("n", int)}
# the coarse-grid options the reference reaches through PCMG's own prefix (tests/Makefile:421-424, :144-149): the library
# takes them under their PETSc names (airhip_pcair_set_option); -mg_coarse_ksp_type richardson is what max_it > 1 means
_MG = ("mg_coarse_ksp_max_it", "mg_coarse_pc_type")


def _value(tok):
    t = tok.lower()
    if t in ("true", "yes", "on"):
        return True
    if t in ("false", "no", "off"):
        return False
    try:
        return int(tok)
    except ValueError:
        pass
    try:
        return float(tok)
    except ValueError:
        return tok


def _is_number(tok):
    try:
        float(tok)
        return True
    except ValueError:
        return False


def parse_petsc_args(line):
    """-> dict(program, pc_type, pc, ksp, gen, other): `program` the driver's name if the line starts with one, `pc` the -pc_air_* / -pc_pflareinv_* options without their prefix (a flag
    given without a value is True, as PetscOptionsBool reads it), `ksp` keyword arguments of pflare_amd.KSP, `gen` those
    of HostMat.adv_diff_fd / adv_1d, `other` everything else (name -> value)."""
    toks = shlex.split(line) if isinstance(line, str) else list(line)
    out = dict(program=None, pc_type=None, pc={}, ksp={}, gen={}, other={})
    if toks and not toks[0].startswith("-"):
        out["program"] = toks[0].rsplit("/", 1)[-1]
        if out["program"] == "adv_1d":
            out["pc_type"] = "air"          # the driver sets it itself (tests/adv_1d.c:142)
    i = 0
    while i < len(toks):
        t = toks[i]
        if not (t.startswith("-") and not _is_number(t)):
            i += 1          # the program name, or a stray value
            continue
        name = t.lstrip("-")
        val = True
        if i + 1 < len(toks) and (not toks[i + 1].startswith("-") or _is_number(toks[i + 1])):
            val = _value(toks[i + 1])
            i += 1
        i += 1
        if name == "pc_type":
            out["pc_type"] = val
        elif name.startswith("pc_air_"):
            out["pc"][name[len("pc_air_"):]] = val
        elif name.startswith("pc_pflareinv_"):
            out["pc"][name[len("pc_pflareinv_"):]] = val
        elif name in _KSP:
            k, conv = _KSP[name]
            out["ksp"][k] = conv(val)
        elif name in _MG:
            out["pc"][name] = val
        elif name in _GEN:
            k, conv = _GEN[name]
            out["gen"][k] = bool(val) if conv == "flag" else conv(val)
        else:
            out["other"][name] = val
    return out


def build(line):
    """The objects of one command line: (pc, ksp, gen, other) with pc a PCAIR / PCPFLAREINV carrying the line's options
    (not set up), ksp a KSP, gen the generator's keyword arguments (`matrix(line)` builds the operator)."""
    from .pc import KSP, PCAIR, PCPFLAREINV
    o = parse_petsc_args(line)
    if o["pc_type"] not in ("air", "pflareinv"):
        raise ValueError(f"-pc_type {o['pc_type']}: this build provides the PC types 'air' and 'pflareinv'")
    pc = (PCAIR if o["pc_type"] == "air" else PCPFLAREINV)(**o["pc"])
    return pc, KSP(**o["ksp"]), o["gen"], o["other"]


def matrix(line):
    """The operator the line's driver assembles: tests/adv_1d.c (-n, the driver's default 100, tests/adv_1d.c:21) or tests/adv_diff_fd.c."""
    from .pc import HostMat
    o = parse_petsc_args(line)
    if o["program"] == "adv_1d":
        return HostMat.adv_1d(o["gen"].get("n", 100))
    return HostMat.adv_diff_fd(**o["gen"])
