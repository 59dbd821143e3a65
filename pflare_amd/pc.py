"""Host-side mirror of the reference's plugin surface for the PCAIR solve path.

Names follow the reference: PC types "air" / "pflareinv" (include/pflare.h:97-98), options are
the `-pc_air_*` / `-pc_pflareinv_*` names of docs/options.md without the prefix, KSP semantics are
PETSc's (`-ksp_type`, `-ksp_pc_side`, `-ksp_rtol`, `-ksp_max_it`, `-ksp_norm_type`).  All arithmetic
runs in libairhip.so on the GPU; this module only marshals arguments.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import (AirhipError, airhip_csr, airhip_invop, airhip_ksp_options, airhip_ksp_result,
                   check, lib)

PCAIR_NAME = "air"
PCPFLAREINV_NAME = "pflareinv"

OP_NONE, OP_ASSEMBLED, OP_DIAG, OP_HORNER, OP_NEWTON = 0, 1, 2, 3, 4


# ------------------------------------------------------------------------------------------
# CSR marshalling
# ------------------------------------------------------------------------------------------
class CsrArrays:
    """Owns contiguous (rowptr int64, col int32, val float64) arrays and the ctypes view."""

    def __init__(self, rowptr, col, val, shape):
        self.rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col, dtype=np.int32)
        self.val = np.ascontiguousarray(val, dtype=np.float64)
        self.shape = (int(shape[0]), int(shape[1]))
        self.c = airhip_csr(self.shape[0], self.shape[1],
                            self.rowptr.ctypes.data_as(C.POINTER(C.c_int64)),
                            self.col.ctypes.data_as(C.POINTER(C.c_int32)),
                            self.val.ctypes.data_as(C.POINTER(C.c_double)))

    @property
    def nnz(self):
        return int(self.rowptr[-1])

    def ref(self):
        return C.byref(self.c)


def as_csr(A) -> CsrArrays:
    """Accepts CsrArrays, HostMat, a scipy-like CSR (indptr/indices/data/shape, sorted indices)
    or a (rowptr, col, val, shape) tuple."""
    if isinstance(A, CsrArrays):
        return A
    if isinstance(A, HostMat):
        v = _view_csr(A.csr_view(), copy=False)   # zero copy; keep the owner alive
        v._owner = A
        return v
    if hasattr(A, "indptr"):
        # the library requires strictly ascending column indices per row (it validates and refuses otherwise);
        # scipy products routinely come unsorted, so canonicalise a copy here
        if hasattr(A, "has_canonical_format") and not A.has_canonical_format:
            A = A.copy()
            A.sum_duplicates()
        return CsrArrays(A.indptr, A.indices, A.data, A.shape)
    rowptr, col, val, shape = A
    return CsrArrays(rowptr, col, val, shape)


def _arr(ptr, n, copy, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    a = np.ctypeslib.as_array(ptr, shape=(n,))
    return a.copy() if copy else a


def _view_csr(c: airhip_csr, copy=True) -> CsrArrays:
    """numpy copies (or, with copy=False, views valid until the owner is released) of a
    library-owned CSR."""
    m, n = int(c.m), int(c.n)
    rp = _arr(c.rowptr, m + 1, copy, np.int64)
    nnz = int(rp[-1])
    return CsrArrays(rp, _arr(c.col, nnz, copy, np.int32), _arr(c.val, nnz, copy, np.float64), (m, n))


class HostMat:
    """A host CSR owned by the library (synthetic operators of the reference's test drivers)."""

    def __init__(self, handle):
        self._h = handle

    @staticmethod
    def adv_1d(n):
        """tests/adv_1d.c:80-105"""
        h = C.c_void_p()
        check(lib().airhip_gen_adv_1d(n, C.byref(h)))
        return HostMat(h)

    @staticmethod
    def adv_diff_fd(dim=2, M=11, N=11, P=11, u=None, v=None, w=None, alpha=0.0, L_x=1.0, L_y=1.0,
                    L_z=1.0, adv_nondim=None):
        """tests/adv_diff_fd.c option handling (:188-257) + ComputeMat (:366-586)."""
        unit = not (u is not None or v is not None or w is not None)
        uu = 1.0 if u is None else float(u)
        vv = 1.0 if v is None else float(v)
        ww = 1.0 if w is None else float(w)
        nd = -1 if adv_nondim is None else int(bool(adv_nondim))
        h = C.c_void_p()
        check(lib().airhip_gen_adv_diff_fd(dim, M, N, P, uu, vv, ww, int(unit), float(alpha), L_x, L_y, L_z,
                                           nd, C.byref(h)))
        return HostMat(h)

    def csr_view(self):
        c = airhip_csr()
        check(lib().airhip_hostmat_get(self._h, C.byref(c)))
        return c

    def csr(self) -> CsrArrays:
        return _view_csr(self.csr_view())

    @property
    def shape(self):
        c = self.csr_view()
        return (int(c.m), int(c.n))

    @property
    def nnz(self):
        c = self.csr_view()
        return int(c.rowptr[c.m])

    def __del__(self):
        try:
            if self._h:
                lib().airhip_hostmat_destroy(C.byref(self._h))
        except Exception:
            pass


def _ptr(t):
    """device pointer of a torch tensor (float64, contiguous, on the GPU)."""
    if hasattr(t, "data_ptr"):
        if str(t.dtype) != "torch.float64" or not t.is_contiguous() or not t.is_cuda:
            raise ValueError("vectors must be contiguous float64 tensors on the GPU")
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(int(t))


def _caller_stream(t):
    """the HIP stream the caller's torch work is queued on (torch's current stream of the tensor's device), as the
    void* the *_on_stream entry points take; None (NULL = legacy default stream) for raw pointers"""
    if hasattr(t, "data_ptr"):
        import torch
        return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return None


def _invop_to_dict(inv: airhip_invop, n, copy=True):
    d = {"kind": int(inv.kind), "diag_scale": bool(inv.diag_scale), "neumann": bool(inv.neumann)}
    if inv.kind == OP_ASSEMBLED:
        d["mat"] = _view_csr(inv.mat, copy)
    elif inv.kind == OP_DIAG:
        d["d"] = _arr(inv.diag, n, copy, np.float64)
    elif inv.kind == OP_HORNER:
        d["coeffs"] = np.ctypeslib.as_array(inv.coef, shape=(inv.ncoef,)).copy()
    elif inv.kind == OP_NEWTON:
        d["re"] = np.ctypeslib.as_array(inv.coef, shape=(inv.ncoef,)).copy()
        d["im"] = np.ctypeslib.as_array(inv.coef_im, shape=(inv.ncoef,)).copy()
    return d


# CFSplittingType of the reference (python/pflare.py, include/pflare.h); only these three are built
CF_PMISR_DDC, CF_DIAG_DOM, CF_CR = 0, 1, 6


def compute_cf_splitting(A, symmetric=False, strong_threshold=0.5, max_luby_steps=-1, cf_splitting_type=CF_PMISR_DDC,
                         ddc_its=1, fraction_swap=0.1, on_device=False):
    """Stand-alone CF splitting with the reference's argument order (pflare.compute_cf_splitting,
    include/pflare.h:113); returns (is_fine, is_coarse) as ascending index arrays.  on_device: the HIP kernels
    (PMISR + DDC only; identical result)."""
    a = as_csr(A)
    markers = np.empty(a.shape[0], dtype=np.int32)
    if on_device:
        if cf_splitting_type != CF_PMISR_DDC:
            raise ValueError("the device splitting implements pmisr_ddc only")
        check(lib().airhip_compute_cf_splitting_on_device(a.ref(), int(bool(symmetric)), float(strong_threshold),
                                                          int(max_luby_steps), int(ddc_its), float(fraction_swap),
                                                          markers.ctypes.data_as(C.POINTER(C.c_int32))))
        return np.flatnonzero(markers == -1), np.flatnonzero(markers != -1)
    check(lib().airhip_compute_cf_splitting(a.ref(), int(bool(symmetric)), float(strong_threshold), int(max_luby_steps),
                                            int(cf_splitting_type), int(ddc_its), float(fraction_swap),
                                            markers.ctypes.data_as(C.POINTER(C.c_int32))))
    return np.flatnonzero(markers == -1), np.flatnonzero(markers != -1)


def compute_diag_dom_submatrix(A, max_dd_ratio):
    """pflare.compute_diag_dom_submatrix (include/pflare.h:114): the diagonally dominant submatrix A(F, F) of the diag_dom
    CF splitting with that ratio, as a HostMat"""
    a = as_csr(A)
    h = C.c_void_p()
    check(lib().airhip_compute_diag_dom_submatrix(a.ref(), float(max_dd_ratio), C.byref(h)))
    return HostMat(h)


class _InvHolder:
    """keeps the numpy arrays of an airhip_invop alive"""

    def __init__(self, d):
        self.keep = []
        self.c = airhip_invop()
        self.c.kind = int(d["kind"])
        self.c.diag_scale = int(bool(d.get("diag_scale", False)))
        self.c.neumann = int(bool(d.get("neumann", False)))
        if self.c.kind == OP_ASSEMBLED:
            m = as_csr(d["mat"]); self.keep.append(m); self.c.mat = m.c
        elif self.c.kind == OP_DIAG:
            a = np.ascontiguousarray(d["d"], dtype=np.float64); self.keep.append(a)
            self.c.diag = a.ctypes.data_as(C.POINTER(C.c_double))
        elif self.c.kind == OP_HORNER:
            a = np.ascontiguousarray(d["coeffs"], dtype=np.float64); self.keep.append(a)
            self.c.ncoef = a.size; self.c.coef = a.ctypes.data_as(C.POINTER(C.c_double))
        elif self.c.kind == OP_NEWTON:
            a = np.ascontiguousarray(d["re"], dtype=np.float64)
            b = np.ascontiguousarray(d["im"], dtype=np.float64); self.keep += [a, b]
            self.c.ncoef = a.size
            self.c.coef = a.ctypes.data_as(C.POINTER(C.c_double))
            self.c.coef_im = b.ctypes.data_as(C.POINTER(C.c_double))


class PCAIR:
    """PC type "air": PCCreate_AIR / PCSetFromOptions / PCSetUp / PCApply (src/PCAIR.c)."""

    def __init__(self, **options):
        self._h = C.c_void_p()
        check(lib().airhip_pcair_create(C.byref(self._h)))
        self.set_options(**options)

    def set_options(self, **options):
        for k, v in options.items():
            self.set_option(k, v)

    def set_option(self, name, value):
        if isinstance(value, bool):
            value = "1" if value else "0"
        check(lib().airhip_pcair_set_option(self._h, name.encode(), repr(value).encode() if isinstance(value, float)
                                            else str(value).encode()))

    def set_ordering_hint(self, key):
        """Locality keys of the operator's rows (device layout only; see include/airhip.h)."""
        k = np.ascontiguousarray(key, dtype=np.int64)
        check(lib().airhip_pcair_set_ordering_hint(self._h, k.ctypes.data_as(C.POINTER(C.c_int64)), k.size))

    # -- setup
    def setup(self, A):
        self._A = as_csr(A)
        check(lib().airhip_pcair_setup(self._h, self._A.ref()))
        self._A = None
        return self

    def setup_host_only(self, A):
        a = as_csr(A)
        check(lib().airhip_pcair_setup_host_only(self._h, a.ref()))
        return self

    def set_hierarchy(self, A_top, levels, coarse_A, coarse_inv, coarse_richardson_its=0):
        """Bring-your-own hierarchy: `levels` is a list of dicts with keys n, F, C, A_ff | aff_diag,
        A_fc, inv_ff, Z, wcol | W, [A_cf, A_cc, inv_cc], smooth_order."""
        L = lib()
        keep = []
        a = as_csr(A_top); keep.append(a)
        check(L.airhip_pcair_begin_levels(self._h, len(levels) + 1, a.ref()))
        for li, lv in enumerate(levels):
            F = np.ascontiguousarray(lv["F"], dtype=np.int32); Cc = np.ascontiguousarray(lv["C"], dtype=np.int32)

            def opt_csr(key):
                if lv.get(key) is None:
                    return None, None
                m = as_csr(lv[key]); keep.append(m)
                return m, m.ref()

            _, aff = opt_csr("A_ff")
            affd = None
            if lv.get("aff_diag") is not None:
                affd_arr = np.ascontiguousarray(lv["aff_diag"], dtype=np.float64); keep.append(affd_arr)
                affd = affd_arr.ctypes.data_as(C.POINTER(C.c_double))
            _, afc = opt_csr("A_fc")
            _, z = opt_csr("Z")
            _, w = opt_csr("W")
            _, acf = opt_csr("A_cf")
            _, acc = opt_csr("A_cc")
            wcol = None
            if lv.get("wcol") is not None:
                wc = np.ascontiguousarray(lv["wcol"], dtype=np.int32); keep.append(wc)
                wcol = wc.ctypes.data_as(C.POINTER(C.c_int32))
            inv_ff = _InvHolder(lv["inv_ff"]); keep.append(inv_ff)
            inv_cc = _InvHolder(lv["inv_cc"]) if lv.get("inv_cc") is not None else None
            so = (C.c_int * 10)(*([int(s) for s in lv["smooth_order"]] + [0] * 10)[:10])
            check(L.airhip_pcair_set_level(
                self._h, li, int(lv["n"]), F.size, Cc.size,
                F.ctypes.data_as(C.POINTER(C.c_int32)), Cc.ctypes.data_as(C.POINTER(C.c_int32)),
                aff, affd, afc, C.byref(inv_ff.c), z, wcol, w, acf, acc,
                C.byref(inv_cc.c) if inv_cc else None, so))
        ca = as_csr(coarse_A); cinv = _InvHolder(coarse_inv); keep += [ca, cinv]
        check(L.airhip_pcair_set_coarse(self._h, ca.ref(), C.byref(cinv.c), int(coarse_richardson_its)))
        check(L.airhip_pcair_finalize(self._h))
        return self

    # -- apply
    def apply(self, x, y):
        """y = M x : one AIR V-cycle (PCApply_AIR_c), torch float64 CUDA tensors.  Ordered on torch's current
        stream (input and output side); no host synchronisation."""
        check(lib().airhip_pcair_apply_on_stream(self._h, _ptr(x), _ptr(y), _caller_stream(x)))
        return y

    def sync(self):
        check(lib().airhip_pcair_sync(self._h))

    # -- inspection
    @property
    def num_levels(self):
        n = C.c_int()
        check(lib().airhip_pcair_num_levels(self._h, C.byref(n)))
        return n.value

    def level_sizes(self, level):
        n, nf, nc, nnz = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        ad, op = C.c_int(), C.c_int()
        check(lib().airhip_pcair_level_sizes(self._h, level, C.byref(n), C.byref(nf), C.byref(nc), C.byref(ad),
                                             C.byref(op), C.byref(nnz)))
        return dict(n=n.value, n_f=nf.value, n_c=nc.value, aff_diag=bool(ad.value), one_point=bool(op.value),
                    nnz_A=nnz.value)

    def _get(self, level, which, copy=True):
        c = airhip_csr(); idx = C.POINTER(C.c_int32)(); vec = C.POINTER(C.c_double)(); ln = C.c_int64()
        check(lib().airhip_pcair_level_get(self._h, level, which.encode(), C.byref(c), C.byref(idx), C.byref(vec),
                                           C.byref(ln)))
        if which in ("F", "C", "wcol"):
            return _arr(idx, ln.value, copy, np.int32)
        if which in ("aff_diag", "M_ff_diag", "M_cc_diag"):
            return _arr(vec, ln.value, copy, np.float64)
        return _view_csr(c, copy)

    def export_level(self, level, copy=True):
        """The level's operands in the reference's level-local numbering.  copy=False returns
        views into the library's host hierarchy (valid until release_host / next setup)."""
        s = self.level_sizes(level)
        out = dict(s)
        _g = self._get
        self._get = lambda lv, w: _g(lv, w, copy)
        try:
            return self._export_level(level, s, out, copy)
        finally:
            self._get = _g

    def _export_level(self, level, s, out, copy):
        out["F"] = self._get(level, "F"); out["C"] = self._get(level, "C")
        out["A_fc"] = self._get(level, "A_fc"); out["Z"] = self._get(level, "Z")
        out["A_cf"] = self._get(level, "A_cf")
        if s["aff_diag"]:
            out["aff_diag"] = self._get(level, "aff_diag")
        else:
            out["A_ff"] = self._get(level, "A_ff")
        if s["one_point"]:
            out["wcol"] = self._get(level, "wcol")
        else:
            out["W"] = self._get(level, "W")
        try:        # only a hierarchy built with full_smoothing_up_and_down holds the level operator
            out["A_full"] = self._get(level, "A_full")
        except AirhipError:
            pass
        full = "A_full" in out
        inv = airhip_invop()
        check(lib().airhip_pcair_level_inv(self._h, level, 0, C.byref(inv)))
        out["inv_ff"] = _invop_to_dict(inv, s["n"] if full else s["n_f"], copy)   # n x n with full smoothing
        so = (C.c_int * 10)()
        check(lib().airhip_pcair_level_smooth_order(self._h, level, so))
        out["smooth_order"] = [int(v) for v in so]
        if any(v < 0 for v in out["smooth_order"]) and not full:
            out["A_cc"] = self._get(level, "A_cc")
            check(lib().airhip_pcair_level_inv(self._h, level, 1, C.byref(inv)))
            out["inv_cc"] = _invop_to_dict(inv, s["n_c"], copy)
        return out

    def export_coarse(self, copy=True):
        c = airhip_csr()
        check(lib().airhip_pcair_coarse_get(self._h, C.byref(c)))
        A = _view_csr(c, copy)
        inv = airhip_invop()
        check(lib().airhip_pcair_level_inv(self._h, self.num_levels - 1, 0, C.byref(inv)))
        return A, _invop_to_dict(inv, A.shape[0], copy)

    def complexities(self):
        g, o, cy, st, by = (C.c_double() for _ in range(5))
        check(lib().airhip_pcair_complexities(self._h, C.byref(g), C.byref(o), C.byref(cy), C.byref(st), C.byref(by)))
        return dict(grid=g.value, operator=o.value, cycle=cy.value, storage=st.value, cycle_alg_bytes=by.value)

    def cycle_alg_bytes(self):
        b = C.c_double()
        check(lib().airhip_pcair_cycle_alg_bytes(self._h, C.byref(b)))
        return b.value

    def setup_times(self):
        a, b = C.c_double(), C.c_double()
        check(lib().airhip_pcair_setup_times(self._h, C.byref(a), C.byref(b)))
        return dict(host_setup_s=a.value, upload_s=b.value)

    def host_mirror_wait(self):
        """Waits for the thread that completes the host copy of a hierarchy set up on the device (include/airhip.h:
        airhip_pcair_host_mirror_wait); returns the seconds that thread ran (0.0 if there was none)."""
        t = C.c_double(0.0)
        check(lib().airhip_pcair_host_mirror_wait(self._h, C.byref(t)))
        return t.value

    def release_host(self):
        check(lib().airhip_pcair_release_host(self._h))

    def use_graph(self, on=True):
        check(lib().airhip_pcair_use_graph(self._h, int(on)))

    def profile(self, on=True):
        check(lib().airhip_pcair_profile(self._h, int(on)))

    def profile_reset(self):
        check(lib().airhip_pcair_profile_reset(self._h))

    def profile_stats(self):
        n = C.c_int()
        check(lib().airhip_pcair_profile_count(self._h, C.byref(n)))
        out = []
        for i in range(n.value):
            name = C.c_char_p(); lvl = C.c_int(); by = C.c_double(); calls = C.c_int64(); ms = C.c_double()
            check(lib().airhip_pcair_profile_get(self._h, i, C.byref(name), C.byref(lvl), C.byref(by), C.byref(calls),
                                                 C.byref(ms)))
            kern = C.c_char_p()
            check(lib().airhip_pcair_profile_kernel(self._h, i, C.byref(kern)))
            out.append(dict(name=name.value.decode(), level=lvl.value, alg_bytes=by.value, calls=calls.value,
                            total_ms=ms.value, kernel=(kern.value or b"").decode()))
        return out

    def __del__(self):
        try:
            if self._h:
                lib().airhip_pcair_destroy(C.byref(self._h))
        except Exception:
            pass


class PCPFLAREINV:
    """PC type "pflareinv" (src/PCPFLAREINV.c): y = mat_inverse x."""

    def __init__(self, **options):
        self._h = C.c_void_p()
        check(lib().airhip_pcinv_create(C.byref(self._h)))
        for k, v in options.items():
            self.set_option(k, v)

    def set_option(self, name, value):
        if isinstance(value, bool):
            value = "1" if value else "0"
        check(lib().airhip_pcinv_set_option(self._h, name.encode(), str(value).encode()))

    def setup(self, A):
        a = as_csr(A)
        check(lib().airhip_pcinv_setup(self._h, a.ref()))
        return self

    def setup_host_only(self, A):
        a = as_csr(A)
        check(lib().airhip_pcinv_setup_host_only(self._h, a.ref()))
        return self

    def apply(self, x, y):
        check(lib().airhip_pcinv_apply_on_stream(self._h, _ptr(x), _ptr(y), _caller_stream(x)))
        return y

    def get_inverse(self, n):
        inv = airhip_invop()
        check(lib().airhip_pcinv_get_inv(self._h, C.byref(inv)))
        return _invop_to_dict(inv, n)

    def as_pcair(self):
        """The zero-level PCAIR this object is built on, as something pflare_amd.dist.distribute can cut for a
        multi-GPU run (borrowed handle: keeps this object alive, is never destroyed itself)."""
        h = C.c_void_p()
        check(lib().airhip_pcinv_as_pcair(self._h, C.byref(h)))
        view = type("_BorrowedPCAIR", (), {})()
        view._h = h
        view._owner = self
        return view

    def __del__(self):
        try:
            if self._h:
                lib().airhip_pcinv_destroy(C.byref(self._h))
        except Exception:
            pass


class DeviceMat:
    """Device-resident AIJ matrix (row-block CSR) with MatMult."""

    def __init__(self, A):
        self._h = C.c_void_p()
        a = as_csr(A)
        check(lib().airhip_mat_create(a.ref(), C.byref(self._h)))
        self.shape = a.shape

    def mult(self, x, y):
        check(lib().airhip_mat_mult(self._h, _ptr(x), _ptr(y)))
        return y

    def blocks(self):
        """diagnostic: the row-block form as it lies on the device (airhip_mat_blocks): descriptors (structured array),
        index regions (bytes), values, row offsets"""
        nb, ib, nz = C.c_int64(), C.c_int64(), C.c_int64()
        check(lib().airhip_mat_blocks(self._h, C.byref(nb), C.byref(ib), C.byref(nz), None, None, None, None))
        desc = np.zeros(nb.value, dtype=np.dtype([("p0", "<i8"), ("row", "<i4"), ("cbase", "<i4"), ("nb", "<i4"), ("nr", "<i4"), ("ip", "<i8")]))
        idx = np.zeros(max(ib.value, 1), dtype=np.uint8); val = np.zeros(max(nz.value, 1)); roff = np.zeros(max(self.shape[0], 1), dtype=np.uint16)
        check(lib().airhip_mat_blocks(self._h, C.byref(nb), C.byref(ib), C.byref(nz), desc.ctypes.data, idx.ctypes.data, val.ctypes.data,
                                      roff.ctypes.data))
        return desc, idx[:ib.value], val[:nz.value], roff[:self.shape[0]]

    def mult_async(self, x, y):
        """y = A x ordered on torch's current stream, no host synchronisation"""
        check(lib().airhip_mat_mult_on_stream(self._h, _ptr(x), _ptr(y), _caller_stream(x)))
        return y

    def __del__(self):
        try:
            if self._h:
                lib().airhip_mat_destroy(C.byref(self._h))
        except Exception:
            pass


KSP_REASONS = {2: "CONVERGED_RTOL", 3: "CONVERGED_ATOL", -3: "DIVERGED_ITS", -4: "DIVERGED_DTOL",
               -5: "DIVERGED_BREAKDOWN", -9: "DIVERGED_NANORINF", 0: "ITERATING"}


class KSP:
    """PETSc-style outer Krylov solve on the device: -ksp_type richardson|gmres, -ksp_pc_side,
    -ksp_norm_type, -ksp_rtol/-ksp_atol/-ksp_divtol, -ksp_max_it, -ksp_gmres_restart."""

    def __init__(self, ksp_type="gmres", pc_side="left", norm_type="preconditioned", rtol=1e-5, atol=1e-50,
                 divtol=1e4, max_it=10000, restart=30):
        o = airhip_ksp_options()
        lib().airhip_ksp_default_options(C.byref(o))
        o.type = {"richardson": 0, "gmres": 1}[ksp_type]
        o.pc_side = {"left": 0, "right": 1}[pc_side]
        o.norm_type = {"preconditioned": 0, "unpreconditioned": 1}[norm_type]
        o.rtol, o.atol, o.dtol, o.max_it, o.restart = rtol, atol, divtol, max_it, restart
        self.o = o

    def solve(self, b, x, pc=None, A=None):
        """Solve A x = b; x holds the initial guess and receives the solution."""
        res = airhip_ksp_result()
        hist = (C.c_double * 4096)()
        hl = C.c_int()
        pcair = pc._h if isinstance(pc, PCAIR) else None
        pcinv = pc._h if isinstance(pc, PCPFLAREINV) else None
        amat = A._h if A is not None else None
        check(lib().airhip_ksp_solve_on_stream(C.byref(self.o), amat, pcair, pcinv, _ptr(b), _ptr(x), C.byref(res), hist,
                                               4096, C.byref(hl), _caller_stream(x)))
        return dict(its=res.its, reason=res.reason, reason_name=KSP_REASONS.get(res.reason, str(res.reason)),
                    rnorm=res.rnorm, solve_ms=res.solve_ms, pc_applies=res.pc_applies, history=[hist[i] for i in range(min(hl.value, 4096))])
