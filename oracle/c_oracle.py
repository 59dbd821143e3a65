"""ctypes binding of oracle/libair_oracle.so (the plain-C solve-phase restatement).
TEST INFRASTRUCTURE ONLY - see the header of air_oracle.c."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libair_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "air_oracle.c")):
            build()
        _lib = C.CDLL(_LIB)
        _lib.oc_create.restype = C.c_void_p
        _lib.oc_create.argtypes = [C.c_int]
        _lib.oc_destroy.argtypes = [C.c_void_p]
        _lib.oc_vcycle.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.oc_set_threads.argtypes = [C.c_int]
        _lib.oc_set_threads.restype = C.c_int
    return _lib


class CHierarchy:
    """Builds the C oracle's hierarchy from the product's exported levels
    (pflare_amd.PCAIR.export_level dicts) - natural level-local numbering."""

    def __init__(self, pc, copy=True):
        from pflare_amd._lib import airhip_csr, airhip_invop   # struct layouts only (identical to oc_csr / oc_inv)
        from pflare_amd.pc import _InvHolder, as_csr
        L = lib()
        self.keep = []
        nl = pc.num_levels
        self.h = C.c_void_p(L.oc_create(nl))
        self.n = None
        for l in range(nl - 1):
            e = pc.export_level(l, copy=copy)
            if self.n is None:
                self.n = e["n"]
            F = np.ascontiguousarray(e["F"], np.int32); Cc = np.ascontiguousarray(e["C"], np.int32)
            self.keep += [F, Cc]

            def ref(key):
                if e.get(key) is None:
                    return None
                m = as_csr(e[key]); self.keep.append(m)
                return C.byref(m.c)

            affd = None
            if e.get("aff_diag") is not None and e.get("A_ff") is None:
                a = np.ascontiguousarray(e["aff_diag"], np.float64); self.keep.append(a)
                affd = a.ctypes.data_as(C.POINTER(C.c_double))
            wcol = None
            if e.get("wcol") is not None:
                w = np.ascontiguousarray(e["wcol"], np.int32); self.keep.append(w)
                wcol = w.ctypes.data_as(C.POINTER(C.c_int32))
            iff = _InvHolder(e["inv_ff"]); self.keep.append(iff)
            icc = _InvHolder(e["inv_cc"]) if e.get("inv_cc") is not None else None
            if icc:
                self.keep.append(icc)
            so = (C.c_int * 10)(*e["smooth_order"])
            rc = L.oc_set_level(self.h, l, C.c_int64(e["n"]), C.c_int64(F.size), C.c_int64(Cc.size),
                                F.ctypes.data_as(C.c_void_p), Cc.ctypes.data_as(C.c_void_p),
                                ref("A_ff"), affd, ref("A_fc"), C.byref(iff.c), ref("Z"), wcol, ref("W"),
                                ref("A_cf"), ref("A_cc"), C.byref(icc.c) if icc else None, so)
            assert rc == 0, rc
        cA, cinv = pc.export_coarse(copy=copy)
        ci = _InvHolder(cinv); self.keep += [cA, ci]
        if self.n is None:
            self.n = cA.shape[0]
        self.coarse_its = 0
        L.oc_set_coarse(self.h, C.byref(cA.c), C.byref(ci.c), 0)

    def set_coarse_richardson(self, its):
        # re-register the coarse level with N richardson steps
        cA, ci = self.keep[-2], self.keep[-1]
        lib().oc_set_coarse(self.h, C.byref(cA.c), C.byref(ci.c), int(its))

    def vcycle(self, b):
        b = np.ascontiguousarray(b, np.float64)
        x = np.empty_like(b)
        lib().oc_vcycle(self.h, b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p))
        return x

    def ksp(self, A, b, x0, ksp_type="gmres", pc_side="left", norm_type="preconditioned", rtol=1e-5, atol=1e-50,
            divtol=1e4, max_it=10000, restart=30, use_pc=True):
        from pflare_amd.pc import as_csr
        a = as_csr(A)
        b = np.ascontiguousarray(b, np.float64); x = np.array(x0, dtype=np.float64, copy=True)
        its = C.c_int(); reason = C.c_int(); rn = C.c_double(); hl = C.c_int()
        hist = (C.c_double * 4096)()
        lib().oc_ksp_solve(self.h if use_pc else None, C.byref(a.c), {"richardson": 0, "gmres": 1}[ksp_type],
                           {"left": 0, "right": 1}[pc_side], {"preconditioned": 0, "unpreconditioned": 1}[norm_type],
                           C.c_double(rtol), C.c_double(atol), C.c_double(divtol), int(max_it), int(restart),
                           b.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), C.byref(its), C.byref(reason),
                           C.byref(rn), hist, 4096, C.byref(hl))
        return x, its.value, reason.value, rn.value, [hist[i] for i in range(min(hl.value, 4096))]

    def __del__(self):
        try:
            lib().oc_destroy(self.h)
        except Exception:
            pass


def set_threads(n):
    return lib().oc_set_threads(int(n))
