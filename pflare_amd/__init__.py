"""pflare_amd - MI355X-native PCAIR / PCPFLAREINV solve phase.

Only what the hot path needs: `csrc/` (HIP kernels, host setup, C ABI -> libairhip.so) and the
host-side mirror of the reference's plugin interface (`pc.py`).
"""
from ._lib import AirhipError, LIB_PATH, check, lib  # noqa: F401
from .pc import (CF_CR, CF_DIAG_DOM, CF_PMISR_DDC, KSP, PCAIR, PCPFLAREINV, CsrArrays, DeviceMat, HostMat,  # noqa: F401
                 as_csr, compute_cf_splitting, compute_diag_dom_submatrix)
