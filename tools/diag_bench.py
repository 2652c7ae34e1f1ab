"""Micro-benchmark of the diagonal-block kernel with phases switched off (debug entry sigp_debug_time_diag)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seaiceextentforecasting_amd import _lib as L
lib = L.load(debug=True)   # libsigp_debug.so (make -C seaiceextentforecasting_amd/csrc debug)
lib.sigp_debug_time_diag.restype = C.c_int
lib.sigp_debug_time_diag.argtypes = [C.c_void_p, L._dp, C.c_int, C.c_int, L._dp, L._dp, L._dp]
h = C.c_void_p(); assert lib.sigp_create(C.byref(h), 0, 0) == 0
rng = np.random.default_rng(0)
B = rng.standard_normal((128, 128)); A = B @ B.T + 128 * np.eye(128)
ms = C.c_double(); Lo = np.zeros((128, 128)); Li = np.zeros((128, 128))
for skip, name in {0: "full", 128: "full, roles dealt as if wave w sat on SIMD w mod 4", 32: "full, no wave priority", 1: "no pivot loops", 2: "no MFMA-wave work", 4: "no tile inverse (wave 7)", 8: "no pivot-wave update", 16: "no global loads / stores", 1 | 8: "no pivot-wave work", 2 | 4: "pivot waves only", 2 | 4 | 16: "pivot waves only, no global traffic", 31: "barriers only"}.items():
    rc = lib.sigp_debug_time_diag(h, L.ptr(A), skip, 400, C.byref(ms), L.ptr(Lo), L.ptr(Li))
    print("flags=%3d %-52s %8.1f us" % (skip, name, ms.value * 1e3))
rc = lib.sigp_debug_time_diag(h, L.ptr(A), 0, 3, C.byref(ms), L.ptr(Lo), L.ptr(Li))
Lr = np.linalg.cholesky(A)
print("L err %.2e  Linv err %.2e" % (np.abs(np.tril(Lo) - Lr).max() / np.abs(Lr).max(), np.abs(np.tril(Li) - np.linalg.inv(Lr)).max() / np.abs(np.linalg.inv(Lr)).max()))
