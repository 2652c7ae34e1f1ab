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
names = {0: "full", 8: "no inverse (factor only)", 8 | 1: "factor w/o B1", 8 | 2: "factor w/o B2", 8 | 4: "factor w/o B3",
         8 | 7: "load/store + barriers only", 16: "full w/o inverse block loop", 7: "inverse only (no factor phases)"}
for skip, name in names.items():
    rc = lib.sigp_debug_time_diag(h, L.ptr(A), skip, 400, C.byref(ms), L.ptr(Lo), L.ptr(Li))
    print("skip=%2d %-34s %8.1f us" % (skip, name, ms.value * 1e3))
rc = lib.sigp_debug_time_diag(h, L.ptr(A), 0, 3, C.byref(ms), L.ptr(Lo), L.ptr(Li))
Lr = np.linalg.cholesky(A)
print("L err %.2e  Linv err %.2e" % (np.abs(np.tril(Lo) - Lr).max() / np.abs(Lr).max(), np.abs(np.tril(Li) - np.linalg.inv(Lr)).max() / np.abs(np.linalg.inv(Lr)).max()))
