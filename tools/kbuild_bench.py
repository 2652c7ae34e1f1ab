"""Ablation timing of the covariance build (debug entry sigp_debug_time_kbuild): distances on the VALU (kbuild_kernel) and in GEMM form (kbuild_mfma_kernel)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seaiceextentforecasting_amd import _lib as L
lib = L.load(debug=True)   # libsigp_debug.so (make -C seaiceextentforecasting_amd/csrc debug)
lib.sigp_debug_time_kbuild.restype = C.c_int
lib.sigp_debug_time_kbuild.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, L._dp]
h = C.c_void_p(); assert lib.sigp_create(C.byref(h), 0, 0) == 0
ms = C.c_double()
lib.sigp_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
for n, d, nb in ((8192, 8, 40), (8192, 8, 160)) if quick else ((8192, 8, 40), (8192, 8, 160), (8192, 32, 8)):
    for kid, kname in ((1, "rbf"),) if quick else ((1, "rbf"), (2, "matern52")):
        for mf in (0, 1):   # mfma 0: distances on the VALU, 1: GEMM form on the matrix pipe
            assert lib.sigp_set_option(h, b"kbuild_mfma", mf) == 0
            for flags in (0, 2, 4, 6):         # 2: no covariance function, 4: no store
                rc = lib.sigp_debug_time_kbuild(h, n, d, nb, kid, flags, 5, C.byref(ms))
                assert rc == 0, rc
                byt = nb * (4.0 * n * (n + 1) + 8.0 * n * d)
                print("n=%d d=%2d nb=%3d %-8s mfma=%d flags=%d : %7.3f ms  %6.3f ms/member  %5.2f TB/s (lower-triangle bytes)" % (n, d, nb, kname, mf, flags, ms.value, ms.value / nb, byt / ms.value / 1e9), flush=True)
