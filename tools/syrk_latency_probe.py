"""The K loop of the trailing-update kernel without its MFMAs (debug flag 512): the period per K-slice that the operand stream alone
allows -- LDS-DMA issue, its latency behind one barrier per slice, fragment reads -- against the matrix pipe's 4096 cycles per wave."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seaiceextentforecasting_amd import _lib as L
lib = L.load(debug=True)
lib.sigp_debug_time_syrk.restype = C.c_int
lib.sigp_debug_time_syrk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, L._dp, L._dp, C.c_int, L._dp]
h = C.c_void_p(); assert lib.sigp_create(C.byref(h), 0, 0) == 0
ms, tf, ghz = C.c_double(), C.c_double(), C.c_double()
for name, small, kt in (("fp32", 2 + 16, 32), ("fp64", 2, 16)):
    for K in (1024, 4096):
        for rt in (31, 127):
            tiles = rt * (rt + 1) // 2
            rounds = -(-tiles // 512)
            for dbg, what in ((0, "full"), (512, "no MFMA"), (512 + 2, "no MFMA, no fragment reads"), (512 + 1, "no MFMA, no DMA"), (512 + 8 + 16, "no MFMA, no C traffic")):
                lib.sigp_debug_time_syrk(h, rt, K, 0, small, 3, C.byref(ms), C.byref(tf), dbg, C.byref(ghz))
                per_slice_us = ms.value * 1e3 / rounds / (K // kt)
                print("%s K=%4d rt=%3d %-28s %7.3f ms  = %5.2f us per K-slice per resident workgroup (%d rounds of 512)  clock %.2f" % (
                    name, K, rt, what, ms.value, per_slice_us, rounds, ghz.value), flush=True)
