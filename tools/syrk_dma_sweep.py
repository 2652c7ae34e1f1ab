"""What the operand DMA costs the trailing-update kernel as a function of the operand footprint: debug entry sigp_debug_time_syrk,
fp32 and fp64, rt x rt tile triangles at K = 1024 / 4096, with (dbg 0) and without (dbg 1) the in-loop global->LDS DMA."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seaiceextentforecasting_amd import _lib as L
lib = L.load(debug=True)
lib.sigp_debug_time_syrk.restype = C.c_int
lib.sigp_debug_time_syrk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, L._dp, L._dp, C.c_int, L._dp]
h = C.c_void_p(); assert lib.sigp_create(C.byref(h), 0, 0) == 0
ms, tf, ghz = C.c_double(), C.c_double(), C.c_double()
for name, small, es in (("fp32", 2 + 16, 4), ("fp64", 2, 8)):
    for K in (1024, 4096):
        for rt in ((31, 45, 63, 90, 127) if len(sys.argv) < 2 else [int(a) for a in sys.argv[1:]]):
            row = []
            for dbg in (0, 1, 0, 1):
                lib.sigp_debug_time_syrk(h, rt, K, 0, small, 3, C.byref(ms), C.byref(tf), dbg, C.byref(ghz))
                row.append((tf.value, ghz.value))
            print("%s K=%4d rt=%3d (%5d tiles, operands %6.1f MB): DMA on %6.1f / %6.1f  off %6.1f / %6.1f TFLOP/s   clocks %.2f %.2f" % (
                name, K, rt, rt * (rt + 1) // 2, rt * 128 * K * es / 1e6, row[0][0], row[2][0], row[1][0], row[3][0], row[0][1], row[1][1]), flush=True)
