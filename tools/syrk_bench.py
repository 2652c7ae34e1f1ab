"""Micro-benchmark of the trailing-update kernel (debug entry sigp_debug_time_syrk) + fp64 MFMA rate probe."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seaiceextentforecasting_amd import _lib as L
lib = L.load(debug=True)   # libsigp_debug.so (make -C seaiceextentforecasting_amd/csrc debug)
lib.sigp_debug_time_syrk.restype = C.c_int
lib.sigp_debug_time_syrk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, L._dp, L._dp, C.c_int, L._dp]
lib.sigp_debug_mfma_peak.restype = C.c_int
lib.sigp_debug_mfma_peak.argtypes = [C.c_void_p, C.c_int, C.c_int, L._dp, C.c_double]
h = C.c_void_p(); assert lib.sigp_create(C.byref(h), 0, 0) == 0
lib.sigp_set_option(h, b"reserve_cus", 0)
ms, tf, ghz = C.c_double(), C.c_double(), C.c_double()
for seed in (0.25, -1.0):
    lib.sigp_debug_mfma_peak(h, 512, 20000, C.byref(tf), seed)
    print("mfma rate probe, %s operands: %.1f TFLOP/s" % ("constant" if seed > 0 else "random", tf.value))
mode = sys.argv[1] if len(sys.argv) > 1 else "ablate"
if mode == "ablate":
    print("rt   K small dbg |    ms   TFLOP/s")
    for rt, K in ((62, 1024), (62, 2048)):
        for small in (2,):
            for dbg in (0, 1, 0, 1):
                lib.sigp_debug_time_syrk(h, rt, K, 0, small, 5, C.byref(ms), C.byref(tf), dbg, C.byref(ghz))
                print("%2d %4d %5d %3d | %6.3f  %6.1f   in-kernel clock %.2f GHz" % (rt, K, small, dbg, ms.value, tf.value, ghz.value))
elif mode == "ablate2":
    # dbg bits: 1 no in-loop DMA, 2 no in-loop fragment reads, 512 no MFMA (compile-time variants of the kernel), 8 no C load,
    # 16 no C store  (timing only; results are garbage)
    print("rt   K dbg |    ms   TFLOP/s  clock")
    for rt, K in ((90, 1024), (90, 4096)):
        for dbg in (0, 1, 2, 3, 24, 25, 27, 0):
            lib.sigp_debug_time_syrk(h, rt, K, 0, 2, 4, C.byref(ms), C.byref(tf), dbg, C.byref(ghz))
            print("%2d %4d %3d | %6.3f  %6.1f   %.2f GHz" % (rt, K, dbg, ms.value, tf.value, ghz.value), flush=True)
elif mode == "f32":
    # the fp32 instantiation (small = 2 + 16): dbg 1 no DMA, 2 no fragment reads, 8+16 no C traffic
    for rt, K in ((127, 1024), (127, 4096)):
        for dbg in (0, 1, 2, 3, 24, 256):
            lib.sigp_debug_time_syrk(h, rt, K, 0, 2 + 16, 3, C.byref(ms), C.byref(tf), dbg, C.byref(ghz))
            print("fp32 %3d %4d dbg %3d | %6.3f ms  %6.1f TFLOP/s  %.2f GHz" % (rt, K, dbg, ms.value, tf.value, ghz.value), flush=True)
elif mode == "cdma":
    # dbg 128: C-tile prologue through LDS-DMA instead of 64 accumulator-layout loads per lane
    for rt, K in ((127, 1024), (127, 512), (127, 256)):
        for dbg in (256, 256 + 128, 256, 256 + 128):
            lib.sigp_debug_time_syrk(h, rt, K, 0, 2, 3, C.byref(ms), C.byref(tf), dbg, C.byref(ghz))
            print("%3d %4d dbg %4d | %6.3f ms  %6.1f TFLOP/s  %.2f GHz" % (rt, K, dbg, ms.value, tf.value, ghz.value), flush=True)
elif mode == "phases":
    # dbg 256: in-kernel cycle counts of the three phases of a tile (printed on stderr by the debug entry)
    for rt, K in ((127, 1024), (127, 512), (127, 128), (90, 4096)):
        for dbg in (256, 256 + 1, 256 + 8 + 16):
            lib.sigp_debug_time_syrk(h, rt, K, 0, 2, 3, C.byref(ms), C.byref(tf), dbg, C.byref(ghz))
            print("%3d %4d dbg %3d | %6.3f ms  %6.1f TFLOP/s  %.2f GHz" % (rt, K, dbg, ms.value, tf.value, ghz.value), flush=True)
else:
    rt, K, small = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    lib.sigp_debug_time_syrk(h, rt, K, 0, small, 3, C.byref(ms), C.byref(tf), 0, None)
    print("%2d %4d %5d | %6.3f  %6.1f" % (rt, K, small, ms.value, tf.value))
