"""fp64 MFMA issue-rate probe (debug entry sigp_debug_mfma_peak): what the chip sustains on pure MFMA."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seaiceextentforecasting_amd import _lib as L
lib = L.load(debug=True)   # libsigp_debug.so (make -C seaiceextentforecasting_amd/csrc debug)
lib.sigp_debug_mfma_peak.restype = C.c_int
lib.sigp_debug_mfma_peak.argtypes = [C.c_void_p, C.c_int, C.c_int, L._dp]
h = C.c_void_p(); assert lib.sigp_create(C.byref(h), 0, 0) == 0
tf = C.c_double()
for blocks, iters in ((256, 2000), (512, 2000), (1024, 2000), (512, 20000)):
    lib.sigp_debug_mfma_peak(h, blocks, iters, C.byref(tf))
    print("blocks=%d iters=%d  %.1f TFLOP/s fp64 MFMA" % (blocks, iters, tf.value))
