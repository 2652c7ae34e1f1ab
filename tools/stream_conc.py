"""Do kernels on different HIP streams overlap on this box? (debug entry sigp_debug_stream_concurrency)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seaiceextentforecasting_amd import _lib as L
lib = L.load(debug=True)   # libsigp_debug.so (make -C seaiceextentforecasting_amd/csrc debug)
f = lib.sigp_debug_stream_concurrency
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, L._dp]
h = C.c_void_p(); assert lib.sigp_create(C.byref(h), 0, 0) == 0
ms = C.c_double()
for slot_streams in (0, 1):
    for ns in (1, 2, 3, 4, 6, 8):
        f(h, ns, 20, 16, 4000, slot_streams, C.byref(ms))     # 16 blocks x ~110 us each
        print("slot_streams=%d nstreams=%d  wall %.2f ms  (%.3f ms per launch-round)" % (slot_streams, ns, ms.value, ms.value / 20))
