import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
from seaiceextentforecasting_amd import _lib as L
lib = L.load(debug=True)   # libsigp_debug.so (make -C seaiceextentforecasting_amd/csrc debug)
lib.sigp_debug_time_syrk.restype = C.c_int
lib.sigp_debug_time_syrk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, L._dp, L._dp, C.c_int, L._dp]
h = C.c_void_p(); assert lib.sigp_create(C.byref(h), 0, 0) == 0
ms, tf, ghz = C.c_double(), C.c_double(), C.c_double()
for patch in (0, 8):
    lib.sigp_debug_time_syrk(h, 40, 256, patch, 2, 1, C.byref(ms), C.byref(tf), 32, C.byref(ghz))
    print("patch", patch, tf.value)
