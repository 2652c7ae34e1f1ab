"""Which CUs does a hipExtStreamCreateWithCUMask mask enable?  Runs a many-block spin kernel and lists (xcc, se, cu)."""
import ctypes as C, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from seaiceextentforecasting_amd import _lib as L
lib = L.load(debug=True)   # libsigp_debug.so (make -C seaiceextentforecasting_amd/csrc debug)
f = lib.sigp_debug_cumask_probe
f.restype = C.c_int; f.argtypes = [C.c_void_p, C.POINTER(C.c_uint), C.c_int, C.POINTER(C.c_uint)]
h = C.c_void_p(); assert lib.sigp_create(C.byref(h), 0, 0) == 0
B = 2048
def probe(bits, label):
    m = (C.c_uint * 8)()
    for b in bits: m[b // 32] |= (1 << (b % 32))
    out = (C.c_uint * (2 * B))()
    rc = f(h, m, B, out)
    cus = collections.Counter()
    for i in range(B):
        xcc, hw = out[2 * i] & 0xf, out[2 * i + 1]
        cus[(xcc, (hw >> 13) & 7, (hw >> 8) & 0xf)] += 1
    perx = collections.Counter(k[0] for k in cus)
    print("%-28s rc=%d distinct CUs=%3d per-xcc=%s" % (label, rc, len(cus), dict(sorted(perx.items()))))
    return cus
probe(range(256), "all 256")
probe(range(32), "bits 0-31")
probe(range(32, 64), "bits 32-63")
probe(range(0, 256, 8), "bits = 0 mod 8")
probe(range(1, 256, 8), "bits = 1 mod 8")
probe(range(0, 128), "bits 0-127")
probe(range(0, 254), "bits 0-253")
probe(range(2, 256), "bits 2-255")
c = probe([0], "bit 0"); print(sorted(c))
c = probe([8], "bit 8"); print(sorted(c))
c = probe([1], "bit 1"); print(sorted(c))
c = probe([0, 1, 2, 3, 4, 5, 6, 7], "bits 0-7"); print(sorted(c))
