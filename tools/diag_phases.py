"""Where one diagonal-block kernel (128 x 128) spends its time: wall-clock stamps written by its first wave at every barrier (debug entry sigp_debug_diag_stamps)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seaiceextentforecasting_amd import _lib as L
lib = L.load(debug=True)
lib.sigp_debug_diag_stamps.restype = C.c_int
lib.sigp_debug_diag_stamps.argtypes = [C.c_void_p, L._dp, L._dp, C.c_int]
h = C.c_void_p(); assert lib.sigp_create(C.byref(h), 0, 0) == 0
rng = np.random.default_rng(0)
B = rng.standard_normal((128, 128)); A = B @ B.T + 128 * np.eye(128)
out = np.zeros(256)
many = len(sys.argv) > 1 and sys.argv[1] == "chain"          # the last of 400 back-to-back launches instead of a lone one
assert lib.sigp_debug_diag_stamps(h, L.ptr(A), L.ptr(out), 256) == 0 if many else lib.sigp_debug_diag_stamps(h, L.ptr(A), L.ptr(out), 54) == 0
print("%s launch; shader clock over the kernel body: %.3f GHz (s_memtime / s_memrealtime)" % ("last of 400 back-to-back" if many else "lone", out[48]))
print("load + first barrier            : %6.2f us" % out[1])
print("pivot column 0                  : %6.2f us (wave 0 done), barrier released at %.2f" % (out[2] - out[1], out[3]))
prev = out[3]
for s in range(8):
    b1, upd, b2, piv, end = out[4 + 5 * s: 9 + 5 * s]
    print("slot %d: opening barrier %5.2f | column update %5.2f | barrier %5.2f | pivots of column %d %5.2f | closing barrier %5.2f | slot total %5.2f"
          % (s, b1 - prev, upd - b1, b2 - upd, s + 1, (piv - b2) if piv > 0 else float('nan'), end - (piv if piv > 0 else b2), end - prev))
    prev = end
b2 = out[6]
print("inside the pivots of column 1 (wave 0): LDS reads of the tile rows %.2f | 16-pivot loop %.2f | 16 inverse square roots + scaling %.2f | LDS writes %.2f us" % (out[50] - b2, out[51] - out[50], out[52] - out[51], out[53] - out[52]))
print("kernel body end                 : %6.2f us" % out[44])

if many:
    print("other roles, microseconds after the slot's middle barrier (wave 0's view of it): arrival at the closing barrier")
    for s_ in range(8):
        mid = out[6 + 5 * s_]
        w7 = out[64 + 8 * s_: 64 + 8 * s_ + 5] - mid
        w2 = out[128 + 8 * s_: 128 + 8 * s_ + 5] - mid
        w1 = out[192 + 8 * s_: 192 + 8 * s_ + 5] - mid
        piv = out[7 + 5 * s_] - mid
        f = lambda v: " ".join("%5.2f" % x if x > -1e3 else "    -" for x in v)
        print("slot %d: pivot wave done %5.2f | wave 7: start, inverse in LDS, flag, stores issued, at barrier: %s | MFMA wave 2: start, updates done, sums done, flag seen, at barrier: %s | wave 1: %s" % (s_, piv, f(w7), f(w2), f(w1)))
    print("head: roles known / arrival at the barrier after the first pivot column (us since kernel start): wave 7 %.2f / %.2f, wave 2 %.2f / %.2f, wave 1 %.2f / %.2f" % (out[64 + 6], out[64 + 5], out[128 + 6], out[128 + 5], out[192 + 6], out[192 + 5]))
