"""XCD-chunked tile walk (option xcd_chunks) against the column-major walk: placement only, so factors must be bit-identical."""
import os
os.environ["SIGP_USE_DEBUG_LIB"] = "1"      # the switches below are measurement switches of libsigp_debug.so (make debug)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR

def factor(n, W, P):
    X, y, Xs = O.synthetic_problem(n, 8, 5, m=1)
    with GPR(kernel="rbf", outer_blocks=W) as gp:
        gp.set_option("xcd_chunks", P)
        gp.fit(X, y, np.sqrt(8.0), 1e-2, Xs=Xs)
        return gp.L_tilde_, gp.nlml_

for n, W in ((4100, 2), (8192, 8), (6000, 4)):
    L0, nl0 = factor(n, W, 0)
    for P in (8, 4, 5):
        L1, nl1 = factor(n, W, P)
        print("n=%d W=%d P=%d: identical %s (max diff %.2e)" % (n, W, P, np.array_equal(L0, L1) and nl0 == nl1, np.abs(L1 - L0).max()), flush=True)
n, d, B = 2100, 8, 12
Xb = np.zeros((B, n, d)); yb = np.zeros((B, n)); Xsb = np.zeros((B, 1, d))
for b in range(B):
    Xb[b], yb[b], Xsb[b] = O.synthetic_problem(n, d, 70 + b, m=1)
ell = np.full(B, 2.5); sn = np.logspace(-2, -1, B)
res = []
for P in (0, 8):
    with GPR(kernel="rbf", outer_blocks=4) as gp:
        gp.set_option("xcd_chunks", P)
        res.append(gp.fit_batch(Xb, yb, Xsb, ell, sn, concurrency=1, group=B))
print("batch identical:", all(np.array_equal(res[0][k], res[1][k]) for k in ("nlml", "mean", "var", "sigma_f")))
