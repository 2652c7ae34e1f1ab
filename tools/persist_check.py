"""Persistent-grid trailing update (option update_wgs) against the one-workgroup-per-tile launch: factors must be bit-identical."""
import os
os.environ["SIGP_USE_DEBUG_LIB"] = "1"      # the switches below are measurement switches of libsigp_debug.so (make debug)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR

def factor(n, W, wgs, thr=0, la=1):
    X, y, Xs = O.synthetic_problem(n, 8, 5, m=1)
    with GPR(kernel="rbf", outer_blocks=W, lookahead=la) as gp:
        gp.set_option("small_tile_threshold", thr)
        gp.set_option("update_wgs", wgs)
        try:
            gp.fit(X, y, np.sqrt(8.0), 1e-2, Xs=Xs)
        except Exception as e:
            return None, str(e)
        return gp.L_tilde_, gp.nlml_

for n, W in ((2100, 4), (4100, 8)):
    L0, nl0 = factor(n, W, 0)
    for la in (1, 0):
        for wgs in (1, 1, 2, 2, 3):
            L1, nl1 = factor(n, W, wgs, la=la)
            if L1 is None:
                print("n=%d W=%d la=%d wgs=%d: EXC %s" % (n, W, la, wgs, nl1), flush=True); continue
            d = np.abs(L1 - L0)
            bi, bj = np.nonzero(d.reshape(L0.shape[0] // 1, -1) > 0) if False else (None, None)
            nb = (L0.shape[0] + 127) // 128
            blocks = [(i, j) for i in range(nb) for j in range(i + 1) if d[i * 128:(i + 1) * 128, j * 128:(j + 1) * 128].max() > 0]
            print("n=%d W=%d la=%d wgs=%d: max diff %.3e, bad blocks (%d) %s" % (n, W, la, wgs, d.max(), len(blocks), blocks[:10]), flush=True)
