"""Single-fit latency against the 32x32-tile tier of the inner updates (option tiny_tile_threshold)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR
for n, d in ((2048, 8), (4096, 8), (8192, 8), (16384, 16)):
    X, y, Xs = O.synthetic_problem(n, d, 20240000, m=1)
    ref = None
    for thr in (0, 256, 512, 1024, 2048, 4096, 0):
        with GPR(kernel="rbf") as gp:
            gp.set_option("tiny_tile_threshold", thr)
            gp.fit(X, y, np.sqrt(d), 1e-2, Xs=Xs)
            nl = gp.nlml_
            ref = nl if ref is None else ref
            t = time.perf_counter()
            for _ in range(4):
                gp.refit(np.sqrt(d), 1e-2)
            dt = (time.perf_counter() - t) / 4
        print("n=%5d tiny<=%4d: %6.2f ms  (nlml identical: %s)" % (n, thr, dt * 1e3, nl == ref), flush=True)
