"""Single-fit latency against the outer panel width (GPR.fit on one handle, whole GPU): best of `reps` refits per width, alternating over the
widths `rounds` times so that clock drift does not favour one.  usage: single_sweep.py [n ...]   (default 2048 4096 8192)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR

sizes = [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192]
for n in sizes:
    d = 8 if n <= 8192 else 16
    X, y, Xs = O.synthetic_problem(n, d, 20240000, m=1)
    T = (n + 127) // 128
    widths = [w for w in (4, 8, 12, 16, 24, 32, 48, 64) if w <= max(8, T)]
    best = {w: 1e9 for w in widths}
    ref = None
    for rnd in range(3):
        for W in widths:
            with GPR(kernel="rbf", outer_blocks=W) as gp:
                gp.fit(X, y, np.sqrt(d), 1e-2, Xs=Xs)
                if ref is None:
                    ref = (gp.nlml_, gp.sigma_f_)
                assert abs(gp.nlml_ - ref[0]) <= 1e-10 * abs(ref[0]) and abs(gp.sigma_f_ - ref[1]) <= 1e-10 * abs(ref[1])     # (K of the updates differs with the width: last bits)
                for _ in range(12 if n <= 4096 else 5):
                    t = time.perf_counter()
                    gp.refit(np.sqrt(d), 1e-2)
                    best[W] = min(best[W], time.perf_counter() - t)
    print("n = %5d (%d block columns): " % (n, T) + "  ".join("W=%d %.3f ms" % (w, best[w] * 1e3) for w in widths), flush=True)
