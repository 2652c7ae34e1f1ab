"""Single-fit latency against outer panel width and panel mode (GPR.fit on one handle, whole GPU)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR

for n, d in ((2048, 8), (4096, 8), (8192, 8), (16384, 16)):
    X, y, Xs = O.synthetic_problem(n, d, 20240000, m=1)
    for mode in ("recursive", "strips"):
        for W in (1, 2, 4, 8, 16):
            with GPR(kernel="rbf", outer_blocks=W, panel_mode=mode) as gp:
                gp.fit(X, y, np.sqrt(d), 1e-2, Xs=Xs)
                reps = 5 if n <= 8192 else 3
                t = time.perf_counter()
                for _ in range(reps):
                    gp.refit(np.sqrt(d), 1e-2)
                dt = (time.perf_counter() - t) / reps
            print("n=%6d %-9s outer=%2d : %7.2f ms/fit  %5.1f TFLOP/s" % (n, mode, W, dt * 1e3, n ** 3 / 3 / dt / 1e12), flush=True)
