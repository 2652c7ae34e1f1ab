"""Single-fit latency under engine options: single_opts.py n [name=value ...] (several option sets separated by '/')."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR
n = int(sys.argv[1]); d = 16 if n >= 16384 else 8
sets = " ".join(sys.argv[2:]).split("/") if len(sys.argv) > 2 else [""]
X, y, Xs = O.synthetic_problem(n, d, 20240000, m=1)
for optset in sets:
    with GPR(kernel="rbf") as gp:
        for kv in optset.split():
            k, v = kv.split("="); gp.set_option(k, int(v))
        gp.fit(X, y, np.sqrt(d), 1e-2, Xs=Xs)
        reps = 5 if n <= 8192 else 3
        t = time.perf_counter()
        for _ in range(reps):
            gp.refit(np.sqrt(d), 1e-2)
        dt = (time.perf_counter() - t) / reps
    print("n=%6d [%s] : %7.2f ms/fit  %5.1f TFLOP/s" % (n, optset.strip(), dt * 1e3, n ** 3 / 3 / dt / 1e12), flush=True)
