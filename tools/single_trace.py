"""Run a few single fits (for rocprofv3 --kernel-trace: where does a latency-bound fit spend its time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
X, y, Xs = O.synthetic_problem(n, 8, 20240000, m=1)
with GPR(kernel="rbf") as gp:
    gp.fit(X, y, np.sqrt(8), 1e-2, Xs=Xs)
    t = time.perf_counter()
    for _ in range(5):
        gp.refit(np.sqrt(8), 1e-2)
    print("ms/fit", (time.perf_counter() - t) / 5 * 1e3)
