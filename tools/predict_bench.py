"""General prediction path (sigp_predict: k*, forward solve v = L \\ k*, mean / variance, north/June1st.py:272-277) after one fit:
time against the number of test points."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
X, y, _ = O.synthetic_problem(n, 8, 20240002, m=1)
rng = np.random.default_rng(1)
with GPR(kernel="rbf") as gp:
    gp.fit(X, y, np.sqrt(8.0), 1e-2)
    for m in (1, 128, 512, 2048, 8192):
        Xs = rng.standard_normal((m, 8))
        gp.predict(Xs)
        t = time.perf_counter()
        for _ in range(2):
            mu, var = gp.predict(Xs)
        dt = (time.perf_counter() - t) / 2
        fl = 2.0 * n * n / 2 * m     # one forward solve with m right-hand sides
        print("n=%d m=%5d : %8.2f ms   %6.2f TFLOP/s (forward solve flops)" % (n, m, dt * 1e3, fl / dt / 1e12), flush=True)
