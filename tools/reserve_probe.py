"""Do the chain kernels of a large single fit run at their stand-alone speed when the update stream is masked off some CUs?
Debug library (reserve_cus is a measurement switch).  Prints ms per fit and the per-class kernel times (HIP-event brackets)."""
import os, sys, time
os.environ["SIGP_USE_DEBUG_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR
for n, d in ((8192, 8), (16384, 16)):
    X, y, Xs = O.synthetic_problem(n, d, 11, m=1)
    for res in (0, 4, 16, 0):
        with GPR(kernel="rbf") as gp:
            gp.set_option("reserve_cus", res)
            gp.fit(X, y, np.sqrt(d), 1e-2, Xs=Xs)
            t = time.perf_counter()
            for _ in range(4):
                gp.refit(np.sqrt(d), 1e-2)
            dt = (time.perf_counter() - t) / 4
            gp.profile(True); gp.profile_reset()
            for _ in range(2):
                gp.refit(np.sqrt(d), 1e-2)
            prof = gp.profile_get()
        print("n=%d reserve_cus=%2d: %.3f ms/fit | " % (n, res, dt * 1e3) + "  ".join("%s %.2f ms/%d" % (k, v["ms"] / 2, v["launches"] // 2) for k, v in prof.items() if v["launches"]), flush=True)
