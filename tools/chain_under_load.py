"""Do the chain kernels of a large single fit (diagonal blocks, panel solves, small updates) run faster beside a trailing update that does not
WRITE its C tiles?  Debug library (SIGP_USE_DEBUG_LIB=1), option update_dbg = 16 (no C store; the factorisation's numbers are then garbage and the
fit ends in LinAlgError -- only the per-class kernel times are read).  fp32 n = 32768 (configs[4] shape) and fp64 n = 16384."""
import os, sys
os.environ["SIGP_USE_DEBUG_LIB"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR
for n, d, dtype, kern, sn in ((32768, 32, "f32", "matern52", 1e-1), (16384, 16, "f64", "rbf", 1e-2)):
    X, y, Xs = O.synthetic_problem(n, d, 11, m=1)
    with GPR(kernel=kern, dtype=dtype) as gp:
        gp.fit(X, y, np.sqrt(d), sn, Xs=Xs)
        for dbg in (0, 16, 8, 24, 0):
            gp.set_option("update_dbg", dbg)
            gp.profile(True); gp.profile_reset()
            for _ in range(2):
                try:
                    gp.refit(np.sqrt(d), sn)
                except np.linalg.LinAlgError:
                    pass
            prof = gp.profile_get()
            print("%s n=%d update_dbg=%2d: " % (dtype, n, dbg) + "  ".join("%s %.2f ms" % (k, v["ms"] / 2) for k, v in prof.items() if v["launches"]), flush=True)
