"""Per-kernel-class time of one MLII evaluation (nlML + exact gradient), HIP-event brackets on every launch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR
for n in (4096, 8192):
    X, y, _ = O.synthetic_problem(n, 8, 20240001, m=1)
    th = np.log([np.sqrt(8.0), 1e-2])
    with GPR(kernel="rbf") as gp:
        gp.set_data(X, y)
        gp.nlml(th, grad="exact")
        t0 = time.perf_counter()
        for _ in range(3):
            gp.nlml(th, grad="exact")
        wall = (time.perf_counter() - t0) / 3
        t0 = time.perf_counter()
        for _ in range(3):
            gp.refit(float(np.exp(th[0])), float(np.exp(th[1])))
        fit = (time.perf_counter() - t0) / 3
        gp.profile(True); gp.profile_reset()
        gp.nlml(th, grad="exact")
        pr = gp.profile_get(); gp.profile(False)
    print("n=%d: MLII %.2f ms (fit alone %.2f ms); bracketed classes: %s" % (n, 1e3 * wall, 1e3 * fit, {k: (round(v["ms"], 2), v["launches"]) for k, v in pr.items() if v["launches"]}), flush=True)
