"""Single-fit latency with and without the fused chain link (panel_chain bit 2), same box, alternating: chain_ab.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR
sizes = [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192, 16384]
for n in sizes:
    d = 16 if n >= 16384 else 8
    X, y, Xs = O.synthetic_problem(n, d, 20240000, m=1)
    res = {}
    for rnd in range(2):
        for chain in (3, 7):
            with GPR(kernel="rbf") as gp:
                gp.set_option("panel_chain", chain)
                gp.fit(X, y, np.sqrt(d), 1e-2, Xs=Xs)
                reps = 10 if n <= 8192 else 4
                t = time.perf_counter()
                for _ in range(reps):
                    gp.refit(np.sqrt(d), 1e-2)
                dt = (time.perf_counter() - t) / reps
                res.setdefault(chain, []).append((dt * 1e3, gp.nlml_))
    assert res[3][0][1] == res[7][0][1], "the fused link changed the bits"
    print("n=%6d  panel_chain=3: %s ms   panel_chain=7 (fused link): %s ms" % (n, " / ".join("%.3f" % r[0] for r in res[3]), " / ".join("%.3f" % r[0] for r in res[7])), flush=True)
