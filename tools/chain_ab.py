"""Single-fit latency under option sets, same box, alternating; the sets must give the same bits.
chain_ab.py n[,n...] [name=value ...] / [name=value ...] / ...   (default sets: old chain, fused link, fused link + rest behind the first diagonal block)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR
sizes = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2048, 4096, 8192, 16384]
sets = [x.strip() for x in " ".join(sys.argv[2:]).split("/")] if len(sys.argv) > 2 else ["panel_chain=3 rest_after_diag=0", "panel_chain=7 rest_after_diag=0", "panel_chain=7 rest_after_diag=1"]
dtype = os.environ.get("DTYPE", "f64")
for n in sizes:
    d = 32 if n >= 32768 else (16 if n >= 16384 else 8)
    kern, sn = ("rbf", 1e-2) if dtype == "f64" else ("matern52", 1e-1)
    X, y, Xs = O.synthetic_problem(n, d, 20240000, m=1)
    res = {}
    for rnd in range(2):
        for k, optset in enumerate(sets):
            with GPR(kernel=kern, dtype=dtype) as gp:
                for kv in optset.split():
                    a, v = kv.split("="); gp.set_option(a, int(v))
                gp.fit(X, y, np.sqrt(d), sn, Xs=Xs)
                reps = 10 if n <= 8192 else 4
                t = time.perf_counter()
                for _ in range(reps):
                    gp.refit(np.sqrt(d), sn)
                dt = (time.perf_counter() - t) / reps
                res.setdefault(k, []).append((dt * 1e3, gp.nlml_))
    same = all(res[k][0][1] == res[0][0][1] for k in res)
    print("n=%6d %s %s" % (n, dtype, "" if same else "BITS DIFFER"), flush=True)
    for k, optset in enumerate(sets):
        print("    [%s]: %s ms" % (optset, " / ".join("%.3f" % r[0] for r in res[k])), flush=True)
