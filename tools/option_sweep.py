"""Single-fit latency against one engine option, alternating over its values (best of the refits of three rounds):
    option_sweep.py <n> <f64|f32> <option> <v1> <v2> ... [other=val ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from seaiceextentforecasting_amd import GPR

n, dtype, opt = int(sys.argv[1]), sys.argv[2], sys.argv[3]
vals = [int(a) for a in sys.argv[4:] if "=" not in a]
fixed = [a.split("=") for a in sys.argv[4:] if "=" in a]
d = 32 if dtype == "f32" else (8 if n <= 8192 else 16)
kern, sn = ("matern52", 1e-1) if dtype == "f32" else ("rbf", 1e-2)
X, y, Xs = bench.synthetic_problem(n, d, 20240003, m=1)
best = {v: 1e9 for v in vals}
res = {}
for rnd in range(3):
    for v in vals:
        with GPR(kernel=kern, dtype=dtype) as gp:
            for k, w in fixed:
                gp.set_option(k, int(w))
            gp.set_option(opt, v)
            gp.fit(X, y, float(np.sqrt(d)), sn, Xs=Xs)
            res[v] = gp.nlml_
            for _ in range(10 if n <= 4096 else 4 if n <= 16384 else 2):
                t = time.perf_counter()
                gp.refit(float(np.sqrt(d)), sn)
                best[v] = min(best[v], time.perf_counter() - t)
print("n = %d %s, %s: " % (n, dtype, opt) + "  ".join("%d: %.3f ms" % (v, best[v] * 1e3) for v in vals) + "   (nlML spread %.1e)" % (max(res.values()) - min(res.values())), flush=True)
