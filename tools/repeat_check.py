"""Run-to-run bit reproducibility of the engine's default schedules (a race in a hand-off shows up here first): the bench's lockstep step
(G members at 4 grid points, n = 8192), a single fp64 fit at n = 16384 and a single fp32 fit at n = 32768, each repeated and compared
BITWISE with its first run.  usage: repeat_check.py [reps] [group]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from seaiceextentforecasting_amd import GPR

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
G = int(sys.argv[2]) if len(sys.argv) > 2 else 160
bad = 0
n, d, years = 8192, 8, 40
Xb = np.zeros((years, n, d)); yb = np.zeros((years, n)); Xsb = np.zeros((years, 1, d))
for b in range(years):
    Xb[b], yb[b], Xsb[b] = bench.synthetic_problem(n, d, 20240002 + b, m=1)
pts = [bench.grid_point(int(i // years), d, "smoke") for i in range(G, 2 * G)]
ell = np.array([p[0] for p in pts]); sn = np.array([p[1] for p in pts])
with GPR(kernel="rbf", outer_blocks=8) as gp:
    gp.upload_batch(Xb, yb, Xsb, group=G, concurrency=1)
    first = None
    t0 = time.perf_counter()
    for r in range(reps):
        out = gp.run_batch(G, G, ell, sn, concurrency=1, group=G)
        assert np.all(out["info"] == 0)
        cur = tuple(np.array(out[k]).copy() for k in ("mean", "var", "nlml", "sigma_f"))
        if first is None:
            first = cur
        elif not all(np.array_equal(a, b) for a, b in zip(first, cur)):
            bad += 1
            print("lockstep step: run %d differs from run 0 (max |d mean| %.3e)" % (r, float(np.max(np.abs(first[0] - cur[0])))), flush=True)
    print("lockstep step of %d fits: %d runs, %.1f s, %s" % (G, reps, time.perf_counter() - t0, "bit-identical" if bad == 0 else "%d DIFFER" % bad), flush=True)
for dtype, kern, n, d, sn_ in (("f64", "rbf", 16384, 16, 1e-2), ("f32", "matern52", 32768, 32, 1e-1)):
    X, y, Xs = bench.synthetic_problem(n, d, 20240003, m=1)
    with GPR(kernel=kern, dtype=dtype) as gp:
        first = None; b0 = bad
        for r in range(max(3, reps // 2)):
            gp.fit(X, y, float(np.sqrt(d)), sn_, Xs=Xs) if r == 0 else gp.refit(float(np.sqrt(d)), sn_)
            mu, var = gp.predict(Xs)
            cur = (np.array(mu), np.array(var), np.float64(gp.nlml_), np.float64(gp.sigma_f_))
            if first is None:
                first = cur
            elif not all(np.array_equal(a, b) for a, b in zip(first, cur)):
                bad += 1
                print("%s n = %d: run %d differs from run 0" % (dtype, n, r), flush=True)
        print("%s single fit n = %d: %s" % (dtype, n, "bit-identical" if bad == b0 else "DIFFERS"), flush=True)
sys.exit(1 if bad else 0)
