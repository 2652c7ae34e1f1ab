"""fp32 engine at mid / large orders: the one-pass residual (refine_sym = 1) against the two-pass form on random orders (ragged tiles, bands and chunks of
kres_sym_kernel), 0..3 ride points, both kernels: refined solutions and predictions must agree to 1e-11, residuals stay below 1e-10."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seaiceextentforecasting_amd import GPR
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(np.asarray(b))), 1e-300))
worst = 0.0; fails = 0; t0 = time.time()
edge = [2047, 2048, 2049, 4095, 4097, 8191, 8193, 16383, 16385, 24575, 24577]
for case in range(cases):
    n = int(rng.choice(edge)) if rng.random() < 0.4 else int(rng.integers(2000, 26000))
    d = int(rng.integers(2, 33)); m = int(rng.integers(0, 4)); kind = str(rng.choice(["rbf", "matern52"]))
    X = rng.standard_normal((n, d)); w = rng.standard_normal(d) / np.sqrt(d); y = np.sin(X @ w) + 0.1 * rng.standard_normal(n)
    Xs = rng.standard_normal((m, d)) if m else None
    got = []
    for sym in (1, 0):
        with GPR(kernel=kind, dtype="f32") as gp:
            gp.set_option("refine_sym", sym)
            gp.fit(X, y, float(np.sqrt(d)), 0.1, Xs=Xs)
            mu, var = gp.predict(Xs) if m else (np.zeros(1), np.ones(1))
            got.append((gp.alpha_.copy(), gp.nlml_, mu, var, gp.refine_residual_))
    e = max(rel(got[0][0], got[1][0]), rel(got[0][1], got[1][1]), rel(got[0][2], got[1][2]), rel(got[0][3], got[1][3]))
    worst = max(worst, e)
    ok = e <= 1e-11 and 0 <= got[0][4] <= 1e-10 and 0 <= got[1][4] <= 1e-10
    if not ok:
        fails += 1
        print("FAIL case %d n=%d d=%d m=%d %s: rel %.2e residuals %.1e %.1e" % (case, n, d, m, kind, e, got[0][4], got[1][4]), flush=True)
    if case % 5 == 4:
        print("  %d cases, %.0f s, worst %.1e" % (case + 1, time.time() - t0, worst), flush=True)
print("%d cases in %.0f s, %d failures, worst relative difference %.1e" % (cases, time.time() - t0, fails, worst))
sys.exit(1 if fails else 0)
