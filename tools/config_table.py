"""Throughput of the BASELINE.json configurations that fit one GPU (for the table in DESIGN.md)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR

def flops(n, d, m=1):
    return n ** 3 / 3 + n ** 2 / 2 + n / 6 + n * n * d + n * n / 2 + 2 * n * n + m * (2 * n * d + n * n + 4 * n)

def single(n, d, kind, dtype="f64", reps=5, sn=1e-2):
    X, y, Xs = O.synthetic_problem(n, d, 20240000, m=1)
    with GPR(kernel=kind, dtype=dtype) as gp:
        gp.fit(X, y, np.sqrt(d), sn, Xs=Xs)
        t = time.perf_counter()
        for _ in range(reps):
            gp.refit(np.sqrt(d), sn)
        dt = (time.perf_counter() - t) / reps
    print("single fit  n=%6d d=%2d %-9s %s : %8.2f ms/fit  %7.1f fits/s  %5.1f TFLOP/s" % (n, d, kind, dtype, dt * 1e3, 1 / dt, flops(n, d) / dt / 1e12), flush=True)

def batch(n, d, kind, group, fits, years=4):
    Xb = np.zeros((years, n, d)); yb = np.zeros((years, n)); Xsb = np.zeros((years, 1, d))
    for b in range(years):
        Xb[b], yb[b], Xsb[b] = O.synthetic_problem(n, d, 20240002 + b, m=1)
    ell = np.full(fits, np.sqrt(d)); sn = np.full(fits, 1e-2)
    with GPR(kernel=kind) as gp:
        gp.fit_batch(Xb, yb, Xsb, ell[:group], sn[:group], concurrency=1, group=group)
        t = time.perf_counter()
        r = gp.run_batch(0, fits, ell, sn, concurrency=1, group=group)
        dt = (time.perf_counter() - t) / fits
        assert np.all(r["info"] == 0)
    print("lockstep    n=%6d d=%2d %-9s f64 G=%2d: %8.2f ms/fit  %7.1f fits/s  %5.1f TFLOP/s" % (n, d, kind, group, dt * 1e3, 1 / dt, flops(n, d) / dt / 1e12), flush=True)

single(64, 4, "rbf")
single(4096, 8, "rbf")
batch(4096, 8, "rbf", 32, 64)
single(8192, 8, "rbf")
batch(8192, 8, "rbf", 16, 32)
single(16384, 16, "rbf", reps=2)
batch(16384, 16, "rbf", 4, 8, years=2)
single(32768, 32, "matern52", dtype="f32", reps=2, sn=1e-1)
single(16384, 16, "matern52", dtype="f32", reps=2, sn=1e-1)
