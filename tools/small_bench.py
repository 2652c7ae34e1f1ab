"""Reference-size batch (bench.py's reference_kernel_grid workload) with one wavefront vs four wavefronts per fit."""
import os
os.environ["SIGP_USE_DEBUG_LIB"] = "1"      # the switches below are measurement switches of libsigp_debug.so (make debug)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from seaiceextentforecasting_amd import GPR, SmallBatch, LGRID, SGRID
rng = np.random.default_rng(20240010)
sets = []
for N in (60, 20, 12):
    for t in range(40):
        n = 6 + t
        X = rng.standard_normal((n, N)); y = rng.standard_normal(n)
        sets.append((X, y, rng.standard_normal((1, N))))
for nt64 in (1, 0, 1, 0):
    with GPR(kernel="netdiffusion") as gp:
        gp.set_option("small_nt64", nt64)
        sb = SmallBatch(gp)
        for X, y, Xs in sets:
            ds = sb.add_dataset(X, y, Xs)
            for e in LGRID:
                for s_ in SGRID:
                    sb.add_fit(ds, e, s_, expm="eigh")
        sb.upload(); r = sb.run()
        gp.profile(True, classes=["small"]); gp.profile_reset()
        t0 = time.perf_counter()
        for _ in range(5):
            r = sb.run()
        dt = (time.perf_counter() - t0) / 5
        pr = gp.profile_get()["small"]
        print("waves per fit %d: %.3f ms per call (kernel alone %.3f ms), %.2f M fits/s, nlml checksum %.10e" % (1 if nt64 else 4, 1e3 * dt, pr["ms"] / 5, len(r["nlml"]) / dt / 1e6, np.sum(r["nlml"][np.isfinite(r["nlml"])])), flush=True)
