"""Covariance build, VALU distances (kbuild_kernel) against GEMM-form distances on the matrix pipe (kbuild_mfma_kernel): kernel time by HIP events
(profile class kbuild) for the headline batch (n = 8192, d = 8, G = 40), a d = 32 single fit and the fp32 configs[4] build, plus max |K~ difference|."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import GPR

VAR = tuple(int(v) for v in os.environ.get("KBUILD_AB_VARIANTS", "0,1").split(","))     # values of option kbuild_mfma to compare

def batch(n, d, G, kern):
    Xb = np.zeros((G, n, d)); yb = np.zeros((G, n)); Xsb = np.zeros((G, 1, d))
    for b in range(G):
        Xb[b], yb[b], Xsb[b] = O.synthetic_problem(n, d, 20240002 + b, m=1)
    ell = np.full(G, np.sqrt(d)); sn = np.full(G, 1e-2)
    for mf in VAR + VAR:
        with GPR(kernel=kern) as gp:
            gp.set_option("kbuild_mfma", mf)
            gp.upload_batch(Xb, yb, Xsb, group=G, concurrency=1)
            gp.run_batch(0, G, ell, sn, concurrency=1, group=G)
            gp.profile(True, classes=["kbuild"]); gp.profile_reset()
            t = time.perf_counter(); r = gp.run_batch(0, G, ell, sn, concurrency=1, group=G); dt = time.perf_counter() - t
            p = gp.profile_get()["kbuild"]
        byt = G * (4.0 * n * (n + 1) + 8.0 * n * d)
        print("batch n=%d d=%d G=%d %s kbuild_mfma=%d: kbuild %.3f ms = %.2f TB/s (lower-triangle bytes), step %.2f ms, nlml[0] %.12f" % (n, d, G, kern, mf, p["ms"], byt / p["ms"] / 1e9, dt * 1e3, r["nlml"][0]), flush=True)

def single(n, d, kern, dtype, sn):
    X, y, Xs = O.synthetic_problem(n, d, 20240004, m=1)
    K = {}
    for mf in VAR + VAR:
        with GPR(kernel=kern, dtype=dtype) as gp:
            gp.set_option("kbuild_mfma", mf)
            gp.fit(X, y, np.sqrt(d), sn, Xs=Xs)
            gp.profile(True, classes=["kbuild"]); gp.profile_reset()
            t = time.perf_counter(); gp.refit(np.sqrt(d), sn); dt = time.perf_counter() - t
            p = gp.profile_get()["kbuild"]
            mu, var = gp.predict(Xs)
            K[mf] = (mu[0], var[0], gp.nlml_)
        print("single n=%d d=%d %s %s kbuild_mfma=%d: kbuild %.3f ms, fit %.2f ms, mean %.15g var %.15g nlml %.15g" % (n, d, kern, dtype, mf, p["ms"], dt * 1e3, *K[mf]), flush=True)
    print("   relative difference %d vs %d: mean %.2e var %.2e nlml %.2e" % ((VAR[1], VAR[0]) + tuple(abs(a - b) / abs(b) for a, b in zip(K[VAR[1]], K[VAR[0]]))), flush=True)

batch(8192, 8, 40, "rbf")
single(8192, 32, "matern52", "f64", 1e-2)
single(32768, 32, "matern52", "f32", 1e-1)
