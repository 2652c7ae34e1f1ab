"""sigp_dist_fit at world = 1 (and the single-GPU entry point beside it) for rocprofv3 --kernel-trace --stats:
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_shard -- python3 tools/shard_profile.py [--n 16384 --d 16 --dtype f64 --kernel rbf --single]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16384); ap.add_argument("--d", type=int, default=16); ap.add_argument("--dtype", default="f64")
ap.add_argument("--kernel", default="rbf"); ap.add_argument("--sn", type=float, default=1e-2); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--single", action="store_true"); ap.add_argument("--rccl", action="store_true"); ap.add_argument("--outer", type=int, default=None, help="outer panel width in 128-column blocks (default: the engine's choice)")
ap.add_argument("--opt", action="append", default=[], help="name=value engine option (single-GPU entry), repeatable")
a = ap.parse_args()
from seaiceextentforecasting_amd import GPR, DistributedGPR
rng = np.random.default_rng(20240003)
X = rng.standard_normal((a.n, a.d)); w = rng.standard_normal(a.d) / np.sqrt(a.d); y = np.sin(X @ w) + 0.1 * rng.standard_normal(a.n); Xs = rng.standard_normal((1, a.d))
ell = np.sqrt(a.d)
if a.single:
    with GPR(kernel=a.kernel, dtype=a.dtype, outer_blocks=a.outer) as g:
        for o in a.opt:
            g.set_option(o.split("=")[0], int(o.split("=")[1]))
        g.fit(X, y, ell, a.sn, Xs=Xs)
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter(); g.refit(ell, a.sn); ts.append(time.perf_counter() - t0)
        extra = " refine_residual %.2e nlml %.10e" % (g.refine_residual_, g.nlml_) if a.dtype == "f32" else ""
    print("single-GPU entry %s: best %.3f ms, all %s%s" % (a.opt, 1e3 * min(ts), [round(1e3 * t, 2) for t in ts], extra))
else:
    with DistributedGPR(a.kernel, 0, 1, None, outer_blocks=a.outer or 8, dtype=a.dtype, stats=True, force_rccl=a.rccl) as g:
        g.fit(X, y, ell, a.sn, Xs=Xs)
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter(); g.refit(ell, a.sn); ts.append(time.perf_counter() - t0)
        print("sigp_dist_fit world=1 (%s): best %.3f ms, all %s, stats %s" % (g.transport, 1e3 * min(ts), [round(1e3 * t, 2) for t in ts], g.stats()))
