"""Randomised sharded fits (sigp_dist_fit) against the oracle, every rank checking its own copy of the results:
    python -m torch.distributed.run --nproc-per-node W --master-addr 127.0.0.1 tools/fuzz_sharded.py [cases] [seed]
Ranks share the box's GPU (host-pointer transport over gloo).  Random order n (block-boundary cases included), feature count, ride
rows, kernel, precision, panel width (1..9 blocks, also wider than the matrix), look-ahead on/off, panel exchange whole / by row pieces."""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import gp_oracle as O
from seaiceextentforecasting_amd import DistributedGPR

rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)          # the same stream on every rank
edge = [1, 2, 127, 128, 129, 255, 256, 257, 383, 385, 1023, 1024, 1025, 1151, 1153, 2047, 2049]
worst = {}
t0 = time.time()
fails = 0
for case in range(cases):
    n = int(rng.choice(edge)) if rng.random() < 0.5 else int(rng.integers(1, 2400))
    d = int(rng.integers(1, 33)); m = int(rng.integers(0, 4))
    kind = str(rng.choice(["rbf", "matern52", "netdiffusion"]))
    dtype = "f32" if (kind != "netdiffusion" and rng.random() < 0.4) else "f64"
    W = int(rng.integers(1, 10)); la = bool(rng.integers(0, 2)); split = bool(rng.integers(0, 2))      # panel exchange: whole-panel broadcast / row pieces + all-gather
    X, y, Xs = O.synthetic_problem(n, d, 7000 + case + 1000 * seed, m=max(m, 1))
    Xs = Xs[:m] if m else None
    if kind == "netdiffusion":
        ell, sn = float(10 ** rng.uniform(-3, -1)), float(10 ** rng.uniform(-2, 0))
    else:
        ell, sn = float(np.sqrt(d) * 10 ** rng.uniform(-0.4, 0.4)), float(10 ** rng.uniform(-2 if dtype == "f64" else -1, 0))
    ref = O.fit_predict(X, y, Xs if m else X[:1], ell, sn, kind=kind, ref_idiom=False)
    with DistributedGPR(kind, rank, world, dist if world > 1 else None, outer_blocks=W, lookahead=la, dtype=dtype, panel_split=split) as dg:
        dg.fit(X, y, ell, sn, Xs=Xs)
        got = {"nlml": dg.nlml_, "sigma_f": dg.sigma_f_}
        if m:
            got["mean"], got["var"] = dg.predict(Xs)
        if kind != "netdiffusion" and case % 2 == 0:        # new points through sigp_dist_predict
            Xn = np.random.default_rng(case).standard_normal((int(rng.integers(1, 10)), d))
            refn = O.fit_predict(X, y, Xn, ell, sn, kind=kind, ref_idiom=False)
            mn, vn = dg.predict(Xn)
            em = float(np.max(np.abs(mn - refn["fmean"])) / max(np.max(np.abs(refn["fmean"])), 1e-300)); ev = float(np.max(np.abs(vn - refn["fvar"]) / np.abs(refn["fvar"])))
            worst[(dtype, "new_mean")] = max(worst.get((dtype, "new_mean"), 0.0), em); worst[(dtype, "new_var")] = max(worst.get((dtype, "new_var"), 0.0), ev)
            if not (em <= (1e-8 if dtype == "f64" else 1e-6) and ev <= (1e-8 if dtype == "f64" else 2e-3)):
                fails += 1
                print("rank %d FAIL case %d n=%d d=%d %s %s W=%d: dist_predict mean %.2e var %.2e" % (rank, case, n, d, kind, dtype, W, em, ev), flush=True)
        if dtype == "f32":
            assert (0 < dg.refine_residual_ or n <= 4) and dg.refine_residual_ <= 1e-9, (case, n, dg.refine_residual_)   # (orders 1-4 can be solved exactly)
    tol = {"mean": 1e-8, "var": 1e-8, "nlml": 1e-9, "sigma_f": 1e-8} if dtype == "f64" else {"mean": 1e-6, "var": 1e-5, "nlml": 5e-5, "sigma_f": 1e-6}
    refd = {"mean": ref["fmean"][:m], "var": ref["fvar"][:m], "nlml": ref["nlml"], "sigma_f": ref["sigma_f"]}
    for k in got:
        a, b = np.atleast_1d(np.asarray(got[k], dtype=float)), np.atleast_1d(np.asarray(refd[k], dtype=float))
        e = float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
        if k == "nlml":                      # nlML passes through zero: bound the absolute error per order as well
            e = min(e, float(np.max(np.abs(a - b))) / (1e-3 * max(n, 1)) * tol[k] / (1e-9 if dtype == "f64" else 5e-5) * 1e-9)
        worst[(dtype, k)] = max(worst.get((dtype, k), 0.0), e)
        if not e <= tol[k]:
            fails += 1
            print("rank %d FAIL case %d n=%d d=%d m=%d %s %s W=%d la=%s: %s rel err %.3e" % (rank, case, n, d, m, kind, dtype, W, la, k, e), flush=True)
print("rank %d/%d: %d cases in %.0f s, %d failures, worst %s" % (rank, world, cases, time.time() - t0, fails, {("%s/%s" % k): "%.1e" % v for k, v in sorted(worst.items())}), flush=True)
if world > 1:
    dist.barrier(); dist.destroy_process_group()
sys.exit(1 if fails else 0)
