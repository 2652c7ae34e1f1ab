#!/usr/bin/env python3
"""bench.py -- GP fits/sec (n x n fp64, kernel build + Cholesky + predict) on N MI355X.

A step = one pass of the hot path over one batch of the BASELINE.json configs[2] workload: the 40 retrospective
"years" (n=8192, d=8, fp64 RBF) at ONE hyper-parameter grid point, i.e. 40 GP fits factorised in lockstep:
kernel-matrix build -> blocked Cholesky (with y and the test point riding along) -> sigma_f, nlML, predictive
mean/variance at m=1 for each.  The metric stays GP fits/sec (= 40 x steps / time).  Inputs (the years' X, y, Xs)
are resident in HBM before the timed region.  N>1: ranks hold different years (independent fits, no data-path
collective) -> weak scaling; value = total fits / max-over-ranks time.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 8192] [--d 8] [--concurrency C]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # one hardware queue per HIP stream (panel / update streams overlap)
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F64_MFMA_TFLOPS = 78.6   # MI355X dense fp64 matrix peak (= fp64 vector peak), SURVEY.md 8(d)


def synthetic_problem(n, d, seed, m=1):
    """SURVEY 8(d) synthetic inputs: X ~ N(0,1), y = sin(X w) + 0.1 eps, Xs ~ N(0,1)."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    w = rng.standard_normal(d) / np.sqrt(d)
    y = np.sin(X @ w) + 0.1 * rng.standard_normal(n)
    Xs = rng.standard_normal((m, d))
    return X, y, Xs


def grid_point(i, d):
    """Step i's hyper-parameters: the 4x4 'smoke' grid of SURVEY 8(d) around l = sqrt(d), sn~ = 1e-2."""
    ells = np.sqrt(d) * np.logspace(-0.5, 0.5, 4)
    sns = np.logspace(-3, -1, 4)
    return ells[i % 4], sns[(i // 4) % 4]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed steps; a step = one lockstep batch of --group fits (all years at one grid point)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--years", type=int, default=40, help="distinct synthetic data sets (retrospective years) resident per rank")
    ap.add_argument("--concurrency", type=int, default=1, help="lockstep groups in flight")
    ap.add_argument("--group", type=int, default=40, help="fits factorised in lockstep per launch")
    ap.add_argument("--outer", type=int, default=8, help="outer panel width in 128-column blocks (K of the trailing update = 128*outer)")
    ap.add_argument("--reserve-cus", type=int, default=None)
    ap.add_argument("--host-timing", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (repeatable)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event brackets")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    dist = None
    backend = os.environ.get("SIGP_BENCH_BACKEND", "nccl")      # "gloo" lets two ranks rehearse on one GPU
    local = local % torch.cuda.device_count()
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from seaiceextentforecasting_amd import GPR

    n, d, m = args.n, args.d, 1
    years = max(1, args.years)
    Xb = np.zeros((years, n, d)); yb = np.zeros((years, n)); Xsb = np.zeros((years, m, d))
    for b in range(years):
        Xb[b], yb[b], Xsb[b] = synthetic_problem(n, d, 20240002 + 1000 * rank + b, m=m)

    gp = GPR(kernel="rbf", device=local, outer_blocks=args.outer, reserve_cus=args.reserve_cus)
    for o in args.opt:
        k, v = o.split("=")
        gp.set_option(k, int(v))
    if args.host_timing:
        gp.set_option("host_timing", 1)
    G = max(1, args.group)                   # fits per step
    K, W = args.steps * G, args.warmup * G   # timed / warm-up FITS; step s = fits [s*G, (s+1)*G) = every year at grid point s
    ell = np.array([grid_point(i // G, d)[0] for i in range(W + K)])
    sn = np.array([grid_point(i // G, d)[1] for i in range(W + K)])
    # upload + slot allocation (outside the timed region), then W untimed warm-up steps
    gp.upload_batch(Xb, yb, Xsb, group=args.group, concurrency=args.concurrency)
    if W > 0:
        r = gp.run_batch(0, W, ell[:W], sn[:W], concurrency=args.concurrency, group=args.group)
        assert np.all(r["info"] == 0)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if not args.no_profile:
        gp.profile(True, classes=["syrk128"])      # the dominant kernel only: brackets inside the timed region
    gp.profile_reset()
    barrier()
    t0 = time.perf_counter()
    r = gp.run_batch(W, K, ell[W:], sn[W:], concurrency=args.concurrency, group=args.group)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    prof = gp.profile_get()
    gp.profile(False)
    prof_all = None
    if not args.no_profile and rank == 0:
        # per-kernel breakdown from one extra, untimed group with every launch bracketed
        gp.profile(True); gp.profile_reset()
        gp.run_batch(W, min(K, args.group), ell[W:W + min(K, args.group)], sn[W:W + min(K, args.group)], concurrency=1, group=args.group)
        prof_all = gp.profile_get(); gp.profile(False)
        prof_all_fits = min(K, args.group)
    m64 = None
    if rank == 0 and world == 1 and not args.no_profile:
        # "m = 1 (also report m = 64)" -- the same steps with 64 test points riding along each fit (untimed extra)
        rng = np.random.default_rng(7)
        Xs64 = rng.standard_normal((years, 64, d))
        gp.upload_batch(Xb, yb, Xs64, group=args.group, concurrency=args.concurrency)
        k64 = min(K, args.group)
        gp.run_batch(W, k64, ell[W:W + k64], sn[W:W + k64], concurrency=args.concurrency, group=args.group)
        torch.cuda.synchronize(); ta = time.perf_counter()
        r64 = gp.run_batch(W, k64, ell[W:W + k64], sn[W:W + k64], concurrency=args.concurrency, group=args.group)
        torch.cuda.synchronize(); tb64 = time.perf_counter() - ta
        assert np.all(r64["info"] == 0) and np.all(np.isfinite(r64["var"]))
        m64 = {"value": k64 / tb64, "unit": "fits/s", "steps": k64, "note": "same workload with m=64 test points per fit (ride-along rows), measured after the timed region"}
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert np.all(r["info"] == 0) and np.all(np.isfinite(r["mean"]))

    fits = K * world
    value = fits / elapsed
    flops_fit = n ** 3 / 3 + n ** 2 / 2 + n / 6 + n * n * d + n * n / 2 + 2 * n * n + m * (2 * n * d + n * n + 4 * n)
    out = {
        "metric": "GP fits/sec (n x n fp64, kernel+Cholesky+predict)", "value": value, "unit": "fits/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(1, args.steps), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[2]: n=%d d=%d fp64 RBF GPR, batch of retrospective years x hyper-parameter grid, "
                               "one step = the %d years at one grid point factorised in lockstep, each fit = kernel build + blocked Cholesky + sigma_f/nlML + predict m=1" % (n, d, G),
                   "fits_per_step": G, "ms_per_fit": 1e3 * elapsed / K, "years_resident": years, "lockstep_group": args.group, "groups_in_flight": args.concurrency, "parallelism": "years sharded over %d GPU(s), no collective" % world},
        "whole_fit_tflops": value * flops_fit / 1e12 / world,
        "whole_fit_frac_of_fp64_mfma_peak": value * flops_fit / 1e12 / world / PEAK_F64_MFMA_TFLOPS,
        # K~ = k(X,X) + sn I with k <= 1: eigenvalues in [sn, n + sn] -> cond(K~) <= (n + sn)/sn over the grid used
        "cond_upper_bound": {"min": float((n + sn[W:].max()) / sn[W:].max()), "max": float((n + sn[W:].min()) / sn[W:].min())},
    }
    if m64 is not None:
        out["m64"] = m64
    if rank == 0:
        dom = prof["syrk128"]
        if dom["launches"] and dom["ms"] > 0:
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "sigp::syrk128_kernel<double,false> (inner + trailing updates C -= P P^T, fp64 v_mfma_f64_16x16x4_f64)",
                               "achieved": ach, "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F64_MFMA_TFLOPS,
                               "traffic": None, "launches": dom["launches"], "avg_launch_ms": dom["ms"] / dom["launches"],
                               "flops_per_launch": dom["flops"] / dom["launches"]}
            # HBM-side traffic of the same kernel comes from separate rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE cannot
            # share a pass, and PMC collection serialises kernels), summarised under profiles/ by tools/collect_profiles.sh
            try:
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_syrk128.json")))
                if n == 8192 and d == 8 and args.group == 40 and args.outer == 8:
                    out["roofline"]["traffic"] = pm["traffic_bytes_per_launch"]
                    out["roofline"]["traffic_unit"] = "bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, profiles/r01_pmc_syrk128.json)"
                    out["roofline"]["algorithmic_c_bytes_per_launch"] = dom["bytes"] / dom["launches"]
            except Exception:
                pass
        else:
            out["roofline"] = None
        kb = prof["kbuild"]
        src, nf = (prof_all, prof_all_fits) if prof_all is not None else (prof, K)
        out["kernels"] = {k: {"ms_per_fit": v["ms"] / nf, "launches_per_fit": v["launches"] / nf,
                              "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 else None,
                              "GBps": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else None} for k, v in src.items()}
        out["kernels_note"] = "per-kernel table from one extra untimed lockstep group with every launch bracketed; roofline from the timed region"
        del kb
    gp.close()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # reference-idiom CPU path (oracle, call-for-call north/June1st.py:264-277) on the host cores, one fit
        ncpu = os.cpu_count()
        cpu_model = "unknown"
        try:
            with open("/proc/cpuinfo") as f:
                cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), "unknown")
        except OSError:
            pass
        import scipy
        try:
            from threadpoolctl import threadpool_info
            ncpu = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [ncpu])
        except Exception:
            pass
        from oracle import gp_oracle as O      # the CPU checker / baseline: imported for this leg only, after the timed region
        # the sample is the FIRST TIMED STEP of the run above: data set W % years, hyper-parameters of step W
        nb = n if n <= 8192 else 8192
        ds0 = W % years
        if nb == n:
            Xc, yc, Xsc = Xb[ds0], yb[ds0], Xsb[ds0]
        else:
            Xc, yc, Xsc = synthetic_problem(nb, d, 20240002, m=1)
        ell0, sn0 = float(ell[W]), float(sn[W])
        t0 = time.perf_counter()
        ref = O.fit_predict(Xc, yc, Xsc, ell0, sn0, kind="rbf", ref_idiom=True)
        tc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / tc, "unit": "fits/s", "cores": ncpu, "kind": "port",
                               "sample": "1 fit (the first timed step: year %d, l=%.4g, sn~=%.3g), n=%d d=%d, oracle ref_idiom=True = the reference's call sequence north/June1st.py:264-277 (NumPy %s / SciPy %s / OpenBLAS, %d BLAS threads, os.cpu_count()=%d, CPU: %s)" % (ds0, ell0, sn0, nb, d, np.__version__, scipy.__version__, ncpu, os.cpu_count(), cpu_model),
                               "seconds": tc}
        t0 = time.perf_counter()
        O.fit_predict(Xc, yc, Xsc, ell0, sn0, kind="rbf", ref_idiom=False)
        tb = time.perf_counter() - t0
        out["cpu_baseline_best_practice"] = {"value": 1.0 / tb, "unit": "fits/s", "cores": ncpu, "kind": "port", "seconds": tb,
                                             "sample": "1 fit, same inputs, oracle ref_idiom=False (one Cholesky + scipy solve_triangular)"}
        if nb == n:   # parity of the timed configuration against the CPU path on the same inputs: the timed lockstep batch's own
            # result for that step, and the same fit through the single-fit entry point
            with GPR(kernel="rbf", device=local) as g2:
                g2.fit(Xc, yc, ell0, sn0, Xs=Xsc)
                mu, var = g2.predict(Xsc)
            relf = lambda a, b: float(abs(a - b) / abs(b))
            out["parity"] = {"batch_step0_mean_rel": relf(r["mean"][0, 0], ref["fmean"][0]), "batch_step0_var_rel": relf(r["var"][0, 0], ref["fvar"][0]),
                             "batch_step0_nlml_rel": relf(r["nlml"][0], ref["nlml"]),
                             "single_fit_mean_rel": relf(mu[0], ref["fmean"][0]), "single_fit_var_rel": relf(var[0], ref["fvar"][0]), "tolerance": 1e-08}
            assert max(out["parity"]["batch_step0_mean_rel"], out["parity"]["batch_step0_var_rel"]) <= 1e-8, out["parity"]
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
