#!/usr/bin/env python3
"""bench.py -- GP fits/sec (n x n fp64, kernel build + Cholesky + predict) on N MI355X.

A step = one pass of the hot path over one batch of the BASELINE.json configs[2] workload: the 40 retrospective
"years" (n=8192, d=8, fp64 RBF) at ONE hyper-parameter grid point, i.e. 40 GP fits factorised in lockstep:
kernel-matrix build -> blocked Cholesky (with y and the test point riding along) -> sigma_f, nlML, predictive
mean/variance at m=1 for each.  The metric stays GP fits/sec (= 40 x steps / time).  Inputs (the years' X, y, Xs)
are resident in HBM before the timed region.

N > 1 (one process per GPU, RCCL for the barrier / max-reduce only -- independent fits need no data-path collective; AFTER the
metric the same ranks run BASELINE configs[3] / [4] as ONE fit sharded over all of them -- the library's own RCCL communicator,
block-row panel broadcast -- and rank 0 adds the untimed "sharded" record to the line):
  --scaling weak   (default) every rank holds its own 40 years and runs the same number of steps: per-GPU work fixed
  --scaling strong the fixed job (steps x 40 fits) is dealt round-robin over the ranks (a rank's lockstep groups then mix
                   years and grid points)
`python bench.py --gpus N` without a torchrun environment starts the N ranks itself (a `torch.distributed.run` child,
spawned before this process touches the GPU); under torchrun WORLD_SIZE must equal --gpus.  (torchrun's own parser claims
abbreviations of ITS options even after the script name, e.g. --n / --d / --no: pass such flags through the self-spawning
form, or as a JSON list in SIGP_BENCH_ARGV with no flags on the command line; --gpus / --steps / --warmup are unaffected.)

    python bench.py [--gpus N] [--steps K] [--warmup W] [--grid smoke|full] [--scaling weak|strong] [--n 8192] [--d 8]
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

# one hardware queue per HIP stream (panel / update / communication streams overlap).  Multi-rank runs also hold two RCCL communicators
# (torch's and the library's), each with a dozen streams of its own: more queues, so that none of them shares one with the library's streams
_USER_HW_QUEUES = os.environ.get("GPU_MAX_HW_QUEUES")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32" if int(os.environ.get("WORLD_SIZE", "1")) > 1 else "16")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F64_MFMA_TFLOPS = 78.6    # MI355X dense fp64 matrix peak (= fp64 vector peak), SURVEY.md 8(d)
PEAK_F32_MFMA_TFLOPS = 157.3   # dense fp32 matrix peak


def synthetic_problem(n, d, seed, m=1):
    """SURVEY 8(d) synthetic inputs: X ~ N(0,1), y = sin(X w) + 0.1 eps, Xs ~ N(0,1)."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    w = rng.standard_normal(d) / np.sqrt(d)
    y = np.sin(X @ w) + 0.1 * rng.standard_normal(n)
    Xs = rng.standard_normal((m, d))
    return X, y, Xs


def grid_axes(d, which):
    """SURVEY 8(d) hyper-parameter grid of configs[2]: l in sqrt(d) logspace(-1, 1, G1), sn~ in logspace(-3, 1, G2);
    full = 20 x 20 (mirrors the reference's north/June1st.py:210-211 axes), smoke = 4 x 4 over the same ranges."""
    g = 20 if which == "full" else 4
    return np.sqrt(d) * np.logspace(-1, 1, g), np.logspace(-3, 1, g)


def grid_point(i, d, which="smoke"):
    """Step i's hyper-parameters: the grid is walked l-fastest and wraps."""
    ells, sns = grid_axes(d, which)
    return ells[i % len(ells)], sns[(i // len(ells)) % len(sns)]


def flops_per_fit(n, d, m=1):
    """SURVEY 8(d) F(n, d, m): potrf + lower-triangle distance build + two TRSV + predict."""
    return n ** 3 / 3 + n ** 2 / 2 + n / 6 + n * n * d + n * n / 2 + 2 * n * n + m * (2 * n * d + n * n + 4 * n)


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start N ranks as a child `torch.distributed.run` and pass its exit
    code on.  Nothing in this process has touched the GPU yet (device_count() does not initialise it)."""
    import torch
    backend = os.environ.get("SIGP_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    if backend == "nccl" and ndev < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible; one rank per GPU over RCCL needs %d "
                         "(SIGP_BENCH_BACKEND=gloo lets ranks share a GPU for rehearsal)" % (args.gpus, ndev, args.gpus))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)]
    env = dict(os.environ)
    env["SIGP_BENCH_ARGV"] = json.dumps(sys.argv[1:])     # torchrun's own argparse would claim abbreviations such as --n / --d
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # the host driver only supports dmabuf IPC (RCCL needs it)
    if _USER_HW_QUEUES is None:
        env["GPU_MAX_HW_QUEUES"] = "32"                   # (see the top of this file)
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed steps; a step = one lockstep batch of --group fits (all years at group / years consecutive grid points)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--years", type=int, default=40, help="distinct synthetic data sets (retrospective years) of the job")
    ap.add_argument("--grid", choices=["smoke", "full"], default="smoke", help="hyper-parameter grid the steps walk: 4x4 or 20x20 over the SURVEY 8(d) ranges")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--concurrency", type=int, default=1, help="lockstep groups in flight")
    ap.add_argument("--group", type=int, default=160, help="fits factorised in lockstep per launch: 160 = the 40 years at 4 grid points (86 GB of matrices; 40: 325, 80: 333, 120: 335, 160: 336.5, 240: 337 fits/s on one box)")
    ap.add_argument("--outer", type=int, default=8, help="outer panel width in 128-column blocks (K of the trailing update = 128*outer)")
    ap.add_argument("--host-timing", action="store_true")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (repeatable)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=3, help="timed CPU repetitions at n=8192 after one warm-up (BASELINE.md section 2 protocol: 1 + 3, median; ~48 s each in the reference idiom); 1 = one cold fit")
    ap.add_argument("--no-sharded", action="store_true", help="N > 1: skip the untimed sharded configs[3]/[4] record")
    ap.add_argument("--sharded-timeout", type=float, default=240.0, help="N > 1: seconds after which the line is printed without the sharded record")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel HIP-event brackets")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed extra records (other configs, MLII, reference-kernel grid)")
    args = ap.parse_args(json.loads(os.environ["SIGP_BENCH_ARGV"]) if len(sys.argv) == 1 and "SIGP_BENCH_ARGV" in os.environ else None)

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)                                        # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU: python -m torch.distributed.run "
                         "--nproc-per-node %d bench.py --gpus %d ...)" % (args.gpus, world, args.gpus, args.gpus))
    # N = 1 runs on the HIP runtime the library links (/opt/rocm's) and never imports torch: the timed region is bracketed with the
    # library's own sigp_synchronize.  torch is plumbing for N > 1 only (process group: barrier + max-reduce of the timing; it must be
    # imported before the library there, see _lib.load).
    torch = dist = None
    backend = os.environ.get("SIGP_BENCH_BACKEND", "nccl")      # "gloo" lets two ranks rehearse on one GPU
    if world > 1:
        import torch
        import torch.distributed as dist
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
        if backend == "nccl" and world > torch.cuda.device_count():
            raise SystemExit("bench.py: %d ranks but %d GPU(s) visible (RCCL needs one GPU per rank)" % (world, torch.cuda.device_count()))
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)

    from seaiceextentforecasting_amd import GPR

    n, d, m = args.n, args.d, 1
    years = max(1, args.years)
    G = max(1, args.group)                   # fits per step
    G_asked = G
    # the group's matrices are ONE allocation of G (n + 128) n doubles (86 GB at the defaults): on a GPU that does not have that much free
    # (shared box, another tenant) fall back to fewer grid points per launch instead of failing -- the record says so
    try:
        import ctypes
        from seaiceextentforecasting_amd import _lib as _L
        hip = ctypes.CDLL(_L.runtime_info().split(" from ")[-1].strip())      # the runtime already mapped into this process (never a second one)
        free_b, total_b = ctypes.c_size_t(0), ctypes.c_size_t(0)
        ndev = ctypes.c_int(0)
        if hip.hipSetDevice(int(local)) == 0 and hip.hipMemGetInfo(ctypes.byref(free_b), ctypes.byref(total_b)) == 0 and hip.hipGetDeviceCount(ctypes.byref(ndev)) == 0:
            per_member = (n + 128) * n * 8.0 * 1.02
            reserve = 70e9 if not args.no_extras else 8e9          # the untimed extras (fp32 group of 4, MLII group of 40) allocate beside it
            share = -(-world // max(1, ndev.value))                # ranks of a rehearsal that share this GPU (1 in a real launch)
            while G > years and G * per_member + reserve > free_b.value / share:
                G -= years
    except Exception:
        pass
    if dist is not None:                                           # every rank runs the same group (the smallest any of them can hold)
        tg = torch.tensor([G], dtype=torch.int64, device=("cuda" if backend == "nccl" else "cpu"))
        dist.all_reduce(tg, op=dist.ReduceOp.MIN)
        G = int(tg.item())
    args.group = G
    total_steps = args.warmup + args.steps
    # ---- which fits this rank runs.  Global fit i = year i % years at grid point i // years; step s = fits [s G, (s + 1) G). -----------------
    if args.scaling == "weak" or world == 1:
        # every rank: its own `years` data sets (different seeds), all steps
        my_years = list(range(years))
        seeds = [20240002 + 1000 * rank + b for b in my_years]
        my_fits = np.arange(total_steps * G)
        fit_step = my_fits // G
        fit_point = my_fits // years
        n_warm = args.warmup * G
    else:
        # strong: the fixed job of steps x G fits is dealt round-robin; the warm-up steps are dealt the same way
        all_fits = np.arange(total_steps * G)
        mine = all_fits[rank::world]
        fit_step = mine // G
        fit_point = mine // years
        yr = mine % years
        period = len(np.unique(yr))          # the rank's year sequence is periodic (years / gcd(world, years))
        assert np.array_equal(yr, np.tile(yr[:period], len(yr) // period + 1)[:len(yr)])
        my_years = [int(v) for v in yr[:period]]
        seeds = [20240002 + b for b in my_years]
        my_fits = mine
        n_warm = int(np.sum(fit_step < args.warmup))
    Xb = np.zeros((len(my_years), n, d)); yb = np.zeros((len(my_years), n)); Xsb = np.zeros((len(my_years), m, d))
    for j, sd in enumerate(seeds):
        Xb[j], yb[j], Xsb[j] = synthetic_problem(n, d, sd, m=m)
    ell = np.array([grid_point(int(s), d, args.grid)[0] for s in fit_point])
    sn = np.array([grid_point(int(s), d, args.grid)[1] for s in fit_point])
    K, W = len(my_fits) - n_warm, n_warm    # this rank's timed / warm-up FITS

    gp = GPR(kernel="rbf", device=local, outer_blocks=args.outer)
    for o in args.opt:
        k, v = o.split("=")
        gp.set_option(k, int(v))
    if args.host_timing:
        gp.set_option("host_timing", 1)
    # upload + slot allocation (outside the timed region), then the untimed warm-up steps
    gp.upload_batch(Xb, yb, Xsb, group=args.group, concurrency=args.concurrency)
    if W > 0:
        r = gp.run_batch(0, W, ell[:W], sn[:W], concurrency=args.concurrency, group=args.group)
        assert np.all(r["info"] == 0)

    def sync():
        gp.synchronize()                     # hipDeviceSynchronize on the library's runtime ...
        if torch is not None:
            torch.cuda.synchronize()         # ... and, N > 1, on torch's view of the device (its collectives)

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        sync()

    if not args.no_profile:
        gp.profile(True, classes=["syrk128"])      # the dominant kernel only: brackets inside the timed region
    gp.profile_reset()
    barrier()
    t0 = time.perf_counter()
    r = gp.run_batch(W, K, ell[W:], sn[W:], concurrency=args.concurrency, group=args.group)
    sync()
    t1 = time.perf_counter()
    barrier()
    prof = gp.profile_get()
    gp.profile(False)
    unbracketed = None
    if not args.no_profile and not args.no_extras:
        # the same steps once more WITHOUT the per-launch HIP-event brackets of the dominant kernel (`value` carries them)
        barrier(); ta = time.perf_counter()
        gp.run_batch(W, K, ell[W:], sn[W:], concurrency=args.concurrency, group=args.group)
        sync(); tub = time.perf_counter() - ta
        barrier()
        unbracketed = tub
    prof_all = None
    if not args.no_profile and rank == 0:
        # per-kernel breakdown from one extra, untimed group with every launch bracketed
        gp.profile(True); gp.profile_reset()
        kk = min(K, args.group)
        gp.run_batch(W, kk, ell[W:W + kk], sn[W:W + kk], concurrency=1, group=args.group)
        prof_all = gp.profile_get(); gp.profile(False)
        prof_all_fits = kk
    unshared = None
    if not args.no_profile and rank == 0 and world == 1 and not args.no_extras:
        # the same kernel NOT sharing the chip with the panel stream's strip solve: one extra, untimed pair of groups with the strip solve
        # serialised behind the trailing update (option strips_after_update; slightly lower fits/s, which is why it is not the default)
        gp.set_option("strips_after_update", 1)
        kk = min(K, 2 * args.group)
        gp.run_batch(W, kk, ell[W:W + kk], sn[W:W + kk], concurrency=1, group=args.group)
        gp.profile(True, classes=["syrk128"]); gp.profile_reset()
        sync(); ta = time.perf_counter()
        gp.run_batch(W, kk, ell[W:W + kk], sn[W:W + kk], concurrency=1, group=args.group)
        sync(); tu = time.perf_counter() - ta
        pu = gp.profile_get()["syrk128"]; gp.profile(False)
        gp.set_option("strips_after_update", 0)
        if pu["ms"] > 0:
            au = pu["flops"] / (pu["ms"] * 1e-3) / 1e12
            unshared = {"achieved": au, "frac": au / PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "launches": pu["launches"], "avg_launch_ms": pu["ms"] / pu["launches"],
                        "fits_per_s_with_this_schedule": kk / tu,
                        "note": "syrk128_kernel when the panel stream's strip solve (MFMA work for the whole chip) waits for the trailing update instead of running beside it: "
                                "the kernel's own rate; the default schedule overlaps them because the batch is ~1 % faster that way"}
    m64 = None
    if rank == 0 and world == 1 and not args.no_profile and not args.no_extras:
        # "m = 1 (also report m = 64)" -- the same steps with 64 test points riding along each fit (untimed extra)
        rng = np.random.default_rng(7)
        Xs64 = rng.standard_normal((len(my_years), 64, d))
        gp.upload_batch(Xb, yb, Xs64, group=args.group, concurrency=args.concurrency)
        k64 = min(K, args.group)
        gp.run_batch(W, k64, ell[W:W + k64], sn[W:W + k64], concurrency=args.concurrency, group=args.group)
        sync(); ta = time.perf_counter()
        r64 = gp.run_batch(W, k64, ell[W:W + k64], sn[W:W + k64], concurrency=args.concurrency, group=args.group)
        sync(); tb64 = time.perf_counter() - ta
        assert np.all(r64["info"] == 0) and np.all(np.isfinite(r64["var"]))
        m64 = {"value": k64 / tb64, "unit": "fits/s", "steps": k64, "note": "same workload with m=64 test points per fit (ride-along rows), measured after the timed region"}
    elapsed = t1 - t0
    if dist is not None:
        t = torch.tensor([elapsed, unbracketed or 0.0], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        if unbracketed is not None:
            unbracketed = float(t[1].item())
    assert np.all(r["info"] == 0) and np.all(np.isfinite(r["mean"]))

    fits = args.steps * G * (world if args.scaling == "weak" else 1)       # whole-job timed fits
    value = fits / elapsed
    flops_fit = flops_per_fit(n, d, m)
    sn_t = sn[W:]
    out = {
        "metric": "GP fits/sec (n x n fp64, kernel+Cholesky+predict)", "value": value, "unit": "fits/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / max(1, args.steps), "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[2]: n=%d d=%d fp64 RBF GPR, batch of retrospective years x hyper-parameter grid (%s: %dx%d over l = sqrt(d) logspace(-1,1), sn~ = logspace(-3,1)), "
                               "one step = %d fits factorised in lockstep (the %d years at %s consecutive grid points), each fit = kernel build + blocked Cholesky + sigma_f/nlML + predict m=1"
                               % (n, d, args.grid, len(grid_axes(d, args.grid)[0]), len(grid_axes(d, args.grid)[1]), G, years, ("%g" % (G / years))),
                   "n": n, "d": d, "fits_per_step": G, "ms_per_fit": 1e3 * elapsed / max(1, fits // world if args.scaling == "weak" else fits), "years_resident_per_rank": len(my_years),
                   "lockstep_group": args.group, "lockstep_group_asked": G_asked, "groups_in_flight": args.concurrency, "grid": args.grid,
                   "parallelism": ("years sharded over %d GPU(s): every rank its own %d years, no data-path collective" % (world, years)) if args.scaling == "weak" else
                                  ("the fixed %d-fit job dealt round-robin over %d GPU(s), no data-path collective" % (fits, world))},
        "whole_fit_tflops": value * flops_fit / 1e12 / world,
        "whole_fit_frac_of_fp64_mfma_peak": value * flops_fit / 1e12 / world / PEAK_F64_MFMA_TFLOPS,
        # K~ = k(X,X) + sn I with k <= 1: eigenvalues in [sn, n + sn] -> cond(K~) <= (n + sn)/sn over the grid points timed
        "cond_upper_bound": {"min": float((n + sn_t.max()) / sn_t.max()), "max": float((n + sn_t.min()) / sn_t.min())},
        "kernel_function_parity": "RBF is not in the reference (its kernel is X expm(lM) X^T): the fit/solve/predict skeleton is pinned by the reference's goldens, the RBF function itself only by scikit-learn (tests/test_oracle_golden.py)",
    }
    if m64 is not None:
        out["m64"] = m64
    if unbracketed is not None:
        out["without_event_brackets"] = {"value": fits / unbracketed, "unit": "fits/s",
                                         "note": "the same timed steps run once more with no HIP-event brackets around the dominant kernel (untimed extra; `value` is measured WITH them, as the roofline needs)"}
    if rank == 0:
        dom = prof["syrk128"]
        if dom["launches"] and dom["ms"] > 0:
            ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "sigp::syrk128_kernel<double,false> (inner + trailing updates C -= P P^T, fp64 v_mfma_f64_16x16x4_f64)",
                               "achieved": ach, "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F64_MFMA_TFLOPS,
                               "traffic": None, "launches": dom["launches"], "avg_launch_ms": dom["ms"] / dom["launches"],
                               "flops_per_launch": dom["flops"] / dom["launches"],
                               "flops_note": "algorithmic: 2*128*128*K per off-diagonal tile, the lower half (128*129*K) per diagonal tile"}
            if unshared is not None:
                out["roofline"]["unshared"] = unshared
            # HBM-side traffic of the same kernel is NOT measured in this run: it comes from separate rocprofv3 --pmc passes of this
            # command (FETCH_SIZE / WRITE_SIZE cannot share a pass, and PMC collection serialises kernels), summarised under
            # profiles/ by tools/collect_profiles.sh + tools/summarize_pmc.py
            # ... and quoted only when the file was collected from THIS kernel code (sha of the csrc files recorded at collection time)
            for rnd in ("r04", "r03", "r02"):
                pth = os.path.join(ROOT, "profiles", "%s_pmc_syrk128.json" % rnd)
                if not os.path.exists(pth):
                    continue
                try:
                    pm = json.load(open(pth))
                    if n == 8192 and d == 8 and args.group == pm.get("lockstep_group", 40) and args.outer == 8:     # (per-launch bytes scale with the members per launch)
                        if pm.get("kernel_code_sha16") == kernel_code_sha16():
                            out["roofline"]["traffic"] = pm["traffic_bytes_per_launch"]
                            out["roofline"]["traffic_source"] = ("from profiles/%s_pmc_syrk128.json (rocprofv3 --pmc passes of this command on this kernel code, sha16 %s; NOT this run): "
                                                                 "bytes per launch = FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE" % (rnd, pm["kernel_code_sha16"]))
                        else:
                            out["roofline"]["traffic_source"] = ("profiles/%s_pmc_syrk128.json was collected from other kernel code (sha16 %s, now %s): not quoted"
                                                                 % (rnd, pm.get("kernel_code_sha16"), kernel_code_sha16()))
                        out["roofline"]["algorithmic_c_bytes_per_launch"] = dom["bytes"] / dom["launches"]
                except Exception:
                    pass
                break
        else:
            out["roofline"] = None
        src, nf = (prof_all, prof_all_fits) if prof_all is not None else (prof, K)
        out["kernels"] = {k: {"ms_per_fit": v["ms"] / nf, "launches_per_fit": v["launches"] / nf,
                              "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 else None,
                              "GBps": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else None} for k, v in src.items()}
        out["kernels_note"] = "per-kernel table from one extra untimed lockstep group with every launch bracketed; roofline from the timed region"
    gp.close()

    if rank == 0 and world == 1 and not args.no_extras:
        try:
            out["other_configs"] = other_configs(local)
        except Exception as e:                   # untimed extras must never cost the metric line
            out["other_configs"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_reps"] = max(1, args.cpu_reps)
        cpu_baseline(out, args, Xb, yb, Xsb, ell, sn, W, len(my_years), r, local)
        if out.get("cpu_baseline", {}).get("value"):
            out["vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]     # (vs_baseline stays null: the reference publishes no number for this metric)
    if world > 1 and not args.no_sharded:
        # ONE fit sharded over all ranks (configs[3], configs[4]): untimed record.  A watchdog prints the metric line without it if a
        # collective hangs -- nothing after the timed region may cost the line.
        import threading

        def give_up():
            if rank == 0:
                out["sharded"] = {"error": "no result within %.0f s (watchdog)" % args.sharded_timeout}
                emit(out)
            os._exit(0)

        wd = threading.Timer(args.sharded_timeout, give_up)
        wd.daemon = True
        wd.start()
        try:
            rec = sharded_record(rank, world, local, dist, backend)
        except Exception as e:                   # noqa: BLE001
            rec = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        wd.cancel()
        if rank == 0:
            out["sharded"] = rec
    if rank == 0:
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


def compact_line(out):
    """The metric line the driver records (it keeps a 2 000-character tail): the contract's fields, `roofline`, `cpu_baseline`, the parity of
    the timed step and one number per other BASELINE configuration -- at most 1.5 KB.  The full record goes to stderr and to
    gpurun_out/bench_verbose.json."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    c = {k: out.get(k) for k in keep}
    cfg = out.get("config", {})
    c["config"] = {"workload": "configs[2]: n=%s d=%s fp64 RBF GPR, %s retrospective years x hyper-parameter grid, one step = fits_per_step fits in lockstep"
                               % (cfg.get("n"), cfg.get("d"), cfg.get("years_resident_per_rank")),
                   "fits_per_step": cfg.get("fits_per_step"), "grid": cfg.get("grid"), "parallelism": "years sharded over %s GPU(s), no data-path collective" % out.get("n_gpus")}
    rf = out.get("roofline")
    if rf:
        c["roofline"] = {"bound": rf["bound"], "kernel": "syrk128_kernel<double> (trailing update, v_mfma_f64_16x16x4_f64)", "achieved": round(rf["achieved"], 3), "peak": rf["peak"],
                         "unit": rf["unit"], "frac": round(rf["frac"], 4), "traffic": rf.get("traffic"), "avg_launch_ms": round(rf["avg_launch_ms"], 4), "launches": rf["launches"]}
        if rf.get("unshared"):
            c["roofline"]["unshared_frac"] = round(rf["unshared"]["frac"], 4)
    else:
        c["roofline"] = None
    cb = out.get("cpu_baseline")
    if cb:
        c["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "seconds": round(cb["seconds"], 2),
                             "sample": "oracle in the reference's call sequence (north/June1st.py:264-277), n=%s, 1 warm-up + %s timed, median" % (cfg.get("n"), out.get("cpu_reps"))}
        c["vs_cpu_baseline"] = round(out.get("vs_cpu_baseline", 0.0), 1)
    if "parity" in out:
        c["parity"] = {k: float("%.2e" % v) for k, v in out["parity"].items() if k.startswith("batch_step0")}
    c["whole_fit_frac"] = round(out.get("whole_fit_frac_of_fp64_mfma_peak", 0.0), 4)
    oc = out.get("other_configs") or {}
    def ms(prefix, key="ms_per_fit"):
        for k, v in oc.items():
            if k.startswith(prefix) and isinstance(v, dict) and key in v:
                return v
        return None
    for name, prefix in (("c1", "configs[1]"), ("c3", "configs[3]"), ("c4", "configs[4] n=")):
        v = ms(prefix)
        if v:
            c[name + "_ms"] = round(v["ms_per_fit"], 3); c[name + "_frac"] = round(v["frac_of_peak"], 4)
    v = ms("configs[4] shape, lockstep")
    if v:
        c["c4_group4_ms"] = round(v["ms_per_fit"], 2)
    ml = oc.get("mlii", {})
    for k, v in ml.items():
        if k.startswith("n=8192 lockstep") and isinstance(v, dict):
            c["mlii_g40_ms"] = round(v["ms_per_evaluation"], 3); c["mlii_g40_frac"] = round(v["frac_of_fp64_mfma_peak"], 4)
    if "sharded" in out and isinstance(out["sharded"], dict):
        c["sharded"] = {k: ({"ms_per_fit": round(v.get("ms_per_fit", 0.0), 3), "row_split_ms_per_fit": round(v.get("row_split_exchange", {}).get("ms_per_fit", 0.0), 3),
                             "speedup_vs_1gpu": round(v.get("single_gpu", {}).get("speedup_of_sharded", 0.0), 3),
                             "transport": str(v.get("transport"))[:40]} if isinstance(v, dict) and "ms_per_fit" in v else str(v)[:80]) for k, v in out["sharded"].items()}
    c["verbose"] = "stderr; gpurun_out/bench_verbose.json"
    return c


def emit(out):
    """Rank 0: the full record to stderr and gpurun_out/bench_verbose.json, then -- LAST, and the only line on stdout -- the compact metric line."""
    txt = json.dumps(out)
    print(txt, file=sys.stderr, flush=True)
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_verbose.json"), "w") as f:
            f.write(txt + "\n")
    except OSError:
        pass
    line = json.dumps(compact_line(out))
    print(line, flush=True)


def kernel_code_sha16():
    """sha256 (first 16 hex digits) of the device code the roofline kernel is compiled from."""
    import hashlib
    hs = hashlib.sha256()
    csrc = os.path.join(ROOT, "seaiceextentforecasting_amd", "csrc")
    for f in ("syrk128.hpp", "gemm_mfma.hpp"):       # (what syrk128_kernel is compiled from)
        with open(os.path.join(csrc, f), "rb") as fh:
            hs.update(fh.read())
    return hs.hexdigest()[:16]


def sharded_record(rank, world, local, dist, backend, configs=("configs[3]", "configs[4]"), reps=3, outer=8):
    """BASELINE configs[3] (n=16384, d=16 fp64 RBF) and configs[4] (n=32768, d=32 fp32 Matern-5/2 + fp64 refinement) as ONE fit
    sharded over all `world` ranks: the library's own sharded fit (sigp_dist_fit: block-cyclic panels, panel broadcast on the
    library's RCCL communicator with look-ahead; fp32: solves on the distributed factor, residual sharded by rows).  Every rank
    calls this; rank 0 returns the record.  Time per fit = max over ranks of the best of `reps` (after one warm-up fit)."""
    import torch
    from seaiceextentforecasting_amd import DistributedGPR, GPR
    shapes = {"configs[3]": ("rbf", "f64", 16384, 16, 4.0, 1e-2, 20240003, PEAK_F64_MFMA_TFLOPS),
              "configs[4]": ("matern52", "f32", 32768, 32, float(np.sqrt(32.0)), 1e-1, 20240004, PEAK_F32_MFMA_TFLOPS)}
    rec = {}
    dev = "cuda" if backend == "nccl" else "cpu"
    for tag in configs:
        kind, dtype, n, d, ell, sn, seed, peak = shapes[tag]
        X, y, Xs = synthetic_problem(n, d, seed, m=1)
        with DistributedGPR(kind, rank, world, dist, device=local, outer_blocks=outer, dtype=dtype, stats=True) as dg:
            times = []
            dg.fit(X, y, ell, sn, Xs=Xs)          # stages X, y, Xs on every rank, allocates, opens the ring
            for r_ in range(reps + 1):
                torch.cuda.synchronize()
                if dist is not None:
                    dist.barrier()
                t0 = time.perf_counter()
                dg.refit(ell, sn)
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
                if dist is not None:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                if r_ > 0:
                    times.append(float(t.item()))
            st = dg.stats()
            # A/B for the streamed broadcast: the same fit with whole-panel broadcasts (dist_segment >= W)
            dg.gp.set_option("dist_segment", 64)
            whole = []
            for r_ in range(reps):
                torch.cuda.synchronize()
                if dist is not None:
                    dist.barrier()
                t0 = time.perf_counter()
                dg.refit(ell, sn)
                t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
                if dist is not None:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                whole.append(float(t.item()))
            dg.gp.set_option("dist_segment", 2)
            dg.refit(ell, sn)
            mu, var = dg.predict(Xs)
            keep = dict(sigma_f=dg.sigma_f_, nlml=dg.nlml_, mean=float(mu[0]), var=float(var[0]), matrix_bytes=dg.matrix_bytes_, transport=dg.transport)
            if dtype == "f32":
                keep["refinement_residual"] = dg.refine_residual_
            # ... and for the panel exchange by ROW PIECES + all-gather (dist_panel_split: the owner factors only the top block, every rank solves
            # 1/world of the rows below it): the same results, another critical path.  LAST, and in a try of its own: whatever happens to it
            # (a collective that times out marks the handle dead) must not cost the record the numbers above
            splitt, st_split, split_nlml, split_err = [], dict(st), float("nan"), None
            try:
                dg.gp.set_option("dist_panel_split", 1)
                dg.refit(ell, sn)
                for r_ in range(reps):
                    torch.cuda.synchronize()
                    if dist is not None:
                        dist.barrier()
                    t0 = time.perf_counter()
                    dg.refit(ell, sn)
                    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
                    if dist is not None:
                        dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    splitt.append(float(t.item()))
                st_split = dg.stats()
                split_nlml = dg.nlml_
            except Exception as e:               # noqa: BLE001
                split_err = "%s: %s" % (type(e).__name__, str(e)[:300])
                splitt = splitt or [float("nan")]
            res = keep
        # per-rank numbers worth a max / sum over the ranks
        v = torch.tensor([st["stall_ms"], st["comm_ms"], st["factor_ms"], st["solve_ms"], res["matrix_bytes"], st["owner_ms"], st["link_bytes"], st_split["owner_ms"],
                          st_split["link_bytes"], st_split["stall_ms"], st["link_panel_max"], st_split["link_panel_max"]], dtype=torch.float64, device=dev)
        vmax = v.clone()
        if dist is not None:
            dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
        best = min(times)
        e = {"workload": "%s: n=%d d=%d %s %s, ONE fit sharded over %d rank(s): 1-D block-cyclic panels of %d x 128 columns, owner-only storage" % (tag, n, d, dtype, kind, world, outer),
             "transport": ("library RCCL communicator (ncclCommCount = %d)" % int(st["comm_ranks"])) if res["transport"] == "rccl" else res["transport"],
             "ms_per_fit": 1e3 * best, "ms_per_fit_all": [round(1e3 * t, 3) for t in times], "fits_per_s": 1.0 / best,
             "tflops": flops_per_fit(n, d) / best / 1e12, "frac_of_peak_all_gpus": flops_per_fit(n, d) / best / 1e12 / (peak * world),
             "panel_broadcast_bytes_per_fit": st["bcast_bytes"], "collectives_per_fit": st["collectives"],
             "ms_per_fit_with_whole_panel_broadcasts": 1e3 * min(whole), "streamed_segments_gain": min(whole) / best,
             "max_over_ranks_ms": {"update_stream_stalled_on_a_panel": float(vmax[0]), "communication_window_first_segment_ready_to_last_arrived": float(vmax[1]), "panel_loop_device_time": float(vmax[2]),
                                   "reductions_solves_refinement_host_time": float(vmax[3])},
             "share_of_communication_window_with_update_work": (1.0 - float(vmax[0]) / float(vmax[1])) if float(vmax[1]) > 0 else None,
             "matrix_bytes_max_rank": float(vmax[4]), "sigma_f": res["sigma_f"], "nlml": res["nlml"], "mean": res["mean"], "var": res["var"],
             "panels": int(-(-(n // 128) // outer)),
             "whole_panel_exchange": {"owner_only_ms_per_fit_max_rank": float(vmax[5]), "bytes_per_directed_link_per_fit_max_rank": float(vmax[6]),
                                      "owner_only_ms_per_panel": float(vmax[5]) * world / max(1, -(-(n // 128) // outer)), "bytes_on_one_link_within_one_panel_max": float(vmax[10])},
             "row_split_exchange": {"ms_per_fit": 1e3 * min(splitt), "ms_per_fit_all": [round(1e3 * t, 3) for t in splitt], "gain_over_streamed_segments": best / min(splitt),
                                    "owner_only_ms_per_fit_max_rank": float(vmax[7]), "bytes_per_directed_link_per_fit_max_rank": float(vmax[8]),
                                    "owner_only_ms_per_panel": float(vmax[7]) * world / max(1, -(-(n // 128) // outer)), "bytes_on_one_link_within_one_panel_max": float(vmax[11]),
                                    "update_stream_stalled_ms_max_rank": float(vmax[9]), "split_panels": int(st_split["split_panels"]),
                                    "rel_diff_nlml_vs_whole_panel_exchange": float(abs(split_nlml - res["nlml"]) / abs(res["nlml"])),
                                    "error": split_err,
                                    "note": "dist_panel_split = 1: top block broadcast (8 MB at W = 8), rows below scattered in `world` pieces, solved where they land, all-gathered in place"}}
        if "refinement_residual" in res:
            e["refinement_residual"] = res["refinement_residual"]
        if rank == 0:
            # the same fit through the single-GPU entry point on rank 0's GPU: what the sharded numbers are checked and priced against
            with GPR(kernel=kind, dtype=dtype, device=local) as g1:
                g1.fit(X, y, ell, sn, Xs=Xs)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                g1.refit(ell, sn)
                torch.cuda.synchronize(); t1 = time.perf_counter() - t0
                m1, v1 = g1.predict(Xs)
                e["single_gpu"] = {"ms_per_fit": 1e3 * t1, "speedup_of_sharded": t1 / best,
                                   "rel_diff_mean": float(abs(res["mean"] - m1[0]) / abs(m1[0])), "rel_diff_var": float(abs(res["var"] - v1[0]) / abs(v1[0])),
                                   "rel_diff_nlml": float(abs(res["nlml"] - g1.nlml_) / abs(g1.nlml_))}
        if dist is not None:
            dist.barrier()
        rec[tag] = e
    return rec if rank == 0 else None


def timed(fn, reps, warm=1):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts


def other_configs(local):
    """The other BASELINE configurations on ONE GPU (untimed extras, not the metric): latency of a single fit at
    configs[1] / [3] / [4] shape with its roofline fraction, the MLII (nlML + exact gradient) evaluation, and the
    reference's own kernel at the reference's own size batched over the 20x20 grid."""
    from seaiceextentforecasting_amd import GPR
    rec = {}

    def single(tag, kernel, dtype, n, d, ell, sn, seed, reps, peak):
        X, y, Xs = synthetic_problem(n, d, seed, m=1)
        with GPR(kernel=kernel, dtype=dtype, device=local) as g:
            g.fit(X, y, ell, sn, Xs=Xs)
            g.synchronize(); t0 = time.perf_counter()
            for _ in range(reps):
                g.refit(ell, sn)
            g.synchronize(); dt = (time.perf_counter() - t0) / reps
            e = {"ms_per_fit": 1e3 * dt, "fits_per_s": 1.0 / dt, "tflops": flops_per_fit(n, d) / dt / 1e12,
                 "frac_of_peak": flops_per_fit(n, d) / dt / 1e12 / peak, "peak_tflops": peak, "note": "one fit at a time (latency), GPR.refit on resident data"}
            if dtype == "f32":
                e["refinement_residual"] = g.refine_residual_
        rec[tag] = e
        return X, y

    single("configs[1] n=4096 d=8 fp64 RBF single fit", "rbf", "f64", 4096, 8, np.sqrt(8.0), 1e-2, 20240001, 5, PEAK_F64_MFMA_TFLOPS)
    single("configs[3] n=16384 d=16 fp64 RBF single fit on one GPU", "rbf", "f64", 16384, 16, 4.0, 1e-2, 20240003, 2, PEAK_F64_MFMA_TFLOPS)
    single("configs[4] n=32768 d=32 fp32 Matern-5/2 + fp64 refinement on one GPU", "matern52", "f32", 32768, 32, np.sqrt(32.0), 1e-1, 20240004, 2, PEAK_F32_MFMA_TFLOPS)
    # configs[4] shape in a lockstep group of 4 (one build + one blocked fp32 Cholesky over the members, refinement member by member)
    try:
        n4, d4, G4 = 32768, 32, 4
        Xb = np.zeros((G4, n4, d4)); yb = np.zeros((G4, n4)); Xsb = np.zeros((G4, 1, d4))
        for b in range(G4):
            Xb[b], yb[b], Xsb[b] = synthetic_problem(n4, d4, 20240004 + b, m=1)
        with GPR(kernel="matern52", dtype="f32", device=local) as g:
            e4 = np.full(G4, np.sqrt(d4)); s4 = np.full(G4, 1e-1)
            g.upload_batch(Xb, yb, Xsb, group=G4, concurrency=1)
            g.run_batch(0, G4, e4, s4, concurrency=1, group=G4)
            g.synchronize(); t0 = time.perf_counter()
            r4 = g.run_batch(0, G4, e4, s4, concurrency=1, group=G4)
            g.synchronize(); dt = (time.perf_counter() - t0) / G4
        assert np.all(r4["info"] == 0)
        rec["configs[4] shape, lockstep group of 4 on one GPU"] = {"ms_per_fit": 1e3 * dt, "fits_per_s": 1.0 / dt, "tflops": flops_per_fit(n4, d4) / dt / 1e12,
                                                                 "frac_of_peak": flops_per_fit(n4, d4) / dt / 1e12 / PEAK_F32_MFMA_TFLOPS, "peak_tflops": PEAK_F32_MFMA_TFLOPS}
        del Xb, yb, Xsb
    except Exception as e:                       # an extra record must never cost the metric line
        rec["configs[4] shape, lockstep group of 4 on one GPU"] = {"error": str(e)[:200]}
    # MLII: nlML + exact gradient (north/June1st.py:235-257 with the true derivative), the O(n^3) hot spot of an optimiser run
    ml = {}
    for n in (4096, 8192):
        X, y, _ = synthetic_problem(n, 8, 20240001, m=1)
        th = np.log([np.sqrt(8.0), 1e-2])
        with GPR(kernel="rbf", device=local) as g:
            g.set_data(X, y)
            g.nlml(th, grad="exact")
            g.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                g.nlml(th, grad="exact")
            g.synchronize(); dt = (time.perf_counter() - t0) / 3
            fl = mlii_flops(n, 8)
            ml["n=%d" % n] = {"ms": 1e3 * dt, "tflops": fl / dt / 1e12, "frac_of_fp64_mfma_peak": fl / dt / 1e12 / PEAK_F64_MFMA_TFLOPS,
                              "algorithmic_flops": fl}
    # the same evaluation for the 40 retrospective years in lockstep (sigp_nlml_grad_batch): what a multi-start optimiser calls per iteration
    try:
        for n, G in ((4096, 40), (8192, 40)):
            Xb = np.zeros((G, n, 8)); yb = np.zeros((G, n))
            for b in range(G):
                Xb[b], yb[b], _ = synthetic_problem(n, 8, 20240002 + b, m=1)
            th = np.tile(np.log([np.sqrt(8.0), 1e-2]), (G, 1))
            with GPR(kernel="rbf", device=local) as g:
                g.upload_batch(Xb, yb, None, group=G, concurrency=1)
                g.nlml_batch(th, grad="exact", group=G)
                g.synchronize(); t0 = time.perf_counter()
                g.nlml_batch(th, grad="exact", group=G)
                g.synchronize(); dt = (time.perf_counter() - t0) / G
            fl = mlii_flops(n, 8)
            ml["n=%d lockstep group of %d" % (n, G)] = {"ms_per_evaluation": 1e3 * dt, "tflops": fl / dt / 1e12, "frac_of_fp64_mfma_peak": fl / dt / 1e12 / PEAK_F64_MFMA_TFLOPS,
                                                      "evaluations_per_s": 1.0 / dt}
            if n == 4096:
                # the optimiser the reference left commented out (north/June1st.py:259-262), for all 40 years at once: BFGS on (log l, log sn~)
                # per year, ONE lockstep device call per round (GPR.optimize_batch)
                with GPR(kernel="rbf", device=local) as g:
                    g.synchronize(); t0 = time.perf_counter()
                    ro = g.optimize_batch(Xb, yb, np.log([np.sqrt(8.0), 1e-1]), group=G, maxiter=30)
                    g.synchronize(); to = time.perf_counter() - t0
                ml["optimise 40 years, n=4096"] = {"seconds": to, "device_calls": int(ro["nfev"]), "iterations_per_year_mean": float(np.mean(ro["nit"])),
                                                   "converged_years": int(np.sum(ro["converged"])), "nlml_mean_at_optimum": float(np.mean(ro["fun"])),
                                                   "note": "includes the upload of the 40 data sets; every device call evaluates nlML + exact gradient for all 40 years in lockstep"}
            del Xb, yb
    except Exception as e:                       # noqa: BLE001 -- an extra record must never cost the metric line
        ml["lockstep_error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
    rec["reference_kernel_grid"] = reference_kernel_grid(local)
    ml["note"] = "one MLII evaluation = fit (n^3/3) + L~^-T by recursive triangular inversion (n^3/3) + lower K~^-1 = U U^T (n^3/3) + O(n^2 d) derivative/reductions"
    rec["mlii"] = ml
    return rec


def reference_kernel_grid(local):
    """The reference's OWN kernel at the reference's OWN size: the retro loop's 3 regions x 40 years
    (n = 6 .. 45 training years, N = 60 / 20 / 12 network areas) x the 20 x 20 grid of north/June1st.py:210-211 = 48 000 fits
    in ONE launch (one workgroup per fit), beside the oracle's loop over a sample of the same fits on the host."""
    from seaiceextentforecasting_amd import GPR, SmallBatch, LGRID, SGRID
    rng = np.random.default_rng(20240010)
    sets = []
    for N in (60, 20, 12):
        for t in range(40):
            n = 6 + t
            X = rng.standard_normal((n, N)) * (1.0 + 0.3 * rng.standard_normal(N))
            y = X @ rng.standard_normal(N) / np.sqrt(N) + 0.5 * rng.standard_normal(n)
            sets.append((X, y, rng.standard_normal((1, N))))
    with GPR(kernel="netdiffusion", device=local) as gp:
        t0 = time.perf_counter()
        sb = SmallBatch(gp)
        for X, y, Xs in sets:
            ds = sb.add_dataset(X, y, Xs)
            for e in LGRID:
                for s_ in SGRID:
                    sb.add_fit(ds, e, s_, expm="eigh")
        sb.upload()
        t_stage = time.perf_counter() - t0
        r = sb.run()                                   # warm-up
        gp.synchronize(); t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            r = sb.run()
        gp.synchronize(); dt = (time.perf_counter() - t0) / reps
    F = len(r["nlml"])
    flops = sum(400 * (n * n * N + n ** 3 / 3 + 4 * n * n) for (X, _, _) in sets for n, N in [X.shape])
    out = {"fits": F, "ok_fits": int(np.sum(r["info"] == 0)), "ms_per_launch": 1e3 * dt, "fits_per_s": F / dt, "host_staging_s": t_stage,
           "gflops": flops / dt / 1e9,
           "note": "smallgp_kernel: K~ from the factored covariance + Cholesky + ride-along solves in LDS; latency/LDS-bound (orders <= 45), not an MFMA kernel; "
                   "time includes the H2D copy of the fit list and the D2H copy of the results; host staging = one eigh of M per data set + packing"}
    try:
        from oracle import gp_oracle as O
        sample = [sets[39], sets[79], sets[119]]        # the largest year of each region
        t0 = time.perf_counter(); cnt = 0
        for X, y, Xs in sample:
            M = O.laplacian_M(X)
            for e in LGRID[::4]:
                for s_ in SGRID[::4]:
                    try:
                        O.fit_predict(X, y, Xs, e, s_, kind="netdiffusion", M=M, ref_idiom=True)
                    except Exception:
                        pass
                    cnt += 1
        tc = time.perf_counter() - t0
        out["cpu_oracle_loop"] = {"fits_per_s": cnt / tc, "sample": "%d fits (3 data sets x 5 x 5 grid points), oracle ref_idiom=True (two expm + two Cholesky + gesv solves per fit, north/June1st.py:264-277)" % cnt}
    except ImportError:
        pass
    return out


def mlii_flops(n, d):
    return flops_per_fit(n, d, 0) + 2 * n ** 3 / 3 + n * n * (3 * d + 30)


def cpu_baseline(out, args, Xb, yb, Xsb, ell, sn, W, nsets, r, local):
    """The reference-idiom CPU path (oracle, call-for-call north/June1st.py:264-277) on the host cores, on a bounded sample:
    BASELINE.md section 2 protocol (1 warm-up + 3 timed, median) at n = 64 and 4096; at n = 8192 --cpu-reps timed fits
    (default 1: a reference-idiom fit takes ~48 s there) of the FIRST TIMED STEP's inputs."""
    from seaiceextentforecasting_amd import GPR
    n, d = args.n, args.d
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
    env = "NumPy %s / SciPy %s / OpenBLAS, %d BLAS threads, os.cpu_count()=%d, CPU: %s" % (np.__version__, scipy.__version__, ncpu, os.cpu_count(), cpu_model)
    proto = {}
    for nn, dd, seed in ((64, 4, 20240000), (4096, 8, 20240001)):
        Xc, yc, Xsc = synthetic_problem(nn, dd, seed, m=1)
        e0, s0 = float(np.sqrt(dd)), 1e-2
        med_ref, _ = timed(lambda: O.fit_predict(Xc, yc, Xsc, e0, s0, kind="rbf", ref_idiom=True), 3)
        med_best, _ = timed(lambda: O.fit_predict(Xc, yc, Xsc, e0, s0, kind="rbf", ref_idiom=False), 3)
        proto["n=%d d=%d" % (nn, dd)] = {"reference_idiom_fits_per_s": 1.0 / med_ref, "reference_idiom_s": med_ref,
                                         "best_practice_fits_per_s": 1.0 / med_best, "best_practice_s": med_best, "protocol": "1 warm-up + 3 timed, median"}
    # n = 8192: the sample is the FIRST TIMED STEP of the run above (its first member: data set W % nsets, hyper-parameters of fit W)
    nb = n if n <= 8192 else 8192
    ds0 = W % nsets
    if nb == n:
        Xc, yc, Xsc = Xb[ds0], yb[ds0], Xsb[ds0]
    else:
        Xc, yc, Xsc = synthetic_problem(nb, d, 20240002, m=1)
    ell0, sn0 = float(ell[W]), float(sn[W])
    reps = max(1, args.cpu_reps)
    ref = {}
    def run_ref():
        ref["r"] = O.fit_predict(Xc, yc, Xsc, ell0, sn0, kind="rbf", ref_idiom=True)
    tc, ts = timed(run_ref, reps, warm=1 if reps > 1 else 0)
    ref = ref["r"]
    out["cpu_baseline"] = {"value": 1.0 / tc, "unit": "fits/s", "cores": ncpu, "kind": "port",
                           "sample": "%d timed fit(s)%s (the first timed step: data set %d, l=%.4g, sn~=%.3g), n=%d d=%d, oracle ref_idiom=True = the reference's call sequence north/June1st.py:264-277 (%s)"
                                     % (reps, " after 1 warm-up, median" if reps > 1 else ", cold", ds0, ell0, sn0, nb, d, env),
                           "seconds": tc, "protocol_other_sizes": proto}
    tb, _ = timed(lambda: O.fit_predict(Xc, yc, Xsc, ell0, sn0, kind="rbf", ref_idiom=False), 1, warm=1 if reps > 1 else 0)
    out["cpu_baseline_best_practice"] = {"value": 1.0 / tb, "unit": "fits/s", "cores": ncpu, "kind": "port", "seconds": tb,
                                         "sample": "same inputs, 1 timed fit%s, oracle ref_idiom=False (one Cholesky + scipy solve_triangular)" % (" after 1 warm-up" if reps > 1 else ", cold")}
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


if __name__ == "__main__":
    main()
