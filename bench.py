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

This file is the TIMED path and the line it prints; every untimed extra record (other configs, MLII, the reference kernel's grid and
optimiser, the sharded fits, the CPU leg) lives in bench_extras.py and runs after the timed region.
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



def pick_group(args, local, world, years, n):
    """The lockstep group the run can hold.  The group's matrices are ONE allocation of G (n + 128) n doubles (86 GB at the defaults): on
    a GPU that does not have that much free (shared box, another tenant) fall back to fewer grid points per launch -- whole years at a
    time -- instead of failing; the line then carries `lockstep_group_asked` != `fits_per_step`.  SIGP_BENCH_FREE_BYTES overrides what
    hipMemGetInfo reports (tests)."""
    G = max(1, args.group)
    try:
        import ctypes
        from seaiceextentforecasting_amd import _lib as _L
        hip = ctypes.CDLL(_L.runtime_info().split(" from ")[-1].strip())      # the runtime already mapped into this process (never a second one)
        free_b, total_b = ctypes.c_size_t(0), ctypes.c_size_t(0)
        ndev = ctypes.c_int(0)
        if hip.hipSetDevice(int(local)) == 0 and hip.hipMemGetInfo(ctypes.byref(free_b), ctypes.byref(total_b)) == 0 and hip.hipGetDeviceCount(ctypes.byref(ndev)) == 0:
            free = float(os.environ.get("SIGP_BENCH_FREE_BYTES", free_b.value))
            per_member = (n + 128) * n * 8.0 * 1.02
            reserve = 70e9 if not args.no_extras else 8e9          # the untimed extras (fp32 group of 4, MLII group of 40) allocate beside it
            share = -(-world // max(1, ndev.value))                # ranks of a rehearsal that share this GPU (1 in a real launch)
            while G > years and G * per_member + reserve > free / share:
                G -= years
    except Exception:
        pass
    return G


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
    if world > 1 and rank == 0 and not args.no_sharded:
        start_line_guardian()
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
    G_asked = max(1, args.group)
    G = pick_group(args, local, world, years, n)                   # fits per step
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
    # ================================================= the timed region =================================================
    barrier()
    t0 = time.perf_counter()
    r = gp.run_batch(W, K, ell[W:], sn[W:], concurrency=args.concurrency, group=args.group)
    sync()
    t1 = time.perf_counter()
    barrier()
    # =====================================================================================================================
    prof = gp.profile_get()
    gp.profile(False)
    elapsed = t1 - t0
    assert np.all(r["info"] == 0) and np.all(np.isfinite(r["mean"]))
    import bench_extras as X_
    ctx = dict(gp=gp, args=args, rank=rank, world=world, W=W, K=K, ell=ell, sn=sn, Xb=Xb, yb=yb, my_years=my_years, d=d, sync=sync, barrier=barrier)
    unbracketed = X_.unbracketed_rerun(ctx) if (not args.no_profile and not args.no_extras) else None
    if dist is not None:
        t = torch.tensor([elapsed, unbracketed or 0.0], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        if unbracketed is not None:
            unbracketed = float(t[1].item())

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
                               "flops_note": "algorithmic: 2*128*128*K per off-diagonal tile, the lower half (128*129*K) per diagonal tile, 2*(1+m)*128*K per tile of the "
                                             "ride-along block row (its rows in use; up to round 4 those tiles were counted -- and multiplied -- whole: x 1.049 for that accounting)"}
            quote_pmc_traffic(out["roofline"], dom, n, d, args)
        else:
            out["roofline"] = None
        if not args.no_profile:
            X_.untimed_kernel_records(ctx, out, prof)              # per-kernel table, the kernel's rate with the strips beside it, m = 64
    gp.close()

    if rank == 0 and world == 1 and not args.no_extras:
        try:
            out["other_configs"] = X_.other_configs(local)
        except Exception as e:                   # untimed extras must never cost the metric line
            out["other_configs"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_reps"] = max(1, args.cpu_reps)
        X_.cpu_baseline(out, args, Xb, yb, Xsb, ell, sn, W, len(my_years), r, local)
        if out.get("cpu_baseline", {}).get("value"):
            out["vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]     # (vs_baseline stays null: the reference publishes no number for this metric)
    if world > 1 and not args.no_sharded:
        if rank == 0:
            emit_provisional(out)
        X_.sharded_with_watchdog(out, args, rank, world, local, dist, backend, emit)
    if rank == 0:
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


def quote_pmc_traffic(rf, dom, n, d, args):
    """HBM-side traffic of the dominant kernel is NOT measured in this run: it comes from separate rocprofv3 --pmc passes of this command
    (FETCH_SIZE / WRITE_SIZE cannot share a pass, and PMC collection serialises kernels), summarised under profiles/ by
    tools/collect_profiles.sh + tools/summarize_pmc.py -- and quoted only when the file was collected from THIS kernel code (sha of the
    csrc files recorded at collection time) at this group size."""
    for rnd in ("r05", "r04", "r03", "r02"):
        pth = os.path.join(ROOT, "profiles", "%s_pmc_syrk128.json" % rnd)
        if not os.path.exists(pth):
            continue
        try:
            pm = json.load(open(pth))
            if n == 8192 and d == 8 and args.group == pm.get("lockstep_group", 40) and args.outer == 8:     # (per-launch bytes scale with the members per launch)
                rf["algorithmic_c_bytes_per_launch"] = dom["bytes"] / dom["launches"]
                if pm.get("kernel_code_sha16") == kernel_code_sha16():
                    rf["traffic"] = pm["traffic_bytes_per_launch"]
                    rf["traffic_over_algorithmic"] = pm["traffic_bytes_per_launch"] / (dom["bytes"] / dom["launches"])
                    rf["traffic_source"] = ("from profiles/%s_pmc_syrk128.json (rocprofv3 --pmc passes of this command on this kernel code, sha16 %s; NOT this run): bytes per launch = "
                                            "FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE.  %.1f x the algorithmic C bytes: the surplus is the panels' K-slices re-fetched past the 4 MB L2 "
                                            "(served by the Infinity Cache).  Closed by measurement: XCD-chunked tile walks cut the fetch to 4.9-6.3 GB per launch and the clock does not rise "
                                            "(MFMA-bound kernel, pipe busy %.2f): fits/s fell 1 %% (profiles/r04_syrk128_traffic_vs_clock.txt).  17.2 GB of it before the diagonal / ride-row tile forms: "
                                            "those tiles are shorter and shift the phase of the tiles that share a B block through L2 (flag test: 16.7 GB with the forms off); a walk that puts "
                                            "them at the end of the launch restores 19.3 GB and changes fits/s by nothing (docs/EXPERIMENTS.md)"
                                            % (rnd, pm["kernel_code_sha16"], pm["traffic_bytes_per_launch"] / (dom["bytes"] / dom["launches"]), pm.get("mfma_pipe_busy_fraction", 0.0)))
                else:
                    rf["traffic_source"] = ("profiles/%s_pmc_syrk128.json was collected from other kernel code (sha16 %s, now %s): not quoted"
                                            % (rnd, pm.get("kernel_code_sha16"), kernel_code_sha16()))
        except Exception:
            pass
        break


def compact_line(out):
    """The metric line the driver records (it keeps a 2 000-character tail): the contract's fields, `roofline`, `cpu_baseline`, the parity of
    the timed step and one number per other BASELINE configuration -- at most 1.5 KB.  The full record goes to stderr and to
    gpurun_out/bench_verbose.json."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    c = {k: out.get(k) for k in keep}
    cfg = out.get("config", {})
    c["config"] = {"workload": "configs[2]: n=%s d=%s fp64 RBF GPR, %s retrospective years x hyper-parameter grid, one step = fits_per_step fits in lockstep"
                               % (cfg.get("n"), cfg.get("d"), cfg.get("years_resident_per_rank")),
                   "fits_per_step": cfg.get("fits_per_step"), "lockstep_group_asked": cfg.get("lockstep_group_asked"), "grid": cfg.get("grid"), "parallelism": "years sharded over %s GPU(s), no data-path collective" % out.get("n_gpus")}
    rf = out.get("roofline")
    if rf:
        c["roofline"] = {"bound": rf["bound"], "kernel": "syrk128_kernel<double> (trailing update, v_mfma_f64_16x16x4_f64)", "achieved": round(rf["achieved"], 3), "peak": rf["peak"],
                         "unit": rf["unit"], "frac": round(rf["frac"], 4), "traffic": rf.get("traffic"), "avg_launch_ms": round(rf["avg_launch_ms"], 4), "launches": rf["launches"]}
        if rf.get("strips_beside_update"):
            c["roofline"]["strips_beside_update_frac"] = round(rf["strips_beside_update"]["frac"], 4)
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
    rk = oc.get("reference_kernel_mlii") or {}
    if "evaluations_per_s" in rk:
        c["refk_mlii_evals_per_s"] = round(rk["evaluations_per_s"], 0)
    c["verbose"] = "stderr; gpurun_out/bench_verbose.json"
    return c


_guard_fd = None      # rank 0, N > 1: write end of the pipe to the child that owns stdout's one line
_guard_pid = None


def start_line_guardian():
    """N > 1, rank 0, BEFORE anything touches the GPU (a plain fork, no exec): a child that owns the metric line.  Rank 0 hands it
    the line measured so far before it enters the untimed sharded fits (never yet run on more than one real GPU) and the final line
    after them; the child prints the LAST line it was given when the pipe closes -- so stdout carries exactly one line whether rank 0
    finishes, is cut off by the watchdog, or dies inside a collective."""
    global _guard_fd, _guard_pid
    r, w = os.pipe()
    sys.stdout.flush(); sys.stderr.flush()
    pid = os.fork()
    if pid == 0:
        os.close(w)
        last = b""
        with os.fdopen(r, "rb") as f:
            for ln in f:
                if ln.strip():
                    last = ln
        if last:
            os.write(1, last if last.endswith(b"\n") else last + b"\n")
        os._exit(0)
    os.close(r)
    _guard_fd, _guard_pid = w, pid


def guardian_hand_over(line, final):
    """The metric line goes to the guardian (or straight to stdout when there is none); `final` closes the pipe and waits for the child's print."""
    global _guard_fd
    if _guard_fd is None:
        if final:
            print(line, flush=True)
        return
    os.write(_guard_fd, line.encode() + b"\n")
    if final:
        os.close(_guard_fd)
        _guard_fd = None
        try:
            os.waitpid(_guard_pid, 0)
        except OSError:
            pass


def emit_provisional(out):
    """Before the sharded record: the line as it stands, for the guardian to print should this process not come back."""
    tmp = dict(out)
    tmp["sharded"] = {"error": "rank 0 did not return from the sharded record (line handed over before it)"}
    guardian_hand_over(json.dumps(compact_line(tmp)), final=False)


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
    guardian_hand_over(json.dumps(compact_line(out)), final=True)


def kernel_code_sha16():
    """sha256 (first 16 hex digits) of the device code the roofline kernel is compiled from."""
    import hashlib
    hs = hashlib.sha256()
    csrc = os.path.join(ROOT, "seaiceextentforecasting_amd", "csrc")
    for f in ("syrk128.hpp", "gemm_mfma.hpp"):       # (what syrk128_kernel is compiled from)
        with open(os.path.join(csrc, f), "rb") as fh:
            hs.update(fh.read())
    return hs.hexdigest()[:16]


if __name__ == "__main__":
    main()
