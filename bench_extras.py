"""bench_extras.py -- the UNTIMED records bench.py adds to its line after the timed region: the dominant kernel's per-kernel table and its
rate when it does not share the chip, the m = 64 variant, the other BASELINE configurations on one GPU, MLII and the lockstep
optimisers (RBF on the blocked engine; the reference's own kernel one workgroup per evaluation), the reference kernel's 20 x 20 grid,
configs[3] / [4] as ONE fit sharded over the ranks, and the CPU leg (the oracle in the reference's call sequence on the host cores).
Nothing in here runs inside bench.py's timed region, and nothing here may cost the metric line (every record fails into an "error" entry).
"""
import os
import time

import numpy as np

from bench import (PEAK_F32_MFMA_TFLOPS, PEAK_F64_MFMA_TFLOPS, flops_per_fit, synthetic_problem)


def unbracketed_rerun(ctx):
    """The same timed steps once more WITHOUT the per-launch HIP-event brackets of the dominant kernel (`value` carries them)."""
    gp, args, W, K, ell, sn = ctx["gp"], ctx["args"], ctx["W"], ctx["K"], ctx["ell"], ctx["sn"]
    ctx["barrier"](); ta = time.perf_counter()
    gp.run_batch(W, K, ell[W:], sn[W:], concurrency=args.concurrency, group=args.group)
    ctx["sync"](); tub = time.perf_counter() - ta
    ctx["barrier"]()
    return tub


def untimed_kernel_records(ctx, out, prof):
    """Rank 0, after the timed region: the per-kernel table (one extra group with every launch bracketed), the dominant kernel's rate when the
    strip solve shares the chip with it (`roofline.strips_beside_update`), and SURVEY 8(d)'s m = 64 variant."""
    gp, args, W, K, ell, sn, sync = ctx["gp"], ctx["args"], ctx["W"], ctx["K"], ctx["ell"], ctx["sn"], ctx["sync"]
    gp.profile(True); gp.profile_reset()
    kk = min(K, args.group)
    gp.run_batch(W, kk, ell[W:W + kk], sn[W:W + kk], concurrency=1, group=args.group)
    src = gp.profile_get(); gp.profile(False)
    out["kernels"] = {k: {"ms_per_fit": v["ms"] / kk, "launches_per_fit": v["launches"] / kk,
                          "tflops": (v["flops"] / (v["ms"] * 1e-3) / 1e12) if v["ms"] > 0 else None,
                          "GBps": (v["bytes"] / (v["ms"] * 1e-3) / 1e9) if v["ms"] > 0 else None} for k, v in src.items()}
    out["kernels_note"] = "per-kernel table from one extra untimed lockstep group with every launch bracketed; roofline from the timed region"
    if ctx["world"] != 1 or args.no_extras:
        return
    # the other placement of the strip solve: BESIDE the rest of the trailing update (option strips_after_update = 0, the default up to
    # round 4).  One extra, untimed pair of groups: same fits/s, but the trailing update's launches then last as long as the two kernels'
    # MFMA work together, which is why the default now runs the strips behind the update.
    gp.set_option("strips_after_update", 0)
    kk = min(K, 2 * args.group)
    gp.run_batch(W, kk, ell[W:W + kk], sn[W:W + kk], concurrency=1, group=args.group)
    gp.profile(True, classes=["syrk128"]); gp.profile_reset()
    sync(); ta = time.perf_counter()
    gp.run_batch(W, kk, ell[W:W + kk], sn[W:W + kk], concurrency=1, group=args.group)
    sync(); tu = time.perf_counter() - ta
    pu = gp.profile_get()["syrk128"]; gp.profile(False)
    gp.set_option("strips_after_update", 1)
    if pu["ms"] > 0 and out.get("roofline"):
        au = pu["flops"] / (pu["ms"] * 1e-3) / 1e12
        out["roofline"]["strips_beside_update"] = {"achieved": au, "frac": au / PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "launches": pu["launches"],
                                                   "avg_launch_ms": pu["ms"] / pu["launches"], "fits_per_s_with_this_schedule": kk / tu,
                                                   "note": "the same kernel with the panel stream's strip solve (MFMA work for the whole chip) running beside the trailing update "
                                                           "(strips_after_update = 0): the launches share the matrix pipe with it and last longer; fits/s is the same -- the step "
                                                           "is bound by the sum of its MFMA work either way"}
    # "m = 1 (also report m = 64)" -- the same steps with 64 test points riding along each fit
    rng = np.random.default_rng(7)
    Xs64 = rng.standard_normal((len(ctx["my_years"]), 64, ctx["d"]))
    gp.upload_batch(ctx["Xb"], ctx["yb"], Xs64, group=args.group, concurrency=args.concurrency)
    k64 = min(K, args.group)
    gp.run_batch(W, k64, ell[W:W + k64], sn[W:W + k64], concurrency=args.concurrency, group=args.group)
    sync(); ta = time.perf_counter()
    r64 = gp.run_batch(W, k64, ell[W:W + k64], sn[W:W + k64], concurrency=args.concurrency, group=args.group)
    sync(); tb64 = time.perf_counter() - ta
    assert np.all(r64["info"] == 0) and np.all(np.isfinite(r64["var"]))
    out["m64"] = {"value": k64 / tb64, "unit": "fits/s", "steps": k64, "note": "same workload with m=64 test points per fit (ride-along rows), measured after the timed region"}


def sharded_with_watchdog(out, args, rank, world, local, dist, backend, emit):
    """ONE fit sharded over all ranks (configs[3], configs[4]): untimed record.  A watchdog prints the metric line without it if a
    collective hangs -- nothing after the timed region may cost the line."""
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
            dg.gp.set_option("dist_panel_split", 0)    # the streamed whole-panel broadcast first (from four ranks on the row-split exchange is the default: it is timed below)
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
                dg.gp.set_option("dist_panel_split", 1)          # (with the next panel's first update divided by rows: dist_lookahead2d, default)
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
        # the model beside the measurement: with the row-split exchange and the next panel's first update divided by rows, what is
        # left on the critical path of a panel is the owner's top-block chain + its hot rows (update, solve, one broadcast of W x W 128-blocks) + the next
        # owner's top-block update; piece solves, all-gather and the rest of the update run beside the next chain.  Inputs: the owner-only device time
        # per panel MEASURED in this run (max over ranks), 150 GB/s per xGMI link, the trailing-update rate of one GPU for the top-block update.
        npan = max(1, -(-(n // 128) // outer))
        esz = 8 if dtype == "f64" else 4
        own_pp = float(vmax[7]) * world / npan                       # ms, row-split path (top chain + hot rows)
        hot_bytes = (outer * 128) * (outer * 128) * esz
        top_upd_ms = 1e3 * (outer * (outer + 1) / 2) * 2 * 128 * 128 * (outer * 128) / (60e12 if dtype == "f64" else 110e12)
        e["critical_path_model"] = {"per_panel_ms": {"owner_top_chain_and_hot_rows_measured": own_pp, "hot_rows_broadcast_at_150GBps": 1e3 * hot_bytes / 150e9,
                                                     "next_owners_top_block_update": top_upd_ms},
                                    "panels": npan,
                                    "ms_per_fit_if_everything_else_hides": npan * (own_pp + 1e3 * hot_bytes / 150e9 + top_upd_ms),
                                    "note": "a MODEL: valid when piece solves + all-gather + the rest of the trailing update (1/world of it per rank) fit beside the next owner's chain; "
                                            "measured ms_per_fit of this run is in row_split_exchange.ms_per_fit (ranks sharing one GPU over gloo measure correctness only)"}
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
    try:
        rec["reference_kernel_mlii"] = reference_kernel_mlii(local)
    except Exception as e:                       # noqa: BLE001
        rec["reference_kernel_mlii"] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    ml["note"] = "one MLII evaluation = fit (n^3/3) + L~^-T by recursive triangular inversion (n^3/3) + lower K~^-1 = U U^T (n^3/3) + O(n^2 d) derivative/reductions"
    rec["mlii"] = ml
    return rec


def reference_kernel_grid(local):
    """The reference's OWN kernel at the reference's OWN size: the retro loop's 3 regions x 40 years
    (n = 6 .. 45 training years, N = 60 / 20 / 12 network areas) x the 20 x 20 grid of north/June1st.py:210-211 = 48 000 fits
    in ONE launch (one workgroup per fit), beside the oracle's loop over a sample of the same fits on the host."""
    from seaiceextentforecasting_amd import GPR, SmallBatch, LGRID, SGRID
    sets = reference_kernel_sets()
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


def reference_kernel_sets():
    """The retro loop's shapes: 3 regions (N = 60 / 20 / 12 network areas) x 40 years (n = 6 .. 45 training years)."""
    rng = np.random.default_rng(20240010)
    sets = []
    for N in (60, 20, 12):
        for t in range(40):
            n = 6 + t
            X = rng.standard_normal((n, N)) * (1.0 + 0.3 * rng.standard_normal(N))
            y = X @ rng.standard_normal(N) / np.sqrt(N) + 0.5 * rng.standard_normal(n)
            sets.append((X, y, rng.standard_normal((1, N))))
    return sets


def reference_kernel_mlii(local):
    """The reference's MLII closure (north/June1st.py:235-257) for its OWN kernel, batched: value + the reference's "gradient" + the exact
    gradient for 3 regions x 40 years x the 20 x 20 grid of thetas in ONE launch (sigp_small_run_grad), and the optimiser call the
    reference left commented out (:259-262) for all 120 (region, year) data sets in lockstep (GPR.optimize_batch, one launch per round)."""
    from seaiceextentforecasting_amd import GPR, SmallBatch, LGRID, SGRID
    sets = reference_kernel_sets()
    with GPR(kernel="netdiffusion", device=local) as gp:
        sb = SmallBatch(gp)
        for X, y, Xs in sets:
            ds = sb.add_dataset(X, y, None)
            for e in LGRID:
                for s_ in SGRID:
                    sb.add_fit(ds, e, s_, expm="eigh")
        sb.upload()
        r = sb.run(grad=True)
        gp.synchronize(); t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            r = sb.run(grad=True)
        gp.synchronize(); dt = (time.perf_counter() - t0) / reps
        F = len(r["nlml"])
        ok = r["info"] == 0
        out = {"evaluations": F, "ok": int(ok.sum()), "ms_per_launch": 1e3 * dt, "evaluations_per_s": F / dt,
               "finite_gradients": int(np.sum(np.all(np.isfinite(r["grad_exact"][ok]), axis=1))),
               "note": "one workgroup per (data set, theta): fit + L~^-1 in LDS + tr(K~^-1 dK~), a~^T dK~ a~ in factored form; includes the H2D copy of the theta list and the D2H copy of 8 doubles per evaluation"}
    with GPR(kernel="netdiffusion", device=local) as gp:
        x0 = np.tile(np.log([1e-2, 1.0]), (len(sets), 1))
        gp.synchronize(); t0 = time.perf_counter()
        ro = gp.optimize_batch([s[0] for s in sets], [s[1] for s in sets], x0, maxiter=40)
        gp.synchronize(); to = time.perf_counter() - t0
        out["optimise_120_region_years"] = {"seconds": to, "launches": int(ro["nfev"]), "converged": int(np.sum(ro["converged"])), "iterations_mean": float(np.mean(ro["nit"])),
                                            "nlml_mean_at_optimum": float(np.mean(ro["fun"][np.isfinite(ro["fun"])])),
                                            "note": "lockstep modified Newton (optim.newton_lockstep): every round = ONE launch carrying 4 step lengths x 3 points per unfinished data set; includes the host's eigh per data set and the upload"}
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
