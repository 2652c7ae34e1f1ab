#!/usr/bin/env python3
"""Time ONE large fit sharded over the ranks of a node (BASELINE configs[3]: n=16384, d=16, fp64 RBF).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/dist_bench.py \
        [--n 16384] [--d 16] [--outer 8] [--reps 3] [--no-lookahead] [--owner-only]

Backend "nccl" (RCCL over xGMI) when every rank has its own GPU; SIGP_BENCH_BACKEND=gloo lets several ranks share one
GPU to rehearse the protocol (timings are then meaningless for scaling: the ranks time-share the card).
Prints one JSON line on rank 0.  Not the bench.py line: the driver's scaling runs use the year-sharded batch."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--d", type=int, default=16)
    ap.add_argument("--outer", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-lookahead", action="store_true")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"], help="f32: configs[4] (use --n 32768 --d 32 --kernel matern52 --sn 0.1)")
    ap.add_argument("--kernel", default="rbf", choices=["rbf", "matern52"])
    ap.add_argument("--sn", type=float, default=1e-2)
    ap.add_argument("--owner-only", action="store_true", help="owner-only storage (sigp_dist_local_*): per-rank matrix bytes ~ 1/world")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    backend = os.environ.get("SIGP_BENCH_BACKEND", "nccl")
    if world > 1 or "MASTER_ADDR" in os.environ:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    from seaiceextentforecasting_amd import DistributedGPR, GPR
    rng = np.random.default_rng(20240003)
    n, d = args.n, args.d
    X = rng.standard_normal((n, d)); w = rng.standard_normal(d) / np.sqrt(d)
    y = np.sin(X @ w) + 0.1 * rng.standard_normal(n); Xs = rng.standard_normal((1, d))
    if args.dtype == "f32":
        Xs = Xs[:1]
    ell, sn = np.sqrt(d), args.sn
    times = []
    with DistributedGPR(args.kernel, rank, world, dist, device=local, outer_blocks=args.outer, lookahead=not args.no_lookahead, dtype=args.dtype,
                        owner_only=args.owner_only) as dg:
        for r in range(args.reps + 1):
            if dist.is_initialized():
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dg.fit(X, y, ell, sn, Xs=Xs)
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
            if dist.is_initialized():
                tt = torch.tensor([t], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                t = float(tt.item())
            if r > 0:
                times.append(t)
        mu, var = dg.predict(Xs)
        sf, nl, mbytes = dg.sigma_f_, dg.nlml_, dg.matrix_bytes_
    single = None
    if rank == 0:      # the same fit through the single-GPU entry point, for reference
        with GPR(kernel=args.kernel, device=local, outer_blocks=args.outer, dtype=args.dtype) as gp:
            for r in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                gp.fit(X, y, ell, sn, Xs=Xs)
                torch.cuda.synchronize(); single = time.perf_counter() - t0
            mu1, var1 = gp.predict(Xs)
        flops = n ** 3 / 3
        best = min(times)
        print(json.dumps({"workload": "one fit n=%d d=%d %s %s sharded over %d rank(s), 1-D block-cyclic panels of %d x 128 columns" % (n, d, args.dtype, args.kernel, world, args.outer),
                          "backend": backend if world > 1 else "none", "lookahead": not args.no_lookahead, "ms_per_fit": [round(1e3 * t, 2) for t in times],
                          "fits_per_s": 1.0 / best, "tflops": flops / best / 1e12, "single_gpu_entry_ms": round(1e3 * single, 2),
                          "mean_rel_vs_single": float(abs(mu[0] - mu1[0]) / abs(mu1[0])), "var_rel_vs_single": float(abs(var[0] - var1[0]) / abs(var1[0])),
                          "sigma_f": sf, "nlml": nl, "owner_only": args.owner_only, "matrix_bytes_rank0": mbytes}))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
