"""One large fit sharded over ranks (BASELINE configs[3] / [4]) through the library's own sharded fit -- the record bench.py
appends as "sharded" when N > 1, on its own:
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/dist_bench.py [--configs configs[3],configs[4]]
With one GPU: N = 1 (no transport), or SIGP_BENCH_BACKEND=gloo and N ranks sharing the GPU (host-pointer transport)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="configs[3],configs[4]")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--outer", type=int, default=8)
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    import bench_extras
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    backend = os.environ.get("SIGP_BENCH_BACKEND", "nccl")
    d = None
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local)) if backend == "nccl" else dist.init_process_group(backend)
        d = dist
    rec = bench_extras.sharded_record(rank, world, local, d, backend, configs=tuple(args.configs.split(",")), reps=args.reps, outer=args.outer)
    if rank == 0:
        print(json.dumps(rec))
    if d is not None:
        d.barrier(); d.destroy_process_group()


if __name__ == "__main__":
    main()
