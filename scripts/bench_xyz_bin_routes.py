#!/usr/bin/env python3
"""`Loss_xyz_bin` forward + backward per rank at zlmo's training shape (B=32 per rank, 21 code planes, 128x128 maps), three routes:

  one-launch   the single-process kernel (no group)
  split        counts launch -> int64 all-reduce -> finish launch (what a sharded job runs since round 6)
  torch        the reference's torch formulas around a float all-reduce (what a sharded job ran up to round 5; LC_AMD_XYZ_BIN_TORCH=1)

Two ranks share GPU 0 over gloo (the rank count does not change a rank's work); the parent process never touches the GPU.
    python scripts/bench_xyz_bin_routes.py [--dtype f16] [--batch 32]
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, args, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if args.route == "torch":
        os.environ["LC_AMD_XYZ_BIN_TORCH"] = "1"
    import torch
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lc_amd.losses import Loss_xyz_bin

    dev = torch.device("cuda:0")
    dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[args.dtype]
    g = torch.Generator().manual_seed(rank)
    B, C, S = args.batch, 21, 128
    logits = (torch.randn(B, C, S, S, generator=g) * 2).to(dt).to(dev).requires_grad_(True)
    bits = (torch.rand(B, C, S, S, generator=g) < 0.5).to(dev)
    vis = torch.randn(B, 1, S, S, generator=g).to(dt).to(dev)
    fn = Loss_xyz_bin(C, group=None if args.route == "one-launch" else dist.group.WORLD).to(dev)

    def step():
        logits.grad = None
        fn(logits, bits, vis).backward()

    for _ in range(10):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    times = []
    for _ in range(args.iters):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e6)
    times.sort()
    ret[rank] = {"median_us": times[len(times) // 2], "p10_us": times[len(times) // 10], "p90_us": times[9 * len(times) // 10]}
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--world", type=int, default=2)
    args = ap.parse_args()
    import torch.multiprocessing as mp

    out = {"shape": [args.batch, 21, 128, 128], "dtype": args.dtype, "world": args.world, "what": "fwd+bwd of Loss_xyz_bin per rank, host clock around a synchronised step"}
    for route in ("one-launch", "split", "torch"):
        args.route = route
        ctx = mp.get_context("spawn")
        ret = ctx.Manager().dict()
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = [ctx.Process(target=worker, args=(r, args.world, port, args, ret)) for r in range(args.world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(600)
        out[route] = dict(ret[0])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
