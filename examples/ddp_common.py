"""What the two training examples share: process-group set-up (RCCL over xGMI by default; gloo, optionally with every rank on
GPU 0, so that the N-rank step can be exercised on a one-GPU box), the single-process twin of an N-rank run (the same
crops, concatenated) and the end-of-run dump the tests compare."""
from __future__ import annotations

import os

import torch


def add_args(ap):
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL over xGMI (one GPU per rank); gloo: CPU-staged "
                    "collectives, works with --share-gpu")
    ap.add_argument("--share-gpu", action="store_true", help="every rank computes on GPU 0 (tests on a one-GPU box; needs --backend gloo)")
    ap.add_argument("--emulate-ranks", type=int, default=0, help="single process: build the crops of this many ranks and concatenate them "
                    "(the run an N-rank job must reproduce)")
    ap.add_argument("--bn-eval", action="store_true", help="batch-norm layers use their running statistics (per-rank batch statistics "
                    "are the one thing a sharded step cannot share without SyncBN; the parity test switches them off)")
    ap.add_argument("--seed-offset", type=int, default=0)
    ap.add_argument("--dump", default=None, help="write losses, NormClipper states and the flattened parameters of every rank to <dump>.rank<r>.pt")


def init(args):
    """-> (world, rank, device, group or None)."""
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    local = 0 if args.share_gpu else local
    # The synthetic blobs are built by torch CPU ops every step.  With torch's default of one OpenMP thread per physical core (128 on the
    # GPU boxes) the pool those ops wake keeps spinning after them and competes with the ONE thread that launches the step's kernels:
    # a quarter of the dense steps then took 75 ms instead of 16 (profiles/r04/g1: p75 74.5 ms; the stall sits in the forward or the
    # backward, never in the lc_* launches; profiles/r05/step_stall.txt: 9 of 37 steps slow at 128 threads, 0 of 37 at 4).  A real
    # data loader runs in worker processes; here the host side is told to stay small.
    torch.set_num_threads(max(1, min(4, torch.get_num_threads())))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    group = None
    if world > 1:
        import torch.distributed as dist

        if args.share_gpu and args.backend == "nccl":
            raise SystemExit("--share-gpu needs --backend gloo (RCCL refuses two ranks per device)")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
        group = dist.group.WORLD
    return world, rank, dev, group


def data_ranks(args, world, rank):
    """The ranks whose crops this process trains on: its own, or all of an emulated job's."""
    return list(range(args.emulate_ranks)) if args.emulate_ranks else [rank]


def cat_blobs(blobs):
    if len(blobs) == 1:
        return blobs[0]
    return {k: (torch.cat([b[k] for b in blobs]) if isinstance(v, torch.Tensor) else v) for k, v in blobs[0].items()}


def freeze_bn(model):
    for m in model.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()


def flat_params(model):
    return torch.cat([p.detach().float().reshape(-1).cpu() for p in model.parameters()])


def dump(args, rank, model, loss_fn, losses, clip_states, params_at_start=None):
    if not args.dump:
        return
    flat = flat_params(model)
    torch.save({"params": flat, "params_at_start": params_at_start, "losses": losses, "clip_states": clip_states,
                "final_clip": {k: float(v) for k, v in loss_fn.state_dict().items() if k.endswith("max_norm")},
                "loss_state": {k: v.detach().cpu() for k, v in loss_fn.state_dict().items()}}, f"{args.dump}.rank{rank}.pt")
