"""What the two training examples share: process-group set-up (RCCL over xGMI by default; gloo, optionally with every rank on
GPU 0, so that the N-rank step can be exercised on a one-GPU box), the single-process twin of an N-rank run (the same
crops, concatenated) and the end-of-run dump the tests compare."""
from __future__ import annotations

import json
import os
import time

import torch

XGMI_LINK_GBPS = 153.0  # one xGMI link of an MI355X, per direction


def add_args(ap):
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL over xGMI (one GPU per rank); gloo: CPU-staged "
                    "collectives, works with --share-gpu")
    ap.add_argument("--share-gpu", action="store_true", help="every rank computes on GPU 0 (tests on a one-GPU box; needs --backend gloo)")
    ap.add_argument("--emulate-ranks", type=int, default=0, help="single process: build the crops of this many ranks and concatenate them "
                    "(the run an N-rank job must reproduce)")
    ap.add_argument("--bn-eval", action="store_true", help="batch-norm layers use their running statistics (per-rank batch statistics "
                    "are the one thing a sharded step cannot share without SyncBN; the parity test switches them off)")
    ap.add_argument("--seed-offset", type=int, default=0)
    ap.add_argument("--report-comm", action="store_true", help="per step and per rank: gradient all-reduce payload (bytes, buckets) and time, the "
                    "small all-reduces of the loss (NormClipper norms, code histogram), next to SURVEY.md section 5's xGMI estimates")
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


class CommReport:
    """What a sharded step sends (the gradient all-reduce; the NormClippers' squared norms; the code histogram's counts) and how
    long it takes, per step and per rank, so that the first run on an 8-GPU node yields SURVEY section 5's comparison without edits.

    * gradients: a DistributedDataParallel communication hook that does what the default one does (divide by the world size, all-reduce the
      bucket) and notes each bucket's bytes and its time from hand-over to completion -- HIP events on the RCCL path (the completion callback
      runs on a stream that has waited for the collective), the host clock on the gloo path (where the collective itself runs on the host).
      `window` = first hand-over to last completion: the span the collectives occupy beside the backward pass; the sum of the bucket times
      counts queueing behind earlier buckets twice and is an upper bound.
    * small all-reduces issued by `lc_amd` (`grad.py`, `losses.py`): `torch.distributed.all_reduce` is wrapped for the run; bytes, count and
      the host time of the call (latency-sized messages: 4 bytes per clipper, 8 (C + 1) bytes for the histogram).
    """

    def __init__(self, world, rank, group, backend):
        self.world, self.rank, self.group, self.backend = world, rank, group, backend
        self.steps, self.cur = [], None
        self._events = []

    def attach(self, ddp_net):
        import torch.distributed as dist

        real = dist.all_reduce
        rep = self

        def counted_all_reduce(tensor, *a, **kw):
            t0 = time.perf_counter()
            out = real(tensor, *a, **kw)
            if rep.cur is not None and not rep._in_hook:
                rep.cur["small_calls"] += 1
                rep.cur["small_bytes"] += tensor.numel() * tensor.element_size()
                rep.cur["small_host_ms"] += (time.perf_counter() - t0) * 1e3
            return out

        self._in_hook = False
        self._real_all_reduce = real
        dist.all_reduce = counted_all_reduce

        def hook(state, bucket):
            buf = bucket.buffer()
            rec = {"bytes": buf.numel() * buf.element_size()}
            if rep.backend == "nccl":
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            else:
                t0 = time.perf_counter()
            buf.div_(rep.world)
            rep._in_hook = True
            fut = real(buf, group=rep.group, async_op=True).get_future()
            rep._in_hook = False

            def done(f):
                if rep.backend == "nccl":
                    e1.record()
                    rec["events"] = (e0, e1)
                else:
                    rec["t"] = (t0, time.perf_counter())
                return f.value()[0]

            if rep.cur is not None:
                rep.cur["buckets"].append(rec)
            return fut.then(done)

        ddp_net.register_comm_hook(None, hook)

    def begin_step(self):
        self.cur = {"buckets": [], "small_calls": 0, "small_bytes": 0, "small_host_ms": 0.0}

    def end_step(self, step_ms):
        """Call after the step's synchronize."""
        c, self.cur = self.cur, None
        per, t_first, t_last = [], None, None
        for b in c["buckets"]:
            if "events" in b:
                per.append(b["events"][0].elapsed_time(b["events"][1]))
            elif "t" in b:
                per.append((b["t"][1] - b["t"][0]) * 1e3)
        window = 0.0
        if c["buckets"]:
            first, last = c["buckets"][0], c["buckets"][-1]
            if "events" in first and "events" in last:
                window = first["events"][0].elapsed_time(last["events"][1])
            elif "t" in first and "t" in last:
                window = (last["t"][1] - first["t"][0]) * 1e3
        self.steps.append({"grad_bytes": sum(b["bytes"] for b in c["buckets"]), "buckets": len(c["buckets"]), "bucket_ms": [round(v, 4) for v in per],
                           "grad_window_ms": round(window, 4), "small_calls": c["small_calls"], "small_bytes": c["small_bytes"],
                           "small_host_ms": round(c["small_host_ms"], 4), "step_ms": round(step_ms, 4)})

    @staticmethod
    def xgmi_estimate_ms(payload_bytes, world):
        """a ring all-reduce is bound by ONE link (2 (w-1)/w of the payload over it); reduce-scatter + all-gather sent
        directly to the w-1 peers over their own links moves 2 / w of the payload per link."""
        gb = payload_bytes / 1e9
        return {"ring_ms": 2 * (world - 1) / world * gb / XGMI_LINK_GBPS * 1e3, "direct_ms": 2 / world * gb / XGMI_LINK_GBPS * 1e3}

    def finish(self, out_path=None):
        """All ranks' step records to rank 0; it prints one line per step and rank, the medians, and the estimate for the measured payload."""
        import torch.distributed as dist

        real = getattr(self, "_real_all_reduce", None)
        if real is not None:  # the wrapper lives for the run only
            dist.all_reduce, self._real_all_reduce = real, None
        gathered = [None] * self.world
        dist.all_gather_object(gathered, self.steps, group=self.group)
        if self.rank != 0 or not self.steps:
            return None
        print("comm report (per step, per rank): gradient all-reduce payload / buckets / window, loss-side small all-reduces")
        for s in range(len(self.steps)):
            for r, steps in enumerate(gathered):
                d = steps[s]
                print(f"  step {s:3d} rank {r}: grads {d['grad_bytes'] / 1e6:8.2f} MB in {d['buckets']} bucket(s), window {d['grad_window_ms']:7.3f} ms "
                      f"(buckets {' '.join('%.3f' % v for v in d['bucket_ms'])}); small all-reduces {d['small_calls']} x, {d['small_bytes']} B, "
                      f"{d['small_host_ms']:.3f} ms host; step {d['step_ms']:.1f} ms")
        tail = [g[2:] if len(g) > 3 else g for g in gathered]
        med = lambda vals: sorted(vals)[len(vals) // 2]  # noqa: E731
        payload = med([d["grad_bytes"] for g in tail for d in g])
        summary = {"world": self.world, "backend": self.backend, "grad_payload_bytes": payload, "buckets": med([d["buckets"] for g in tail for d in g]),
                   "grad_window_ms_median": med([d["grad_window_ms"] for g in tail for d in g]),
                   "small_allreduces_per_step": med([d["small_calls"] for g in tail for d in g]),
                   "small_bytes_per_step": med([d["small_bytes"] for g in tail for d in g]),
                   "small_host_ms_median": med([d["small_host_ms"] for g in tail for d in g]),
                   "step_ms_median": med([d["step_ms"] for g in tail for d in g]),
                   "xgmi_estimate": self.xgmi_estimate_ms(payload, self.world),
                   "survey_estimate_104MB_8gpu": self.xgmi_estimate_ms(104e6, 8)}
        summary["grad_window_share_of_step"] = round(summary["grad_window_ms_median"] / max(summary["step_ms_median"], 1e-9), 4)
        print("comm summary " + json.dumps(summary))
        if out_path:
            with open(out_path, "w") as f:
                json.dump({"summary": summary, "per_rank_steps": gathered}, f)
        return summary
