#!/usr/bin/env python3
"""End-to-end training step around the HIP hot path (BASELINE configs[2]/[3] plumbing, synthetic data):

    backbone (ResNet-34-style encoder + 3-stage up-sampling decoder, plain PyTorch-ROCm: MIOpen/hipBLASLt own the conv GEMMs)
      -> keypoint logits (B,S,64,64), bf16 under autocast -> lc_amd.ptnet.sparse_head (fused spatial softmax + soft-argmax, HIP,
         consumes the 16-bit maps natively)
      -> lc_amd.losses.Loss_fn (Laplace keypoint NLL + LC loss via the fused HIP kernel, warm-up blend)
      -> backward -> gradient all-reduce over RCCL (DistributedDataParallel) -> Adam step

    python examples/train_sparse_ddp.py --steps 20                       # one GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_sparse_ddp.py --steps 20

The backbone is NOT part of the hot path this repo re-implements; it is here so that the drop-in
surface can be exercised exactly as `train.py:23-80` uses it.  Random-init weights, synthetic crops and poses.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lc_amd import synth  # noqa: E402
from lc_amd.config import AttrDict  # noqa: E402
from lc_amd.losses import Loss_fn  # noqa: E402
from lc_amd.ptnet import sparse_head  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ddp_common  # noqa: E402


class Block(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.c1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.b1 = nn.BatchNorm2d(cout)
        self.c2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.b2 = nn.BatchNorm2d(cout)
        self.down = None
        if stride != 1 or cin != cout:
            self.down = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        y = F.relu(self.b1(self.c1(x)))
        y = self.b2(self.c2(y))
        return F.relu(y + (x if self.down is None else self.down(x)))


class KeypointNet(nn.Module):
    """34-layer residual encoder ([3,4,6,3] basic blocks) + three x2 up-sampling stages: 256x256 crop -> (S,64,64) logits."""

    def __init__(self, sparse_cnt=64, width=64, layers=(3, 4, 6, 3)):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(3, width, 7, 2, 3, bias=False), nn.BatchNorm2d(width), nn.ReLU(inplace=True),
                                  nn.MaxPool2d(3, 2, 1))
        chans = [width, width * 2, width * 4, width * 8]
        blocks, cin = [], width
        for i, (c, n) in enumerate(zip(chans, layers)):
            for j in range(n):
                blocks.append(Block(cin, c, 2 if (j == 0 and i > 0) else 1))
                cin = c
        self.encoder = nn.Sequential(*blocks)
        ups, c = [], cin
        for _ in range(3):
            ups += [nn.ConvTranspose2d(c, 256, 3, 2, 1, output_padding=1, bias=False), nn.BatchNorm2d(256), nn.ReLU(inplace=True),
                    nn.Conv2d(256, 256, 3, 1, 1, bias=False), nn.BatchNorm2d(256), nn.ReLU(inplace=True)]
            c = 256
        self.decoder = nn.Sequential(*ups)
        self.head = nn.Conv2d(256, sparse_cnt, 1)

    def forward(self, rgb):
        return self.head(self.decoder(self.encoder(self.stem(rgb))))  # (B,S,64,64) for a 256x256 input


def synthetic_blob(B, S, dev, seed):
    """A batch of the reference's blob shape (dataset.py:451-489) with random crops and consistent pose labels."""
    b = synth.make_batch(B, S, seed=seed)
    g = torch.Generator().manual_seed(seed)
    blob = dict(rgb_in=torch.rand(B, 3, 256, 256, generator=g), pose_best=b["pose"], out_K=b["K"], pts3d=b["pts3d"], bbox_3d=b["bbox_3d"],
                msk_noc=torch.ones(B, 64, 64, dtype=torch.bool), msk_vis=torch.ones(B, 64, 64))
    return {k: v.to(dev) for k, v in blob.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (configs/gsplmo.yaml: 32)")
    ap.add_argument("--sparse-cnt", type=int, default=64)
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--bf16", action="store_true", default=True)
    ap.add_argument("--fp32", dest="bf16", action="store_false", help="no autocast (the two-rank parity test: bf16 round-off differs with the batch size)")
    ap.add_argument("--graphs", action="store_true", help="replay the Loss_fn step as hipGraphs once the warm-up ramp is over")
    ddp_common.add_args(ap)
    args = ap.parse_args()
    world, rank, dev, group = ddp_common.init(args)
    local = dev.index
    torch.manual_seed(0)
    model = KeypointNet(args.sparse_cnt, args.width).to(dev).to(memory_format=torch.channels_last)
    cfg = AttrDict(pose_loss_cfg=dict(type="cov", clip_weight_grad=True), pose_loss_start_step=4, pose_loss_start_epoch=0,
                   w_loss_kpts=1, w_loss_pose=0.7)
    # group + 1 / world: the NormClipper norms are those of the whole batch's mean loss (lc_amd/grad.py), as in the reference's single process
    loss_fn = Loss_fn(cfg, AttrDict(), group=group, shard_loss_scale=1.0 / world).to(dev)
    model.loss_fn = loss_fn  # train.py:31: rides in the checkpoint
    net = model
    if world > 1:
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local], bucket_cap_mb=64, gradient_as_bucket_view=True)
    report = None
    if args.report_comm and world > 1:
        report = ddp_common.CommReport(world, rank, group, args.backend)
        report.attach(net)
    if args.bn_eval:
        ddp_common.freeze_bn(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    times, losses, clip_states = [], [], []
    params_at_start = ddp_common.flat_params(model) if args.dump else None
    graphed = None
    for step in range(args.steps):
        blob = ddp_common.cat_blobs([synthetic_blob(args.batch, args.sparse_cnt, dev, seed=args.seed_offset + 1000 * r + step)
                                     for r in ddp_common.data_ranks(args, world, rank)])
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        if report is not None:
            report.begin_step()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.bf16):
            logits = net(blob["rgb_in"].contiguous(memory_format=torch.channels_last))
        out = sparse_head(logits)  # fused HIP head (ptnet.py:59-66) on the bf16 logits as they are: fp32 statistics, bf16 gradient
        if args.graphs and step > cfg.pose_loss_start_step:  # the blending factor is 1 from here on: the step is static
            if graphed is None:
                from lc_amd.graphs import GraphedLoss
                labels = {k: v for k, v in blob.items() if k != "rgb_in"}  # the crops are no input of the loss: keep them out of the graph's static inputs
                graphed = GraphedLoss(loss_fn, labels, out, 0, step, 100)
            loss_dict, w_loss_dict = graphed({k: v for k, v in blob.items() if k != "rgb_in"}, out)
        else:
            loss_dict, w_loss_dict = loss_fn(blob, out, 0, step, 100)
        loss = sum(w_loss_dict.values())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        torch.cuda.synchronize(dev)
        times.append(time.perf_counter() - t0)
        if report is not None:
            report.end_step(times[-1] * 1e3)
        losses.append(float(loss))
        clip_states.append({k: float(v) for k, v in loss_fn.state_dict().items() if k.endswith("max_norm")})
        if rank == 0:
            print(f"step {step:3d}  loss {float(loss):9.4f}  kpts {float(loss_dict['loss_kpts']):8.4f}  pose {float(loss_dict['loss_pose']):8.4f}"
                  f"  {times[-1] * 1e3:7.1f} ms")
    if rank == 0 and len(times) > 3:
        tail = sorted(times[2:])
        t = tail[len(tail) // 2]
        q = lambda f: tail[min(len(tail) - 1, int(f * len(tail)))] * 1e3  # noqa: E731
        # (on the shared pool a fraction of the steps carries a ~55-70 ms stall that is also there when the step is a single graph
        # replay with no host work in it; the quartiles show both modes)
        print(f"quartiles of the step time [ms]: min {q(0):.1f}  p25 {q(0.25):.1f}  median {q(0.5):.1f}  p75 {q(0.75):.1f}  max {q(1):.1f}")
        print(f"median step {t * 1e3:.1f} ms -> {args.batch * world / t:.0f} crops/s on {world} GPU(s), bf16 backbone, fp32/fp64 LC loss")
    ddp_common.dump(args, rank, model, loss_fn, losses, clip_states, params_at_start)
    if report is not None:
        report.finish(f"{args.dump}.comm.json" if args.dump else None)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
