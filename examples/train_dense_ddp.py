#!/usr/bin/env python3
"""Training step of the DENSE-correspondence heads around the HIP hot path (BASELINE configs[2] / [4], synthetic data):

    backbone, fp16 / bf16 autocast: `--trunk cdpn` (the encoder/decoder of train_sparse_ddp.py, 64x64 maps: glmo / gycbv) or
             `--trunk os8` (examples/os8_trunk.py: dilated output-stride-8 encoder + atrous pyramid + skip decoder, 128x128 maps: zlmo / zycbv)
      -> (B, C, S, S) maps: xyz_noc (3) | xyz_noc_bin (sum of code bits), xyz_weights (2), msk_vis (1) + a per-sample weight scale
      -> lc_amd.losses.Loss_fn dense branch: joint-softmax front end (HIP), [ZebraPose code decode (HIP)], LC loss at
         N = ceil(S / sample)^2 (HIP), NormClipper hooks (HIP, whole-batch norm all-reduced over RCCL when sharded), L1 / code / segmentation terms
      -> backward -> DistributedDataParallel gradient all-reduce -> Adam

    python examples/train_dense_ddp.py --steps 10 [--bin]                 # glmo's shape: 64x64 maps, stride 2, N = 1024
    python examples/train_dense_ddp.py --steps 10 --zlmo                  # configs/zlmo.yaml's own shape: B=32, OS8 trunk, 128x128 maps, 21 code
                                                                          # planes (max_bit_cnt 7), dense_sample 3 => N = 1849, fp16 + GradScaler,
                                                                          # loss block of zlmo.yaml:74-83
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_dense_ddp.py --steps 10 [--zlmo] [--report-comm]

Targets are geometrically consistent (a synthetic surface seen from a known pose), weights are random-init: the point is the
plumbing and the step time, not a trained model.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lc_amd import floatbits, synth  # noqa: E402
from lc_amd.config import AttrDict  # noqa: E402
from lc_amd.losses import Loss_fn  # noqa: E402
from lc_amd.transforms import gen_uv, quaternion_rep_to_RT  # noqa: E402
from train_sparse_ddp import KeypointNet  # noqa: E402
from os8_trunk import OS8Trunk  # noqa: E402
import ddp_common  # noqa: E402

BITS = (7, 7, 6)  # default bit budget of the small cases; --zlmo: 7,7,7 (configs/zlmo.yaml:8 max_bit_cnt 7 per axis)


class DenseNet(nn.Module):
    """ptnet.py:20-38,68-80 structure: one trunk, channel slices per head, weight scale = exp(Linear(mean feature))."""

    def __init__(self, noc_channels, width=64, trunk="cdpn"):
        super().__init__()
        self.os8 = trunk == "os8"
        self.trunk = OS8Trunk(noc_channels + 3, width) if self.os8 else KeypointNet(sparse_cnt=noc_channels + 3, width=width)
        self.noc_channels = noc_channels
        self.weight_scale_layer = nn.Linear(self.trunk.feature_dim if self.os8 else 256, 1)
        nn.init.zeros_(self.weight_scale_layer.weight)
        nn.init.constant_(self.weight_scale_layer.bias, 3.0)

    def forward(self, rgb):
        t = self.trunk
        if self.os8:
            raw, feature = t(rgb)
        else:
            feature = t.decoder(t.encoder(t.stem(rgb)))
            raw = t.head(feature)
        c = self.noc_channels
        scale = self.weight_scale_layer(feature.flatten(start_dim=-2).mean(dim=-1).float()).exp()[..., None, None]
        return raw[:, :c], raw[:, c:c + 2], scale, raw[:, c + 2:c + 3]


def synthetic_blob(B, dev, seed, binary, S=64, bits=BITS):
    """Blob of the reference's dense shape (dataset.py:451-489): a noisy planar patch at ~500 mm seen through an SxS output grid."""
    g = torch.Generator().manual_seed(seed)
    b = synth.make_batch(B, 4, seed=seed + 7, rotate_K=False)
    K = b["K"].clone()
    K[:, 0, 0] = K[:, 1, 1] = 440.0 * S / 64
    K[:, 0, 2] = K[:, 1, 2] = S / 2
    pose = b["pose"].clone()
    pose[:, 4:6] = 0
    pose[:, 6] = 500.0
    R, t = quaternion_rep_to_RT(pose)
    rays = torch.cat((gen_uv((S, S)), torch.ones(S, S, 1)), -1).reshape(1, -1, 3) @ torch.linalg.inv(K).mT
    Xm = (rays * (500.0 + 10 * torch.randn(B, S * S, 1, generator=g)) - t[:, None]) @ R
    noc_scale = torch.tensor(synth.EXTENT_MM).expand(B, 3).contiguous()
    noc = (Xm / noc_scale[:, None]).mT.reshape(B, 3, S, S).clamp(-0.999, 0.999)
    msk = (torch.rand(B, S, S, generator=g) > 0.3)
    blob = dict(rgb_in=torch.rand(B, 3, 256, 256, generator=g), pose_best=pose, out_K=K, bbox_3d=b["bbox_3d"], noc_scale=noc_scale,
                msk_noc=msk, msk_vis=msk.float())
    if binary:
        mod_bits, raw_bits = floatbits.nn_noc2target(noc.permute(0, 2, 3, 1), list(bits))
        blob.update(xyz_noc_bin_tgt=mod_bits, xyz_noc_bin_raw=raw_bits, bit_cnt=list(bits))
    else:
        blob["xyz_noc_tgt"] = noc * msk[:, None]
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in blob.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--bin", action="store_true", help="ZebraPose binary-code head (zlmo/zycbv) instead of the continuous xyz head")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16", "fp32"])
    ap.add_argument("--trunk", default="cdpn", choices=["cdpn", "os8"], help="cdpn: stride-4 encoder/decoder, 64x64 maps (glmo / gycbv); os8: dilated "
                    "output-stride-8 encoder + atrous pyramid + skip decoder, 128x128 maps (zlmo / zycbv: model/zebra_DeepLabV3.py)")
    ap.add_argument("--sample", type=int, default=2, help="loss.pose_loss_cfg.dense_sample: stride of the correspondence sub-sampling")
    ap.add_argument("--bits", default=None, help="code bits per axis, e.g. 7,7,7 (implies --bin)")
    ap.add_argument("--zlmo", action="store_true", help="configs/zlmo.yaml's own shape and loss block: --bin --bits 7,7,7 --trunk os8 --sample 3, "
                    "w_loss_pose 0.03 / w_loss_noc_bin 3 / w_loss_seg 1 (L1), clip_weight_grad only, fp16 autocast unless --dtype says otherwise")
    ap.add_argument("--graphs", action="store_true", help="replay the Loss_fn step as hipGraphs (one per sub-sampling phase; eager inside the warm-up ramp)")
    ap.add_argument("--np-seed", type=int, default=0, help="seed of the sub-sampling phase draws (losses.py:152), the SAME on every rank: the reference's one "
                    "process draws one (top, left) phase per step for the whole batch, so the ranks of a sharded step must draw the same one")
    ddp_common.add_args(ap)
    args = ap.parse_args()
    if args.zlmo:
        args.bin, args.trunk, args.sample, args.bits = True, "os8", 3, args.bits or "7,7,7"
    bits = tuple(int(v) for v in args.bits.split(",")) if args.bits else BITS
    args.bin = args.bin or bool(args.bits)
    S = 128 if args.trunk == "os8" else 64
    world, rank, dev, group = ddp_common.init(args)
    local = dev.index
    torch.manual_seed(0)
    np.random.seed(args.np_seed)
    model = DenseNet(sum(bits) if args.bin else 3, args.width, args.trunk).to(dev).to(memory_format=torch.channels_last)
    if args.zlmo:  # configs/zlmo.yaml:74-83 (the ramp shortened from 3000 steps to 4 so that a short run crosses it)
        cfg = AttrDict(pose_loss_cfg=dict(dense_sample=3, clip_weight_grad=True), seg_loss_type="L1", pose_loss_start_step=4, pose_loss_start_epoch=0,
                       w_loss_pose=0.03, w_loss_noc_bin=3, w_loss_seg=1)
    else:
        cfg = AttrDict(pose_loss_cfg=dict(type="cov", clip_weight_grad=True, clip_scale_grad=True, clip_pts_grad=not args.bin, dense_sample=args.sample,
                                          max_err_len=32), pose_loss_start_step=4, pose_loss_start_epoch=0, loss_pose_nz_step=0,
                       w_loss_seg=1, w_loss_pose=0.05, seg_loss_type="L1", **({"w_loss_noc_bin": 1} if args.bin else {"w_loss_noc": 1}))
    # group + 1 / world: the NormClipper norms are those of the whole batch's mean loss (lc_amd/grad.py), as in the reference's single process
    loss_fn = Loss_fn(cfg, AttrDict(), sum(bits) if args.bin else 0, group=group, shard_loss_scale=1.0 / world).to(dev)
    model.loss_fn = loss_fn
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
    amp = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": None}[args.dtype]
    scaler = torch.amp.GradScaler("cuda", enabled=amp is torch.float16)
    times, losses, clip_states = [], [], []
    params_at_start = ddp_common.flat_params(model) if args.dump else None
    graphed = None
    for step in range(args.steps):
        blob = ddp_common.cat_blobs([synthetic_blob(args.batch, dev, seed=args.seed_offset + 1000 * r + step, binary=args.bin, S=S, bits=bits)
                                     for r in ddp_common.data_ranks(args, world, rank)])
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        if report is not None:
            report.begin_step()
        with torch.autocast("cuda", dtype=amp or torch.float16, enabled=amp is not None):
            noc, wlogits, wscale, vis = net(blob["rgb_in"].contiguous(memory_format=torch.channels_last))
        # the heads' maps go to the loss in the autocast type: the kernels read fp16 / bf16 natively, compute in fp32 and write the maps'
        # gradients in the maps' type (lc_amd/_lib.py: hip_maps) -- no up-cast copy at the boundary
        out = {"xyz_noc_bin" if args.bin else "xyz_noc": noc, "xyz_weight_logits": wlogits, "xyz_weights_scale": wscale, "msk_vis_logits": vis}
        if args.graphs:
            if graphed is None:
                from lc_amd.graphs import GraphedLoss
                labels = {k: v for k, v in blob.items() if k != "rgb_in"}  # the crops are no input of the loss: keep them out of the graph's static inputs
                graphed = GraphedLoss(loss_fn, labels, out, 0, step, 100)
            loss_dict, w_loss_dict = graphed({k: v for k, v in blob.items() if k != "rgb_in"}, out, step=step)  # the guard runs the ramp steps eagerly, captures at the plateau
        else:
            loss_dict, w_loss_dict = loss_fn(blob, out, 0, step, 100)
        loss = sum(w_loss_dict.values())
        opt.zero_grad(set_to_none=True)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        torch.cuda.synchronize(dev)
        times.append(time.perf_counter() - t0)
        if report is not None:
            report.end_step(times[-1] * 1e3)
        losses.append(float(loss))
        clip_states.append({k: float(v) for k, v in loss_fn.state_dict().items() if k.endswith("max_norm")})
        if rank == 0:
            terms = "  ".join(f"{k[5:]} {float(v):8.4f}" for k, v in loss_dict.items())
            print(f"step {step:3d}  loss {float(loss):9.4f}  {terms}  {times[-1] * 1e3:7.1f} ms")
        assert torch.isfinite(loss), "non-finite loss"
    if rank == 0 and len(times) > 3:
        tail = sorted(times[2:])
        t = tail[len(tail) // 2]
        q = lambda f: tail[min(len(tail) - 1, int(f * len(tail)))] * 1e3  # noqa: E731
        # (round 4's summaries showed p75 = 74 ms next to a 16 ms median: the host's 128-thread OpenMP pool, woken by the per-step CPU blob
        # construction, spinning beside the launch thread -- ddp_common.init now caps it; profiles/r05/step_stall.txt has the A/B)
        print(f"quartiles of the step time [ms]: min {q(0):.1f}  p25 {q(0.25):.1f}  median {q(0.5):.1f}  p75 {q(0.75):.1f}  max {q(1):.1f}")
        print(f"median step {t * 1e3:.1f} ms -> {args.batch * world / t:.0f} crops/s on {world} GPU(s), {args.dtype} backbone, "
              f"{args.trunk} trunk, {S}x{S} maps, {'binary-code (%d planes)' % sum(bits) if args.bin else 'continuous-xyz'} dense head, "
              f"N={(-(-S // args.sample)) ** 2} correspondences per sample")
    ddp_common.dump(args, rank, model, loss_fn, losses, clip_states, params_at_start)
    if report is not None:
        report.finish(f"{args.dump}.comm.json" if args.dump else None)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
