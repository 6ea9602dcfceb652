#!/usr/bin/env python3
"""Which ingredient of the whole-step graph goes non-finite on replay?  (debug aid, GPU box)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
from lc_amd.config import AttrDict  # noqa: E402
from lc_amd.graphs import GraphedTrainStep  # noqa: E402
from lc_amd.losses import Loss_fn  # noqa: E402
from lc_amd.ptnet import sparse_head  # noqa: E402
from train_sparse_ddp import KeypointNet, synthetic_blob  # noqa: E402

dev = torch.device("cuda:0")


def run(name, opt_kind, autocast, loss_kind, bn_eval=False):
    torch.manual_seed(0)
    model = KeypointNet(16, 16).to(dev).to(memory_format=torch.channels_last)
    if bn_eval:
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eval()
    cfg = AttrDict(pose_loss_cfg=dict(type="cov", clip_weight_grad=True), pose_loss_start_step=4, pose_loss_start_epoch=0, w_loss_kpts=1, w_loss_pose=0.7)
    loss_fn = Loss_fn(cfg, AttrDict()).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True) if opt_kind == "adam" else torch.optim.SGD(model.parameters(), lr=1e-4)

    def loss_of(inp):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            lg = model(inp["rgb_in"].contiguous(memory_format=torch.channels_last))
        if loss_kind == "square":
            return lg.float().pow(2).mean(), {}
        o = sparse_head(lg)
        if loss_kind == "head":
            return (o["pts2d"].pow(2).mean() + o["pts2d_std"].pow(2).mean()), {}
        ld, wd = loss_fn(inp, o, 0, 5, 100)
        if loss_kind == "kpts":
            return ld["loss_kpts"], {}
        return sum(wd.values()), {}

    step = GraphedTrainStep(loss_of, opt, synthetic_blob(4, 16, dev, 0))
    out = []
    for i in range(8):
        loss, _ = step(synthetic_blob(4, 16, dev, i))
        torch.cuda.synchronize()
        out.append(f"{float(loss):.4g}")
    print(f"{name:40s} {out}", flush=True)


for rep in range(4):
    run("adam  autocast full-loss", "adam", True, "full")
for rep in range(3):
    run("adam  autocast square", "adam", True, "square")
for rep in range(3):
    run("adam  autocast head-only", "adam", True, "head")
for rep in range(3):
    run("adam  autocast kpts-only", "adam", True, "kpts")
for rep in range(3):
    run("adam  fp32 full-loss", "adam", False, "full")
