#!/usr/bin/env python3
"""Bisect the flaky NaN of examples/train_sparse_ddp.py --graph-step (debug aid, GPU box)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
from lc_amd.config import AttrDict  # noqa: E402
from lc_amd.graphs import GraphedTrainStep  # noqa: E402
from lc_amd.losses import Loss_fn  # noqa: E402
from lc_amd.ptnet import sparse_head  # noqa: E402
from train_sparse_ddp import KeypointNet, synthetic_blob  # noqa: E402

flags = set(sys.argv[1:])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = KeypointNet(16, 16).to(dev).to(memory_format=torch.channels_last)
cfg = AttrDict(pose_loss_cfg=dict(type="cov", clip_weight_grad=True), pose_loss_start_step=4, pose_loss_start_epoch=0, w_loss_kpts=1, w_loss_pose=0.7)
loss_fn = Loss_fn(cfg, AttrDict()).to(dev)
if "noattach" not in flags:
    model.loss_fn = loss_fn
opt = torch.optim.Adam(model.parameters(), lr=1e-4, capturable=True)


def loss_of(inp, s=5):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        lg = model(inp["rgb_in"].contiguous(memory_format=torch.channels_last))
    o = sparse_head(lg)
    ld, wd = loss_fn(inp, o, 0, s, 100)
    aux = {} if "noaux" in flags else {"loss_kpts": ld["loss_kpts"], "loss_pose": ld["loss_pose"]}
    if "detachaux" in flags:
        aux = {k: v.detach() for k, v in aux.items()}
    return sum(wd.values()), aux


whole = None
out = []
for step in range(8):
    blob = synthetic_blob(4, 16, dev, seed=step)
    if "nosync" not in flags:
        torch.cuda.synchronize(dev)
    if whole is None:
        whole = GraphedTrainStep(loss_of, opt, blob)
    loss, aux = whole(blob)
    torch.cuda.synchronize(dev)
    out.append(f"{float(loss):.4g}")
print(sorted(flags), out)
