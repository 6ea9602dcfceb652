#!/usr/bin/env python3
"""Where do the ~60 ms stalls of some dense example steps come from (profiles/r04/g1: median 16 ms, p75 74 ms)?  Per step: wall clock,
GPU time between events around the step, and the wall clock of each phase (forward / loss / backward / optimiser) with a synchronize
after each; run twice: blob built on the CPU every step (as the example does) and one blob reused."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import train_dense_ddp as ex  # noqa: E402
from lc_amd.config import AttrDict  # noqa: E402
from lc_amd.losses import Loss_fn  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
np.random.seed(0)
model = ex.DenseNet(3, 64).to(dev).to(memory_format=torch.channels_last)
cfg = AttrDict(pose_loss_cfg=dict(type="cov", clip_weight_grad=True, clip_scale_grad=True, clip_pts_grad=True, dense_sample=2, max_err_len=32),
               pose_loss_start_step=4, pose_loss_start_epoch=0, loss_pose_nz_step=0, w_loss_seg=1, w_loss_pose=0.05, seg_loss_type="L1", w_loss_noc=1)
loss_fn = Loss_fn(cfg, AttrDict(), 0).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4)


def run(reuse_blob, steps=40, idle_ms=0):
    rows = []
    blob0 = ex.synthetic_blob(32, dev, seed=0, binary=False)
    for step in range(steps):
        blob = blob0 if reuse_blob else ex.synthetic_blob(32, dev, seed=step, binary=False)
        if idle_ms:
            time.sleep(idle_ms * 1e-3)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t = [time.perf_counter()]
        e0.record()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            noc, wl, ws, vis = model(blob["rgb_in"].contiguous(memory_format=torch.channels_last))
        torch.cuda.synchronize(dev); t.append(time.perf_counter())
        ld, wd = loss_fn(blob, {"xyz_noc": noc, "xyz_weight_logits": wl, "xyz_weights_scale": ws, "msk_vis_logits": vis}, 0, step, 100)
        loss = sum(wd.values())
        torch.cuda.synchronize(dev); t.append(time.perf_counter())
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.cuda.synchronize(dev); t.append(time.perf_counter())
        opt.step()
        e1.record()
        torch.cuda.synchronize(dev); t.append(time.perf_counter())
        rows.append([1e3 * (t[-1] - t[0]), e0.elapsed_time(e1)] + [1e3 * (b - a) for a, b in zip(t, t[1:])])
    return np.array(rows)


print("host threads:", torch.get_num_threads(), "of", os.cpu_count(), "cpus")
for name, kw in (("blob built on the CPU every step", dict(reuse_blob=False)), ("one blob reused, no idle time", dict(reuse_blob=True)),
                 ("one blob reused, 150 ms of host sleep before every step", dict(reuse_blob=True, idle_ms=150)),
                 ("blob built on the CPU every step, torch.set_num_threads(4)", dict(reuse_blob=False, threads=4))):
    if "threads" in kw:
        torch.set_num_threads(kw.pop("threads"))
    r = run(**kw)[3:]
    slow = r[:, 0] > 2 * np.median(r[:, 0])
    print(f"== {name}: median wall {np.median(r[:, 0]):.1f} ms, {int(slow.sum())} of {len(r)} steps above twice the median")
    print("   wall / gpu-event / fwd / loss / bwd / opt [ms], fast steps (median):", np.round(np.median(r[~slow], 0), 1).tolist())
    if slow.any():
        print("   the same, slow steps (median):                                    ", np.round(np.median(r[slow], 0), 1).tolist())
        print("   slow step indices:", (np.flatnonzero(slow) + 3).tolist())
        print("   where the slow steps spend it (per step, fwd / loss / bwd / opt):", np.round(r[slow][:6, 2:], 0).tolist())
