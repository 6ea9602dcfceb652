#!/usr/bin/env python3
"""The lc_* kernels that stream maps inside a zlmo-shaped training step (configs/zlmo.yaml: B=32, 21 code planes, 128x128 maps, dense_sample 3),
each launched `reps` times back to back on FRESH-ish inputs (a ring of input buffers larger than the 256 MiB Infinity Cache, so a launch does not
re-read what the previous one left in it) -- for rocprofv3 (`--kernel-trace --stats`, `--pmc ...`) and for a quick event-timed table.
    python scripts/ubench/zlmo_stream.py [--dtype f16] [--reps 30] [--only xyz_bin_fwd,...]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16", choices=["f32", "f16", "bf16"])
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    from lc_amd import dense, dense_aux, floatbits
    from lc_amd.grad import NormClipper

    dev = torch.device("cuda:0")
    dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    e = torch.empty((), dtype=dt).element_size()
    B, C, S, sample = a.batch, 21, 128, 3
    HW, N = S * S, (-(-S // sample)) ** 2
    g = torch.Generator().manual_seed(0)
    ring = max(2, int(300e6 // (B * C * HW * (e + 1))) + 1)  # input sets: more bytes than the Infinity Cache holds
    sets = []
    for i in range(ring):
        lg = (torch.randn(B, C, S, S, generator=g) * 2).to(dt).to(dev)
        sets.append(dict(lg=lg, bits=(torch.rand(B, C, S, S, generator=g) < 0.5).to(dev), raw=(torch.rand(B, C, S, S, generator=g) < 0.5).to(torch.uint8).to(dev),
                         vis=torch.randn(B, 1, S, S, generator=g).to(dt).to(dev), msk=(torch.rand(B, S, S, generator=g) < 0.7).to(dev),
                         wl=torch.randn(B, 2, S, S, generator=g).to(dt).to(dev)))
    hist = torch.full((C,), 0.5, device=dev)
    wscale = torch.full((B, 1, 1, 1), 20.0, device=dev)
    noc_scale = torch.full((B, 3), 40.0, device=dev)
    bits3 = [7, 7, 7]
    out = {}

    def timed(name, fn):
        if a.only and name not in a.only.split(","):
            return
        for i in range(3):
            fn(sets[i % ring])
        torch.cuda.synchronize()
        evs = []
        for i in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s = sets[i % ring]
            e0.record()
            fn(s)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        d = sorted(x.elapsed_time(y) * 1e3 for x, y in evs)
        out[name] = round(d[len(d) // 2], 2)

    # code loss: forward, backward
    timed("xyz_bin_fwd", lambda s: dense_aux.xyz_bin_loss(s["lg"], s["bits"], s["vis"], hist, 0.05))
    leaves = [s["lg"].clone().requires_grad_(True) for s in sets]
    losses = [dense_aux.xyz_bin_loss(x, s["bits"], s["vis"], hist, 0.05) for x, s in zip(leaves, sets)]
    idx = {id(s): i for i, s in enumerate(sets)}
    timed("xyz_bin_bwd", lambda s: torch.autograd.grad(losses[idx[id(s)]], leaves[idx[id(s)]], retain_graph=True))
    # training decode on the strided subset: forward, backward
    b3 = floatbits._bits3(bits3, C)
    timed("decode_gt_fwd", lambda s: floatbits.decode_with_gt_strided(s["lg"], s["raw"], bits3, s["msk"], sample=sample, top_left=(1, 2), out_scale=noc_scale))
    outs = [floatbits.decode_with_gt_strided(x, s["raw"], bits3, s["msk"], sample=sample, top_left=(1, 2), out_scale=noc_scale) for x, s in zip(leaves, sets)]
    gp = torch.randn(outs[0].shape, generator=g).to(dev)
    timed("decode_gt_bwd", lambda s: torch.autograd.grad(outs[idx[id(s)]], leaves[idx[id(s)]], gp, retain_graph=True))
    # front end on the weight logits (binary-code heads: no xyz planes), forward and backward
    wls = [s["wl"].clone().requires_grad_(True) for s in sets]
    timed("frontend_fwd", lambda s: dense.dense_front_end(None, s["wl"], wscale, None, sample=sample, top_left=(1, 2)))
    fo = [dense.dense_front_end(None, w, wscale, None, sample=sample, top_left=(1, 2)) for w in wls]
    gi = torch.randn(fo[0][1].shape, generator=g).to(dev)
    timed("frontend_bwd", lambda s: torch.autograd.grad(fo[idx[id(s)]][1], wls[idx[id(s)]], gi, retain_graph=True))
    # the clipper on the weight logits' gradient
    clip = NormClipper().to(dev)
    timed("norm_clip (sqnorm + apply)", lambda s: clip.clip(s["wl"]))
    print(json.dumps({"shape": [B, C, S, S], "dtype": a.dtype, "sample": sample, "N": N, "ring": ring, "median_us_events": out}))


if __name__ == "__main__":
    main()
