#!/usr/bin/env python3
"""Golden trajectories of the reference's `lib.utils.grad.NormClipper` (imported from /root/reference in this container):
a sequence of gradient tensors -> clipped tensors, `max_norm` after every call, `last_norm`.  Inputs and outputs only.

    python tests/golden/gen_golden_clip.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from lib.utils.grad import NormClipper  # noqa: E402


def run(name, shapes, scales, **kw):
    g = torch.Generator().manual_seed(len(name))
    out = {}
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        clip = NormClipper(**kw)
        for i, (shape, sc) in enumerate(zip(shapes, scales)):
            grad = (torch.randn(*shape, generator=torch.Generator().manual_seed(100 + i), dtype=torch.float64) * sc).to(torch.float32)
            res = clip.clip(grad.to(dtype))  # fp32-representable inputs for both precisions
            if tag == "f32":
                out[f"in_{i}"] = grad.numpy()
            out[f"{tag}_out_{i}"] = res.numpy()
            out[f"{tag}_max_norm_{i}"] = np.asarray(clip.max_norm.item())
            out[f"{tag}_last_norm_{i}"] = np.asarray(float(clip.last_norm))
    out["steps"] = np.asarray(len(shapes))
    for k, v in kw.items():
        out["kw_" + k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, f"clip_{name}.npz"), **out)
    print(name, [float(out[f"f64_max_norm_{i}"]) for i in range(len(shapes))])


if __name__ == "__main__":
    # weight-logit hook shape (B,2,H,W): growing, spiking and shrinking gradients around the running maximum
    run("weights", [(4, 2, 16, 16)] * 7, [1.0, 1.2, 30.0, 0.5, 0.4, 5.0, 0.1])
    # scale / points clippers (rel_thresh=2), tiny tensors, odd sizes (scalar tail of the float4 stream)
    run("scale", [(5, 1, 1, 1)] * 5, [10.0, 400.0, 3.0, 3.0, 1000.0], rel_thresh=2)
    run("points", [(3, 67, 3)] * 5, [2.0, 0.1, 50.0, 1.0, 1.0], rel_thresh=2, initial_max_norm=10)
    # a zero gradient on the first call leaves max_norm at 0: the second call starts again from initial_max_norm
    run("zero_first", [(2, 2, 8, 8)] * 4, [0.0, 3.0, 3.0, 100.0], initial_max_norm=20)
