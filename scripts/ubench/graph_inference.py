#!/usr/bin/env python3
"""Is the dense test-time pipeline hipGraph-capturable, and what does a replay cost next to the eager call?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd.config import AttrDict  # noqa: E402
from lc_amd.inference import solve_pnp  # noqa: E402
from lc_amd.synth import dense_inputs  # noqa: E402

dev = torch.device("cuda:0")
B, S = int(os.environ.get("B", 64)), int(os.environ.get("SIZE", 64))
if os.environ.get("HEAD", "xyz") == "bin":  # ZebraPose structure: binary surface codes + a model transform (zlmo: SIZE=128)
    from lc_amd.synth import bin_inputs
    gt, out = bin_inputs(B=B, H=S, W=S, seed=3)
else:
    gt, out = dense_inputs(B=B, H=S, W=S, seed=3)
out["xyz_weight_logits"] = out["xyz_weight_logits"] + 3 * gt["msk_vis"][:, None]
out["msk_vis_logits"] = (gt["msk_vis"][:, None] * 2 - 1) * 4
gt = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in gt.items()}
# network outputs are contiguous NCHW tensors (a convolution's output); the synthetic generator hands xyz_noc over as a strided
# view of a channels-last array, which the front end would first copy (one 6 us torch launch per call that is not the pipeline's)
out = {k: v.to(dev).contiguous() for k, v in out.items()}
cfg = AttrDict(dense_point_select="quantile_in_mask", quantile=0.5, dense_sample=2, solvers=["weighted", "weighted_filtered"])


def timeit(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


eager = solve_pnp(cfg, out, gt)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        solve_pnp(cfg, out, gt)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    res = solve_pnp(cfg, out, gt)
graph.replay()
torch.cuda.synchronize()
same = all(torch.equal(res[k], eager[k]) for k in eager)
print(f"captured: results equal to eager: {same}")
print(f"eager  {timeit(lambda: solve_pnp(cfg, out, gt)):7.1f} us per call ({B} objects, {S}x{S} maps)")
runs = sorted(timeit(graph.replay, 200) for _ in range(9))
print(f"replay {runs[4]:7.1f} us per call (median of 9 x 200 replays; min {runs[0]:.1f}, max {runs[-1]:.1f})")
