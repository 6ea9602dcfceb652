#!/usr/bin/env python3
"""Where the host time of the EAGER test-time call goes (cProfile of lc_amd.inference.solve_pnp on the synthetic 64-object batch):
the replayed graph costs ~64 us of GPU time, the eager call ~104 us of wall clock, i.e. it is bound by Python + ctypes + allocator."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd.config import AttrDict  # noqa: E402
from lc_amd.inference import solve_pnp  # noqa: E402
from lc_amd.synth import dense_inputs  # noqa: E402

dev = torch.device("cuda:0")
if os.environ.get("CFG"):  # a reference config's knobs (lc_amd.synth.TEST_TIME_CONFIGS)
    from lc_amd import synth

    cfg, gt, out = synth.test_time_inputs(os.environ["CFG"], B=64, seed=3)
    cfg = AttrDict(cfg)
else:
    gt, out = dense_inputs(B=64, H=64, W=64, seed=3)
    out["xyz_weight_logits"] = out["xyz_weight_logits"] + 3 * gt["msk_vis"][:, None]
    out["msk_vis_logits"] = (gt["msk_vis"][:, None] * 2 - 1) * 4
    cfg = AttrDict(dense_point_select="quantile_in_mask", quantile=0.5, dense_sample=2, solvers=["weighted", "weighted_filtered"])
gt = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in gt.items()}
out = {k: v.to(dev).contiguous() for k, v in out.items()}
for _ in range(20):
    solve_pnp(cfg, out, gt)
torch.cuda.synchronize()
N = 2000
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    solve_pnp(cfg, out, gt)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print(f"{N} eager calls; per-call microseconds = tottime / {N} * 1e6")
st.print_stats(28)
