#!/usr/bin/env python3
"""Timing experiment: the wide front end + selection kernel with and without its threshold search (build -DLC_SELECT_SKIP_SEARCH: wrong
results, never shipped) -- what the search costs in kernel time, which the phase clock of ONE wavefront cannot tell (its waits at the
barriers are the other wavefronts' work)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lc_amd import build  # noqa: E402

LIB = build.variant_path("skipsearch")
if "--build" in sys.argv:
    build.build_variant("skipsearch", ["-DLC_SELECT_SKIP_SEARCH"])
    sys.exit(0)
if "--child" in sys.argv:
    import torch
    from lc_amd import synth
    from lc_amd.dense import dense_front_end_select

    dev = torch.device("cuda:0")
    cfg, gt, out = synth.test_time_inputs("zlmo", B=64, seed=3)
    wl, ws, vl = out["xyz_weight_logits"].to(dev), out["xyz_weights_scale"].to(dev), out["msk_vis_logits"].to(dev)
    fn = lambda: dense_front_end_select(None, wl, ws, None, vl, "quantile_in_mask", quantile=0.2, sample=1)  # noqa: E731
    fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 200 * 1e3)
    print(f"{sorted(ts)[3]:.2f}")
    sys.exit(0)
for name, env in (("shipped", {}), ("without the threshold search", {"LC_AMD_LIB": LIB})):
    o = subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, **env), capture_output=True, text=True)
    print(f"{name:32s} {o.stdout.strip().splitlines()[-1] if o.stdout.strip() else o.stderr[-300:]} us per launch (back to back in a graph)")
