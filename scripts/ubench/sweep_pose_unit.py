#!/usr/bin/env python3
"""The pose-unit launch (lc_pose_unit2_f32: LC loss fwd+bwd + weighted PnP in one grid) over B = 256 ... 65536 poses of N = 64
points: event-timed launch duration and poses/s per batch size, one JSON line each.  scripts/profile_round.sh also runs it under
rocprofv3 (kernel trace; SQ counters) and scripts/summarize_prof.py groups those dispatches by grid size."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd import _lib, synth  # noqa: E402

BATCHES = tuple(int(x) for x in os.environ["LC_SWEEP_BATCHES"].split(",")) if os.environ.get("LC_SWEEP_BATCHES") else (256, 512, 1024, 2048, 4096, 16384, 65536)


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    P = _lib.ptr
    N = 64
    reps = int(os.environ.get("LC_SWEEP_REPS", "30"))
    for B in BATCHES:
        b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=977).items()}
        loss = torch.empty(B, device=dev)
        o = [torch.empty_like(b["pts2d"]), torch.empty_like(b["inv_std"]), torch.empty_like(b["pts3d"]), torch.empty_like(b["start"]),
             torch.empty(B, device=dev), torch.empty(B, device=dev, dtype=torch.int32)]
        go = torch.full((B,), 1.0 / B, device=dev)
        sd = b["inv_std"].contiguous()

        def one():
            rc = lib.lc_pose_unit2_f32(P(b["K"]), P(b["pose"]), P(b["pts3d"]), P(b["pts2d"]), P(b["inv_std"]), None, P(b["bbox_3d"]), P(go), B, N, 32.0, 3.0, 4.0, P(loss), P(o[0]), P(o[1]), P(o[2]), P(sd), P(b["start"]), P(o[3]), P(o[4]), P(o[5]), None, 50, 1e-6, None, 0, _lib.stream_ptr(dev))
            assert rc == 0
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            one()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        assert int(o[5].sum().item()) == 0
        print(json.dumps({"B": B, "N": N, "grid_workgroups": 2 * B, "us_per_launch": round(us, 2), "poses_per_s": B / (us * 1e-6)}), flush=True)


if __name__ == "__main__":
    main()
