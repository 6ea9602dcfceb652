#!/usr/bin/env python3
"""A/B of PnP-kernel build variants at the metric shape (B=256, N=64): event-timed lc_pnp_lm3_f32 and lc_pose_unit2_f32 launches,
each library in its own child process (LC_AMD_LIB).  usage: pnp_ab.py name=path.so [name=path.so ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, os, sys, torch
sys.path.insert(0, %r)
from lc_amd import _lib, synth
lib = _lib.load(); P = _lib.ptr
dev = torch.device("cuda:0")
res = {}
for B in (256, 65536):
    b = {k: v.to(dev) for k, v in synth.make_batch(B, 64, seed=0).items()}
    st = torch.empty_like(b["start"]); tr = torch.empty(B, device=dev); ret = torch.empty(B, device=dev, dtype=torch.int32)
    loss = torch.empty(B, device=dev); du = torch.empty_like(b["pts2d"]); ds = torch.empty_like(b["pts2d"]); dx = torch.empty_like(b["pts3d"])
    go = torch.full((B,), 1.0 / B, device=dev)
    s = _lib.stream_ptr(dev)
    def pnp():
        assert lib.lc_pnp_lm3_f32(P(b["K"]), P(b["pts3d"]), P(b["pts2d"]), None, P(b["inv_std"]), None, None, P(b["start"]), P(st), P(tr), P(ret), None, B, 64, 50, 1e-6, 0, 0, None, 0, s) == 0
    def unit():
        assert lib.lc_pose_unit2_f32(P(b["K"]), P(b["pose"]), P(b["pts3d"]), P(b["pts2d"]), P(b["inv_std"]), None, P(b["bbox_3d"]), P(go), B, 64, 32.0, 3.0, 4.0, P(loss), P(du), P(ds), P(dx), P(b["inv_std"]), P(b["start"]), P(st), P(tr), P(ret), None, 50, 1e-6, None, 0, s) == 0
    for name, fn in (("pnp", pnp), ("unit", unit)):
        reps = 300 if B == 256 else 20
        for _ in range(10): fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        res[f"{name}_B{B}_us"] = round(best, 2)
    res[f"invalid_B{B}"] = int(ret.sum().item())
    res[f"state_sum_B{B}"] = float(st.double().sum().item())
print(json.dumps(res))
''' % ROOT

for arg in sys.argv[1:]:
    name, path = arg.split("=", 1)
    out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, LC_AMD_LIB=os.path.abspath(path)), capture_output=True, text=True)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    print(name, line[-1] if line else out.stderr[-500:], flush=True)
