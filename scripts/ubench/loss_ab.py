#!/usr/bin/env python3
"""A/B of builds of the LC-loss kernel: event-timed lc_cov_loss3_fwd_bwd_f32 launches (ctypes) at the metric shape and the dense
shapes, with a checksum of the outputs.  usage: loss_ab.py name=path.so [...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import json, sys, torch
sys.path.insert(0, %r)
from lc_amd import _lib, synth
lib = _lib.load(); P = _lib.ptr
dev = torch.device("cuda:0")
res = {}
for B, N in ((256, 64), (4096, 64), (32, 1024), (32, 1849), (256, 256)):
    b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=0).items()}
    go = torch.full((B,), 1.0 / B, device=dev)
    loss = torch.empty(B, device=dev); du = torch.empty_like(b["pts2d"]); ds = torch.empty_like(b["pts2d"]); dx = torch.empty_like(b["pts3d"])
    nws = int(lib.lc_cov_loss_workspace_bytes(B, N)); ws = torch.zeros(max(nws, 1), dtype=torch.uint8, device=dev)
    s = _lib.stream_ptr(dev)
    def run():
        assert lib.lc_cov_loss3_fwd_bwd_f32(P(b["K"]), P(b["pose"]), P(b["pts3d"]), P(b["pts2d"]), P(b["inv_std"]), None, P(b["bbox_3d"]), P(go), B, N, 32.0, 3.0, 4.0, 0,
                                            P(loss), P(du), P(ds), P(dx), None, P(ws) if nws else None, nws, s) == 0
    for _ in range(10): run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
    res[f"B{B}_N{N}_us"] = round(best, 2)
    res[f"B{B}_N{N}_sum"] = float(loss.double().sum().item()) + float(du.double().abs().sum().item()) + float(ds.double().abs().sum().item())
print(json.dumps(res))
''' % ROOT

for arg in sys.argv[1:]:
    name, path = arg.split("=", 1)
    out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, LC_AMD_LIB=os.path.abspath(path)), capture_output=True, text=True, timeout=600)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    print(name, line[-1] if line else out.stderr[-800:], flush=True)
