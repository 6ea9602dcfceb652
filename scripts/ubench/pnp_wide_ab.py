#!/usr/bin/env python3
"""A/B of builds of the wide (N > 64) LM solve at the dense shapes: event-timed lc_pnp_lm3_f32 launches (ctypes, ~6 us of host time
per call: below that the figure is host-bound) and a checksum of the outputs.  usage: pnp_wide_ab.py name=path.so [...]"""
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
for B, N, frac in ((64, 1024, 0.5), (128, 1024, 0.5), (32, 1024, 1.0), (32, 1849, 1.0), (64, 300, 1.0), (16, 4096, 1.0), (64, 256, 1.0), (256, 256, 1.0), (256, 128, 1.0)):
    b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=0).items()}
    cnt = torch.full((B,), int(N * frac), dtype=torch.int32, device=dev)
    st = torch.empty_like(b["start"]); tr = torch.empty(B, device=dev); ret = torch.empty(B, device=dev, dtype=torch.int32)
    s = _lib.stream_ptr(dev)
    def pnp():
        assert lib.lc_pnp_lm3_f32(P(b["K"]), P(b["pts3d"]), P(b["pts2d"]), None, P(b["inv_std"]), None, P(cnt), P(b["start"]), P(st), P(tr), P(ret), None, B, N, 50, 1e-6, 0, 0, None, 0, s) == 0
    for _ in range(10): pnp()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): pnp()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 100 * 1e3)
    res[f"B{B}_N{N}_n{int(N * frac)}_us"] = round(best, 2)
    res[f"B{B}_N{N}_sum"] = float(st.double().sum().item()) + float(ret.sum().item())
print(json.dumps(res))
''' % ROOT

for arg in sys.argv[1:]:
    name, path = arg.split("=", 1)
    out = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, LC_AMD_LIB=os.path.abspath(path)), capture_output=True, text=True, timeout=600)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    print(name, line[-1] if line else out.stderr[-800:], flush=True)
