#!/usr/bin/env python3
"""Tiled (one wavefront-sized workgroup per 64 correspondences) against one-workgroup form of the LC-loss kernel at the dense
shapes: bit-equality of the outputs and event-timed launches.  usage: tiled_loss.py [name=path.so ...]  (build variants,
each in its own child process via LC_AMD_LIB; no argument = the shipped library)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, torch
sys.path.insert(0, %r)
from lc_amd import synth, cov_mixed as cm
dev = torch.device("cuda:0")
def t(fn, reps=50):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
for B, N in ((32, 1024), (32, 1849), (64, 1024), (64, 4096), (8, 1024), (1, 4096), (256, 1024)):
    b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=1).items()}
    args = (b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], None, b["bbox_3d"])
    a = cm.loss_cov_mixed_fused(*args, tiled=False); c = cm.loss_cov_mixed_fused(*args, tiled=True)
    torch.cuda.synchronize()
    eq = all(bool(torch.equal(x, y)) for x, y in zip(a[:4], c[:4]))
    ws = cm.tiled_workspace(dev, B, N)
    hdr = None if ws is None else ws.view(torch.int32)[:4].tolist()
    print("  B=%%d N=%%d equal=%%s hdr=%%s one-workgroup %%.1f us tiled %%.1f us" %% (B, N, eq, hdr, t(lambda: cm.loss_cov_mixed_fused(*args, tiled=False)),
          t(lambda: cm.loss_cov_mixed_fused(*args, tiled=True))), flush=True)
''' % ROOT

for arg in (sys.argv[1:] or ["shipped="]):
    name, path = arg.split("=", 1)
    env = dict(os.environ, LC_AMD_LIB=os.path.abspath(path)) if path else dict(os.environ)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    print(name, flush=True)
    print(out.stdout if out.returncode == 0 else out.stdout + out.stderr[-800:], flush=True)
