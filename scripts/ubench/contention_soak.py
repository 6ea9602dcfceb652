#!/usr/bin/env python3
"""Soak of "contention costs time, never a pose": zlmo's test-time chain (64 objects x 16 384 candidates; split selection + two split solves) run
`rounds` times, eager and replayed, while a helper kernel (tests/native/occupy.hip) holds a random share of the chip in a random way for a
random time; every result compared bit for bit with the undisturbed call; rescues counted from the workspaces' tails.
    python scripts/ubench/contention_soak.py [rounds=40] [seed=1]      (on the MI355X)"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lc_amd import splitws, synth  # noqa: E402
from lc_amd.config import AttrDict  # noqa: E402
from lc_amd.inference import GraphedSolvePnP, solve_pnp  # noqa: E402

rounds, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 40), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
FILLER = sys.argv[3] if len(sys.argv) > 3 else "held-units"  # | matmul | head-backward: a second stream of real work instead of the helper kernel
so = os.path.join(ROOT, "build", "tests", "liboccupy.so")
if not os.path.exists(so):
    import subprocess
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "tests", "native", "occupy.hip"), "-o", so], check=True)
lib = ctypes.CDLL(so)
lib.occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
dev = torch.device("cuda:0")
cus = torch.cuda.get_device_properties(0).multi_processor_count
cfg, gt_c, out_c = synth.test_time_inputs("zlmo", B=64, seed=5)
cfg = AttrDict(cfg)
mv = lambda d: {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in d.items()}  # noqa: E731
gt, out = mv(gt_c), mv(out_c)
want = solve_pnp(cfg, out, gt)["weighted-filtered"].clone()
solver = GraphedSolvePnP(cfg, out, gt)
assert torch.equal(solver(out, gt)["weighted-filtered"], want)
torch.cuda.synchronize()
side = torch.cuda.Stream()
g = torch.Generator().manual_seed(seed)


def rescues():
    total = 0
    for (kind, *_), ws in list(splitws._CACHE.items()) + [(("pnp",), solver._split_ws[0]), (("select",), solver._split_ws[1])]:
        pb = splitws.PNP_POSE_BYTES if kind == "pnp" else splitws.SELECT_POSE_BYTES
        n = ws.numel() // pb
        total += int(ws[:n * pb].view(torch.int32).view(n, pb // 4)[:, -30].sum())
    return total


bad = 0
slow = []
r0 = rescues()
if FILLER == "matmul":
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
elif FILLER == "head-backward":
    from lc_amd.ptnet import spatial_softargmax_2d_std
    logits = synth.make_head_logits(64, 64, 64, 64, seed=1).to(dev).requires_grad_(True)
for it in range(rounds):
    if FILLER == "held-units":
        how = ("waves", "lds")[int(torch.randint(0, 2, (1,), generator=g))]
        free = int(torch.randint(1, 65, (1,), generator=g))
        ms = float(torch.rand(1, generator=g)) * 40 + 5
        rc = lib.occupy(2 * (cus - free) if how == "waves" else cus - free, 1024 if how == "waves" else 256, 0 if how == "waves" else 160 * 1024, int(ms * 1e5),
                        ctypes.c_void_p(side.cuda_stream))
        assert rc == 0
    else:
        with torch.cuda.stream(side):
            for _ in range(30):
                if FILLER == "matmul":
                    a @ a
                else:
                    m_, s_ = spatial_softargmax_2d_std(logits)
                    torch.autograd.grad(m_.sum() + s_.sum(), logits)
    t0 = time.perf_counter()
    eager = solve_pnp(cfg, out, gt)["weighted-filtered"]
    replay = solver(out, gt)["weighted-filtered"].clone()
    torch.cuda.synchronize()
    slow.append((time.perf_counter() - t0) * 1e3)
    bad += int(not torch.equal(eager, want)) + int(not torch.equal(replay, want))
    side.synchronize()
r1 = rescues()
assert torch.equal(solve_pnp(cfg, out, gt)["weighted-filtered"], want)
print(f"{rounds} rounds (seed {seed}) of the zlmo chain, eager + replayed, beside " + ("a helper holding all but 1..64 compute units for 5..45 ms:" if FILLER == "held-units" else f"a second stream of 30 x {FILLER}:"))
print(f"results that differ from the undisturbed call: {bad} of {2 * rounds}")
print(f"units (poses / objects) recomputed by rescue launches: {r1 - r0}")
print(f"wall clock of a disturbed pair of calls [ms]: median {sorted(slow)[len(slow) // 2]:.1f}, max {max(slow):.1f}  (undisturbed: ~0.3)")
sys.exit(1 if bad else 0)
