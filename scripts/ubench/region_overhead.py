"""What a timed region of bench.py costs beyond its K kernel launches: wall clock (synchronize .. replay .. synchronize) against the GPU span
by events, for the K-step region graph at K = 20 and K = 200.  (HSA_ENABLE_INTERRUPT=0 changes nothing: the wait already polls.)"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from lc_amd import _lib, synth
from lc_amd.inference import quiet_capture
lib = _lib.load(); P = _lib.ptr
dev = torch.device("cuda:0"); B = 256
b = {k: v.to(dev) for k, v in synth.make_batch(B, 64, seed=0).items()}
st = torch.empty_like(b["start"]); tr = torch.empty(B, device=dev); ret = torch.empty(B, device=dev, dtype=torch.int32)
loss = torch.empty(B, device=dev); du = torch.empty_like(b["pts2d"]); ds = torch.empty_like(b["pts2d"]); dx = torch.empty_like(b["pts3d"])
go = torch.full((B,), 1.0 / B, device=dev)
def unit():
    assert lib.lc_pose_unit2_f32(P(b["K"]), P(b["pose"]), P(b["pts3d"]), P(b["pts2d"]), P(b["inv_std"]), None, P(b["bbox_3d"]), P(go), B, 64, 32.0, 3.0, 4.0, P(loss), P(du), P(ds), P(dx), P(b["inv_std"]), P(b["start"]), P(st), P(tr), P(ret), None, 50, 1e-6, None, 0, _lib.stream_ptr(dev)) == 0
for K in (20, 200):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): unit()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with quiet_capture(), torch.cuda.graph(g):
        for _ in range(K): unit()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    walls, calls, evs = [], [], []
    for _ in range(51):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record(); g.replay(); t1 = time.perf_counter(); e1.record(); torch.cuda.synchronize(); t2 = time.perf_counter()
        walls.append((t2 - t0) * 1e6); calls.append((t1 - t0) * 1e6); evs.append(e0.elapsed_time(e1) * 1e3)
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"K={K}: wall {med(walls):.1f} us, replay() call returns after {med(calls):.1f} us, GPU span by events {med(evs):.1f} us = {med(evs)/K:.2f} us per step; wall - span = {med(walls)-med(evs):.1f} us")
    # without events
    walls = []
    for _ in range(51):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); walls.append((time.perf_counter() - t0) * 1e6)
    print(f"      wall without events {med(walls):.1f} us = {med(walls)/K:.2f} per step")
    # stream sync instead of device sync
    cs = torch.cuda.current_stream()
    walls = []
    for _ in range(51):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); g.replay(); cs.synchronize(); walls.append((time.perf_counter() - t0) * 1e6)
    print(f"      wall with stream.synchronize {med(walls):.1f} us")
