import ctypes, os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from lc_amd import _lib, splitws
from lc_amd.dense import dense_front_end_select
from tests.test_gpu_contention import _select_inputs, _tail, SEL_POSE_BYTES
DEV="cuda:0"
lib = ctypes.CDLL("build/tests/liboccupy.so")
lib.occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
side = torch.cuda.Stream()
B,H,W=64,128,128
xyz, wl, ws_, ns, vl = _select_inputs(B,H,W,5)
kw = dict(seg_thresh=0.5, sample=1, quantile=0.2, min_count=6, seed=3)
want = dense_front_end_select(xyz, wl, ws_, ns, vl, "quantile_in_mask", split=True, **kw)
torch.cuda.synchronize()
work = splitws.get("select", torch.device(DEV), _lib.load().lc_dense_frontend_select_workspace_bytes(B, H, W, 0, 0, 1), True)
print("tail0", [int(t.sum()) for t in _tail(work,B,SEL_POSE_BYTES)])
for free in (32, 8, 2):
    for lds in (163840, 160*1024-512):
        t0=time.time()
        rc = lib.occupy((256-free), 256, lds, int(40*1e5), ctypes.c_void_p(side.cuda_stream))
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        got = dense_front_end_select(xyz, wl, ws_, ns, vl, "quantile_in_mask", split=True, **kw)
        e1.record()
        torch.cuda.synchronize()
        print("free",free,"lds",lds,"rc",rc,"select ms", e0.elapsed_time(e1), "wall", time.time()-t0, "tail", [int(t.sum()) for t in _tail(work,B,SEL_POSE_BYTES)], "equal", torch.equal(got[3],want[3]))
