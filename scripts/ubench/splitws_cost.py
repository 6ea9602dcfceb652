#!/usr/bin/env python3
"""Host cost of the split forms' bookkeeping on the eager path (per call, microseconds)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lc_amd import _lib, splitws  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
torch.zeros(1, device=dev)


def per_call(fn, n=20000):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


print(f"lc_dense_frontend_select_workspace_bytes  {per_call(lambda: lib.lc_dense_frontend_select_workspace_bytes(64, 128, 128, 0, 0, 1)):6.2f} us")
print(f"splitws.get (cached)                       {per_call(lambda: splitws.get('select', dev, 1 << 20)):6.2f} us")
print(f"torch.cuda.is_current_stream_capturing     {per_call(torch.cuda.is_current_stream_capturing):6.2f} us")
print(f"torch.cuda.current_stream(dev).cuda_stream {per_call(lambda: torch.cuda.current_stream(dev).cuda_stream):6.2f} us")
print(f"_lib.stream_ptr(dev)                       {per_call(lambda: _lib.stream_ptr(dev)):6.2f} us")
