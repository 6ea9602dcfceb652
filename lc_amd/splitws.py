"""Workspaces of the launches that give ONE unit of work to SEVERAL workgroups (the weighted-PnP solve of thousands of points per pose:
`lc_pnp_lm3_f32`; front end + selection of thousands of candidates per object: `lc_dense_frontend_select3`).  Such a workspace is zeroed once
and then belongs to those launches -- each leaves it ready for the next on the same stream -- so it is cached per (kind, device, stream), or
owned by a captured graph.  The workgroups of such a launch wait for each other (bounded), and every such launch is followed by a rescue
launch that recomputes -- bit for bit -- the units whose workgroups did not all meet (lc_amd/csrc/lc_common.h: SplitSum), so results never depend
on what else holds compute units; contended split launches are merely slow, which is why callers that overlap many launches on several
streams may turn the forms off (`no_split`, or LC_AMD_PNP_SPLIT=0 for the process)."""
from __future__ import annotations

import contextlib
import os

import torch

from . import _lib

_CACHE = {}   # (kind, device index, stream) -> zeroed-once uint8 tensor
_OWNED = {}   # kind -> tensor handed over by `owned`
_OFF = 0


def is_off() -> bool:
    """Inside `no_split()`, or LC_AMD_PNP_SPLIT=0: launches whose workgroups wait for each other are not to be used."""
    return bool(_OFF) or os.environ.get("LC_AMD_PNP_SPLIT", "1") == "0"


@contextlib.contextmanager
def no_split():
    """Launches inside the block take one workgroup per unit whatever their shape.  For callers that run several of them CONCURRENTLY on one
    device (side streams): a split launch fills the chip by itself (its workgroups take a compute unit each), and two of them admitted half and
    half would wait for workgroups that cannot start until the wait's bound (milliseconds) hands the units concerned to the rescue launch --
    correct, but slower than one workgroup per unit from the start."""
    global _OFF
    _OFF += 1
    try:
        yield
    finally:
        _OFF -= 1


@contextlib.contextmanager
def owned(**by_kind):
    """Launches inside the block use these workspaces (kind -> zeroed uint8 tensor, allocated by the caller BEFORE a stream capture: the launches
    keep them consistent from one to the next, so a replayed graph needs no fill node) instead of the per-stream ones.  kind -> None: back to
    the per-stream workspace.  The caller orders the launches that share one."""
    prev = {k: _OWNED.get(k) for k in by_kind}
    _OWNED.update(by_kind)
    try:
        yield
    finally:
        for k, v in prev.items():
            if v is None:
                _OWNED.pop(k, None)
            else:
                _OWNED[k] = v


_SEEN_RESCUES = {}


def get(kind: str, dev, need: int, split=None):
    """-> a workspace of at least `need` bytes for launches of `kind` on the current stream of `dev`, or None (need == 0, split off)."""
    if need <= 0 or split is False or (split is None and is_off()) or _OFF:
        return None
    if os.environ.get("LC_AMD_SPLIT_DEBUG") == "1" and not torch.cuda.is_current_stream_capturing():  # a debug log: it synchronises
        n = rescues(kind, dev)
        if n != _SEEN_RESCUES.get(kind, 0):
            print(f"lc_amd.splitws: {n} unit(s) of '{kind}' launches recomputed by rescue launches so far (contended split launches: consider no_split())", flush=True)
            _SEEN_RESCUES[kind] = n
    ws = _OWNED.get(kind)
    if ws is not None:
        index = lambda d: d.index if d.index is not None else torch.cuda.current_device()  # noqa: E731  (torch.device("cuda") == the current device)
        if ws.is_cuda and index(ws.device) == index(dev) and ws.numel() >= need:
            return ws
        import warnings

        warnings.warn(f"lc_amd.splitws: the owned '{kind}' workspace ({ws.numel()} bytes on {ws.device}) does not serve a launch that needs {need} bytes "
                      f"on {dev}; falling back to a {'zero-filled tensor inside the capture' if torch.cuda.is_current_stream_capturing() else 'per-stream workspace'}",
                      RuntimeWarning, stacklevel=3)
    if torch.cuda.is_current_stream_capturing():  # a graph owns its workspace; without `owned` its zero-fill is a node of the graph
        return torch.zeros(need, device=dev, dtype=torch.uint8)
    key = (kind, dev.index if dev.index is not None else torch.cuda.current_device(), _lib.raw_stream(dev))
    ws = _CACHE.get(key)
    if ws is None or ws.numel() < need:
        if len(_CACHE) >= 64:
            _CACHE.clear()
        ws = _CACHE[key] = torch.zeros(need, device=dev, dtype=torch.uint8)
    return ws


def rescues(kind: str, dev=None) -> int:
    """Units the rescue launches had to recompute so far on the workspaces of `kind` ('pnp' | 'select') that this process holds for `dev`
    (`lc_split_workspace_rescues`; synchronises).  Non-zero = the split launches were contended: results are unaffected, the launches were
    slow -- wrap the calls in `no_split()`.  LC_AMD_SPLIT_DEBUG=1 makes `get` print it whenever it grows."""
    lib = _lib.load()
    total = 0
    for (k, index, _stream), ws in list(_CACHE.items()) + [((k, None, None), w) for k, w in _OWNED.items() if w is not None]:
        if k != kind or (dev is not None and index is not None and index != (dev.index if dev.index is not None else torch.cuda.current_device())):
            continue
        with _lib.on_device(ws.device):
            n = int(lib.lc_split_workspace_rescues(_lib.ptr(ws), ws.numel(), 0 if kind == "pnp" else 1, _lib.stream_ptr(ws.device)))
        if n < 0:
            raise RuntimeError(lib.lc_amd_last_error().decode())
        total += n
    return total


PNP_POSE_BYTES = 2 * 8 * 64 * 8 + 128        # lc_common.h kSplitPoseBytes (tests/test_gpu_pnp_split.py checks it against lc_pnp_lm_workspace_bytes)
SELECT_POSE_BYTES = 2 * 8 * 256 * 8 + 128    # lc_select.hip kSelSplitPoseBytes


def max_bytes(kind: str, dev=None) -> int:
    """The largest workspace launches of `kind` ('pnp' | 'select') can ask for on `dev`: the split forms admit at most half as many units as the device
    has compute units (two workgroups per unit at least) -- 128 on a 256-CU MI355X, more on a larger device, fewer on a partition."""
    cus = torch.cuda.get_device_properties(dev if dev is not None else torch.cuda.current_device()).multi_processor_count
    return max(cus // 2, 1) * (PNP_POSE_BYTES if kind == "pnp" else SELECT_POSE_BYTES)


PNP_MAX_BYTES = 128 * PNP_POSE_BYTES         # on a 256-CU device (kept for callers that size buffers without a device at hand)
SELECT_MAX_BYTES = 128 * SELECT_POSE_BYTES
