"""Workspaces of the launches that give ONE unit of work to SEVERAL workgroups (the weighted-PnP solve of thousands of points per pose:
`lc_pnp_lm3_f32`; front end + selection of thousands of candidates per object: `lc_dense_frontend_select3`).  Such a workspace is zeroed once
and then belongs to those launches -- each leaves it ready for the next on the same stream -- so it is cached per (kind, device, stream), or
owned by a captured graph.  The workgroups of such a launch wait for each other and the launch fills the chip: callers that overlap launches on
several streams turn the forms off (`no_split`, or LC_AMD_PNP_SPLIT=0 for the process)."""
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
    half would wait for workgroups that cannot start -- until the wait's bound (about a second) fails the units concerned."""
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


def get(kind: str, dev, need: int, split=None):
    """-> a workspace of at least `need` bytes for launches of `kind` on the current stream of `dev`, or None (need == 0, split off)."""
    if need <= 0 or split is False or (split is None and is_off()) or _OFF:
        return None
    ws = _OWNED.get(kind)
    if ws is not None and ws.device == dev and ws.numel() >= need:
        return ws
    if torch.cuda.is_current_stream_capturing():  # a graph owns its workspace; without `owned` its zero-fill is a node of the graph
        return torch.zeros(need, device=dev, dtype=torch.uint8)
    key = (kind, dev.index if dev.index is not None else torch.cuda.current_device(), _lib.raw_stream(dev))
    ws = _CACHE.get(key)
    if ws is None or ws.numel() < need:
        if len(_CACHE) >= 64:
            _CACHE.clear()
        ws = _CACHE[key] = torch.zeros(need, device=dev, dtype=torch.uint8)
    return ws


PNP_MAX_BYTES = 128 * (2 * 8 * 64 * 8 + 128)       # lc_pnp_lm_workspace_bytes at its largest batch (include/lc_amd.h)
SELECT_MAX_BYTES = 128 * (2 * 8 * 256 * 8 + 128)   # lc_dense_frontend_select_workspace_bytes likewise
