"""Dense-correspondence front end: network output -> (pts2d, inv_std2d, pts3d) for Loss_cov_mixed / PnP.

`dense_front_end` fuses what the reference does with ~10 torch ops between the dense head and the loss
(`losses.py:355-356` joint softmax over all 2*H*W weight logits times the per-sample scale, `losses.py:142-161`
`dense_pnp_matching_from_xyz`: strided sub-sampling with phase, noc_scale multiply, transposes) into one HIP launch
forward and one backward.  Same random phase source (`np.random.randint`) so seeded runs line up with the reference.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import Tensor

from . import _lib, splitws


def _n_points(H, W, top, left, sample):
    return ((H - top + sample - 1) // sample) * ((W - left + sample - 1) // sample)


def _head_maps(xyz, wlogits, vis_logits=None):
    """The heads' maps as the kernels take them (`_lib.hip_maps`): fp32 / fp16 / bf16 in their own type, dense batches or channel slices of
    the network output in place.  -> ((xyz, wlogits, vis), (batch strides), map_dtype)"""
    B, _, H, W = wlogits.shape
    if xyz is not None and xyz.dtype == torch.float32 and wlogits.dtype != torch.float32:
        # test time, binary-code heads: coordinate planes decoded to fp32 next to logits in the network's 16-bit type -- each in its own type
        (wl, vis), (ws_, vs), code = _lib.hip_maps(xyz_weight_logits=wlogits, msk_vis_logits=None if vis_logits is None else vis_logits.reshape(B, H, W))
        (x,), (xs,), xcode = _lib.hip_maps(xyz_noc=xyz)
        return (x, wl, vis), (xs, ws_, vs), (code, xcode)
    maps, strides, code = _lib.hip_maps(xyz_noc=xyz, xyz_weight_logits=wlogits, msk_vis_logits=None if vis_logits is None else vis_logits.reshape(B, H, W))
    return maps, strides, (code, code)


def _weight_scale(xyz_weights_scale, B):
    """(B,1,1,1) | (B,) weight scale -> contiguous (B,) in its own element type (fp32 under autocast, where `exp` is an fp32 op; the model's
    16-bit type in a pure half-precision model): the kernels read either."""
    (ws,), _, _ = _lib.hip_maps(xyz_weights_scale=xyz_weights_scale.reshape(B))
    return ws if ws.is_contiguous() else ws.contiguous()


def _launch_fwd(xyz, wlogits, wscale, noc_scale, top, left, sample, vis_logits=None, vis_thresh=0.5):
    lib = _lib.load()
    (xyz, wlogits, vis_logits), (xs, ws_, vs), (code, xcode) = _head_maps(xyz, wlogits, vis_logits)
    B, _, H, W = wlogits.shape
    N = _n_points(H, W, top, left, sample)
    f = dict(device=wlogits.device, dtype=torch.float32)
    pts2d, inv_std, lse = torch.empty(B, N, 2, **f), torch.empty(B, N, 2, **f), torch.empty(B, **f)
    pts3d = torch.empty(B, N, 3, **f) if xyz is not None else None
    vis = torch.empty(B, N, device=wlogits.device, dtype=torch.uint8) if vis_logits is not None else None
    with _lib.on_device(wlogits.device):
        rc = lib.lc_dense_frontend_fwd3(_lib.ptr(xyz), _lib.ptr(wlogits), _lib.ptr(wscale), _lib.ptr(noc_scale), _lib.ptr(vis_logits),
                                        float(vis_thresh), code, xcode, _lib.MAP_DTYPES[wscale.dtype], xs, ws_, vs, B, H, W, top, left, sample, _lib.ptr(pts2d), _lib.ptr(inv_std),
                                        _lib.ptr(pts3d), _lib.ptr(lse), _lib.ptr(vis), _lib.stream_ptr(wlogits.device))
    _lib.check(rc, "lc_dense_frontend_fwd3")
    return (pts2d, inv_std, pts3d, lse) if vis_logits is None else (pts2d, inv_std, pts3d, lse, vis)


def _launch_bwd(wlogits, wscale, noc_scale, lse, g_inv_std, g_pts3d, shape, top, left, sample, need, xyz_dtype=None):
    lib = _lib.load()
    B, H, W = shape
    (wlogits,), (ws_,), code = _lib.hip_maps(xyz_weight_logits=wlogits)
    m = dict(device=wlogits.device, dtype=wlogits.dtype)  # the gradient of a map in the map's own type (dense)
    # an fp32 coordinate map next to 16-bit weight logits (`_head_maps`): its gradient is written in ITS type by a launch of its own --
    # the kernel writes both gradient maps in one element type, and rounding an fp32 leaf's gradient to fp16 can flush it to zero
    xyz_apart = need[0] and xyz_dtype is not None and xyz_dtype != wlogits.dtype
    d_xyz = torch.empty(B, 3, H, W, **m) if need[0] and not xyz_apart else None
    d_wl = torch.empty(B, 2, H, W, **m) if need[1] else None
    d_ws = torch.empty(B, device=wlogits.device, dtype=wscale.dtype) if need[2] else None
    with _lib.on_device(wlogits.device):
        rc = 0
        if d_xyz is not None or d_wl is not None or d_ws is not None:
            rc = lib.lc_dense_frontend_bwd2(_lib.ptr(wlogits), _lib.ptr(wscale), _lib.ptr(noc_scale), _lib.ptr(lse), _lib.ptr(g_inv_std),
                                            _lib.ptr(g_pts3d), code, _lib.MAP_DTYPES[wscale.dtype], ws_, B, H, W, top, left, sample, _lib.ptr(d_xyz), _lib.ptr(d_wl), _lib.ptr(d_ws),
                                            _lib.stream_ptr(wlogits.device))
        if rc == 0 and xyz_apart:  # no weight gradient asked for: the logits are not read, only the scatter of g_pts3d x noc_scale runs
            d_xyz = torch.empty(B, 3, H, W, device=wlogits.device, dtype=xyz_dtype)
            rc = lib.lc_dense_frontend_bwd2(_lib.ptr(wlogits), _lib.ptr(wscale), _lib.ptr(noc_scale), _lib.ptr(lse), None, _lib.ptr(g_pts3d),
                                            _lib.MAP_DTYPES[xyz_dtype], _lib.MAP_DTYPES[wscale.dtype], ws_, B, H, W, top, left, sample, _lib.ptr(d_xyz), None, None,
                                            _lib.stream_ptr(wlogits.device))
    _lib.check(rc, "lc_dense_frontend_bwd2")
    return d_xyz, d_wl, d_ws


class _DenseFrontEndFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, wlogits, wscale, noc_scale, top, left, sample):
        pts2d, inv_std, pts3d, lse = _launch_fwd(xyz, wlogits, wscale, noc_scale, top, left, sample)
        ctx.save_for_backward(wlogits, wscale, noc_scale, lse)
        ctx.cfg = (wlogits.shape[0], wlogits.shape[2], wlogits.shape[3], top, left, sample)
        ctx.xyz_dtype = None if xyz is None else xyz.dtype
        ctx.mark_non_differentiable(pts2d)
        if pts3d is None:  # binary-code heads: no continuous xyz input
            pts3d = wlogits.new_zeros(0)
            ctx.mark_non_differentiable(pts3d)
        return pts2d, inv_std, pts3d

    @staticmethod
    def backward(ctx, _g2, g_inv_std, g_pts3d):
        wlogits, wscale, noc_scale, lse = ctx.saved_tensors
        B, H, W, top, left, sample = ctx.cfg
        need = list(ctx.needs_input_grad[:3])
        if g_pts3d is not None and g_pts3d.numel() == 0:
            g_pts3d, need[0] = None, False
        g_inv_std = None if g_inv_std is None else g_inv_std.contiguous().to(torch.float32)
        g_pts3d = None if g_pts3d is None else g_pts3d.contiguous().to(torch.float32)
        d_xyz, d_wl, d_ws = _launch_bwd(wlogits, wscale, noc_scale, lse, g_inv_std, g_pts3d, (B, H, W), top, left, sample, need, ctx.xyz_dtype)
        return d_xyz, d_wl, d_ws, None, None, None, None


def dense_front_end(xyz_noc: Tensor, xyz_weight_logits: Tensor, xyz_weights_scale: Tensor, noc_scale: Tensor = None, sample: int = 2,
                    top_left=None):
    """xyz_noc (B,3,H,W), xyz_weight_logits (B,2,H,W), xyz_weights_scale (B,1,1,1) [or (B,)], noc_scale (B,3)|None
    -> pts2d (B,N,2) pixel grid, inv_std2d (B,N,2), pts3d (B,N,3); differentiable w.r.t. the first three."""
    top, left = np.random.randint(0, sample, size=2) if top_left is None else top_left  # losses.py:152
    B = xyz_weight_logits.shape[0]
    # the maps in their own element type and layout: no `.float()` / `.contiguous()` copy in front of the launch (`_lib.hip_maps`; here,
    # outside the autograd function, the two rare copies it can make -- mixed element types, samples not contiguous -- are differentiable)
    (xyz, wl, _), _, _ = _head_maps(xyz_noc, xyz_weight_logits)
    ws = _weight_scale(xyz_weights_scale, B)
    ns = None if noc_scale is None else _lib.require_hip_f32("noc_scale", noc_scale)
    pts2d, inv_std, pts3d = _DenseFrontEndFn.apply(xyz, wl, ws, ns, int(top), int(left), int(sample))
    return pts2d, inv_std, (pts3d if xyz is not None else None)


@torch.no_grad()
def dense_front_end_with_visibility(xyz_noc: Tensor, xyz_weight_logits: Tensor, xyz_weights_scale: Tensor, noc_scale: Tensor,
                                    msk_vis_logits: Tensor, seg_thresh: float = 0.5, sample: int = 2, top_left=(0, 0)):
    """Test-time form of `dense_front_end` (no autograd) that also returns the visibility mask of the sampled pixels,
    `sigmoid(msk_vis_logits) > seg_thresh` on the stride slice (`test.py:88-90`), from the same launch: (pts2d, inv_std2d, pts3d,
    visible (B,N) bool)."""
    top, left = top_left
    B, _, H, W = xyz_weight_logits.shape
    ws = _weight_scale(xyz_weights_scale, B)
    ns = None if noc_scale is None else _lib.require_hip_f32("noc_scale", noc_scale)
    pts2d, inv_std, pts3d, _lse, vis = _launch_fwd(xyz_noc, xyz_weight_logits, ws, ns, int(top), int(left), int(sample), msk_vis_logits, seg_thresh)
    return pts2d, inv_std, pts3d, vis.view(torch.bool)


SELECT_MODES = {"mask": 0, "quantile": 1, "quantile_in_mask": 2}


FUSED_SELECT_MAX_POINTS = 16384  # lc_dense_frontend_select3 (its keys sit in up to 64 KB of LDS): 128x128 maps at stride 1


def _select_buffers(out, B, N, dev, what):
    if out is None:
        f32, i32 = dict(device=dev, dtype=torch.float32), dict(device=dev, dtype=torch.int32)
        return (torch.empty(B, N, 2, **f32), torch.empty(B, N, 2, **f32), torch.empty(B, N, 3, **f32), torch.empty(B, **i32),
                torch.empty(B, N, **i32))
    o_u, o_w, o_x, o_c, o_i = out
    if not (o_u.shape == o_w.shape == (B, N, 2) and o_x.shape == (B, N, 3) and o_c.shape == (B,) and o_i.shape == (B, N)
            and o_u.dtype == o_w.dtype == o_x.dtype == torch.float32 and o_c.dtype == o_i.dtype == torch.int32
            and all(t.is_contiguous() and t.device == dev for t in out)):
        raise ValueError(f"{what}: `out` buffers must be contiguous tensors of the result shapes on the input's device")
    return out


@torch.no_grad()
def dense_front_end_select(xyz_noc: Tensor, xyz_weight_logits: Tensor, xyz_weights_scale: Tensor, noc_scale: Tensor,
                           msk_vis_logits: Tensor, mode: str, *, seg_thresh: float = 0.5, sample: int = 2, top_left=(0, 0),
                           quantile: float = 0.0, square_weights: bool = True, min_count: int = 4, seed: int = 0, out=None,
                           pose_index_offset: int = 0, split=None):
    """`dense_front_end_with_visibility` followed by `dense_select(..., mask=visible)` as ONE launch (test time, at most
    FUSED_SELECT_MAX_POINTS sampled pixels per object): the (B,N,.) rows in between are never written.  Returns what `dense_select`
    returns -- (pts2d, weights, pts3d, counts, index), bit for bit.  xyz_noc = None: the selection alone -- pts3d comes back unwritten, for the
    caller to fill from `index` / `counts` (binary-code heads: `floatbits.decode_selected_rows`, which decodes the selected pixels only).
    split (default: on unless `splitws.no_split()` / LC_AMD_PNP_SPLIT=0): rows of more than 4096 candidates of at most 128 objects are selected by
    several workgroups per object (`lc_dense_frontend_select3` + workspace) -- the same outputs bit for bit, also when something else holds compute units
    (the one-workgroup kernel behind the split launch selects again whatever object's workgroups did not all meet)."""
    lib = _lib.load()
    top, left = top_left
    B, _, H, W = xyz_weight_logits.shape
    (xyz, wl, vl), (xs, wls, vs), (code, xcode) = _head_maps(xyz_noc, xyz_weight_logits, msk_vis_logits)
    ws = _weight_scale(xyz_weights_scale, B)
    ns = None if noc_scale is None else _lib.require_hip_f32("noc_scale", noc_scale)
    N = -(-(H - top) // sample) * -(-(W - left) // sample)
    dev = wl.device
    o_u, o_w, o_x, o_c, o_i = _select_buffers(out, B, N, dev, "dense_front_end_select")
    with _lib.on_device(dev):
        work = splitws.get("select", dev, int(lib.lc_dense_frontend_select_workspace_bytes(B, H, W, int(top), int(left), int(sample))), split)
        rc = lib.lc_dense_frontend_select3(_lib.ptr(xyz), _lib.ptr(wl), _lib.ptr(ws), _lib.ptr(ns), _lib.ptr(vl), float(seg_thresh), code, xcode, _lib.MAP_DTYPES[ws.dtype], xs, wls, vs,
                                           B, H, W, int(top), int(left), int(sample), SELECT_MODES[mode], float(quantile), int(square_weights),
                                           int(min_count), int(seed) & 0xFFFFFFFF, int(pose_index_offset), _lib.ptr(o_u), _lib.ptr(o_w),
                                           _lib.ptr(o_x) if xyz is not None else None, _lib.ptr(o_i), _lib.ptr(o_c), _lib.ptr(work),
                                           0 if work is None else work.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "lc_dense_frontend_select3")
    return o_u, o_w, o_x, o_c, o_i


@torch.no_grad()
def dense_select(pts2d: Tensor, inv_std2d: Tensor, pts3d: Tensor, mode: str, *, mask: Tensor = None, quantile: float = 0.0,
                 counts: Tensor = None, index: Tensor = None, square_weights: bool = True, min_count: int = 4, seed: int = 0, out=None):
    """Test-time point selection (`test.py:39-45,94-113`) as ONE launch: (B,N,.) rows -> survivors compacted to the front of
    padded (B,N,.) rows + `counts` (B,) int32 + their source indices (B,N) int32.  No host synchronisation, no ragged lists:
    the results go straight into `gpu_solver.solve_device(..., n_points=counts)` / `cer_solver.solve(..., n_points=counts)`.

    mode: 'mask' (keep mask), 'quantile' (keep summed weight >= per-sample quantile), 'quantile_in_mask' (the quantile is
    rescaled by the visible fraction and applied inside the mask).  `counts`/`index` describe an already compacted input
    (second-stage selection, e.g. by the RANSAC inlier mask).  Returns (pts2d, weights, pts3d, counts, index);
    weights = inv_std2d**2 when `square_weights` (the solver's inverse covariance, `test.py:92`).
    out: optional (pts2d, weights, pts3d, counts, index) buffers of those shapes to write into (e.g. halves of a larger batch)."""
    lib = _lib.load()
    U = _lib.require_hip_f32("pts2d", pts2d)
    S = _lib.require_hip_f32("inv_std2d", inv_std2d)
    X = _lib.require_hip_f32("pts3d", pts3d)
    B, N = U.shape[:2]
    dev = U.device
    m = None
    if mask is not None:
        m = mask.to(device=dev).reshape(B, N).contiguous()
        if m.dtype == torch.bool:
            m = m.view(torch.uint8)  # same bytes (0 / 1): no launch
        elif m.dtype != torch.uint8:
            m = (m != 0).view(torch.uint8)
    cnt_in = None if counts is None else counts.to(device=dev, dtype=torch.int32).contiguous()
    idx_in = None if index is None else index.to(device=dev, dtype=torch.int32).contiguous()
    if out is not None:
        o_u, o_w, o_x, o_c, o_i = out
        if not (o_u.shape == U.shape and o_w.shape == S.shape and o_x.shape == X.shape and o_c.shape == (B,) and o_i.shape == (B, N)
                and o_u.dtype == o_w.dtype == o_x.dtype == torch.float32 and o_c.dtype == o_i.dtype == torch.int32
                and all(t.is_contiguous() and t.device == dev for t in out)):
            raise ValueError("dense_select: `out` buffers must be contiguous tensors of the result shapes on the input's device")
    else:
        o_u, o_w, o_x = torch.empty_like(U), torch.empty_like(S), torch.empty_like(X)
        o_i = torch.empty(B, N, device=dev, dtype=torch.int32)
        o_c = torch.empty(B, device=dev, dtype=torch.int32)
    with _lib.on_device(dev):
        rc = lib.lc_dense_select_f32(_lib.ptr(U), _lib.ptr(S), _lib.ptr(X), _lib.ptr(m), _lib.ptr(cnt_in), _lib.ptr(idx_in), B, N,
                                     SELECT_MODES[mode], float(quantile), int(square_weights), int(min_count), int(seed) & 0xFFFFFFFF,
                                     _lib.ptr(o_u), _lib.ptr(o_w), _lib.ptr(o_x), _lib.ptr(o_i), _lib.ptr(o_c), _lib.stream_ptr(dev))
    _lib.check(rc, "lc_dense_select_f32")
    return o_u, o_w, o_x, o_c, o_i
