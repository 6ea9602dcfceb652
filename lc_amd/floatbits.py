"""`floatbits.py` call surface (ZebraPose-style binary surface codes) on the HIP decode kernels.

    nn_logits2noc_with_gt(logits, gt_raw_bits, bit_cnt, gt_msk)   <- floatbits.py:50-72  (training, differentiable)
    nn_logits2noc(logits, bit_cnt)                                <- floatbits.py:35-48  (inference)
    nn_noc2target(noc, bit_cnt)                                   <- floatbits.py:13-33  (label prep, no grad, plain torch)
Tensors keep the network layout (B,C,H,W); outputs are (B,H,W,3) like the reference.  `nearest_lut` is not supported
(no reference call site passes it)."""
from __future__ import annotations

import math
from collections import abc
from typing import List, Union

import torch
from torch import Tensor

from . import _lib

_black_background = True


def set_black_background(black=True):
    global _black_background
    _black_background = black


def _bits3(bit_cnt, C):
    if isinstance(bit_cnt, abc.Sequence):
        bits = [int(b) for b in bit_cnt]
    else:
        bits = [int(bit_cnt)] * 3
    if len(bits) != 3 or sum(bits) != C:
        raise ValueError(f"lc_amd.floatbits: bit_cnt {bit_cnt} does not match {C} code channels")
    return bits


def _as_u8(name, t):
    if not t.is_cuda:
        raise RuntimeError(f"lc_amd: {name} is on {t.device}; the HIP path needs tensors on the MI355X (there is no CPU fallback "
                           f"in the product path)")
    if t.dtype == torch.bool:
        return t.contiguous().view(torch.uint8)  # same bytes (0 / 1): no launch
    return (t != 0).to(torch.uint8).contiguous() if t.dtype != torch.uint8 else t.contiguous()


def _code_logits(logits):
    """The code logits as the kernels take them: fp32 / fp16 / bf16, in place wherever every sample is contiguous (`_lib.hip_maps`)."""
    (lg,), (bs,), code = _lib.hip_maps(logits=logits)
    return lg, bs, code


def _launch_decode_gt(logits, gt_bits, gt_msk, bits, top, left, sample, out_scale=None, out_xform=None):
    lib = _lib.load()
    lg, bs, code = _code_logits(logits)
    B, C, H, W = lg.shape
    N = ((H - top + sample - 1) // sample) * ((W - left + sample - 1) // sample)
    out = torch.empty(B, N, 3, device=lg.device, dtype=torch.float32)
    with _lib.on_device(lg.device):
        rc = lib.lc_bits_decode_gt_fwd3(_lib.ptr(lg), _lib.ptr(gt_bits), _lib.ptr(gt_msk), _lib.ptr(out_scale), _lib.ptr(out_xform), code, bs,
                                        B, C, H, W, *bits, int(_black_background), top, left, sample, _lib.ptr(out),
                                        _lib.stream_ptr(lg.device))
    _lib.check(rc, "lc_bits_decode_gt_fwd3")
    return out


def _launch_decode_gt_bwd(logits, gt_bits, gt_msk, g_out, bits, top, left, sample, black, out_scale=None, out_xform=None):
    lib = _lib.load()
    lg, bs, code = _code_logits(logits)
    B, C, H, W = lg.shape
    d = torch.empty(B, C, H, W, device=lg.device, dtype=lg.dtype)  # the gradient of a map in the map's own type, dense
    with _lib.on_device(lg.device):
        rc = lib.lc_bits_decode_gt_bwd3(_lib.ptr(lg), _lib.ptr(gt_bits), _lib.ptr(gt_msk), _lib.ptr(out_scale), _lib.ptr(out_xform),
                                        _lib.ptr(g_out), code, bs, B, C, H, W, *bits, int(black), top, left, sample, _lib.ptr(d),
                                        _lib.stream_ptr(lg.device))
    _lib.check(rc, "lc_bits_decode_gt_bwd3")
    return d


class _DecodeGtFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, gt_bits, gt_msk, bits, top, left, sample, out_scale=None, out_xform=None):
        out = _launch_decode_gt(logits, gt_bits, gt_msk, bits, top, left, sample, out_scale, out_xform)
        ctx.save_for_backward(logits, gt_bits, gt_msk, out_scale, out_xform)
        ctx.cfg = (tuple(bits), top, left, sample, _black_background)
        return out

    @staticmethod
    def backward(ctx, g):
        logits, gt_bits, gt_msk, out_scale, out_xform = ctx.saved_tensors
        bits, top, left, sample, black = ctx.cfg
        d = _launch_decode_gt_bwd(logits, gt_bits, gt_msk, g.contiguous().to(torch.float32), list(bits), top, left, sample, black, out_scale,
                                  out_xform)
        return d, None, None, None, None, None, None, None, None


def decode_with_gt_strided(logits: Tensor, gt_raw_bits: Tensor, bit_cnt, gt_msk: Tensor, sample: int = 1, top_left=(0, 0), *,
                           out_scale: Tensor = None, out_xform: Tensor = None) -> Tensor:
    """(B,C,H,W) logits -> (B,N,3) normalised coordinates of the strided pixel subset (losses.py:163-184 order:
    sub-sample first, decode second).  out_scale (B,3) / out_xform (B,4,4): the callers' `noc * noc_scale` and
    `(xyz - T[:, :3, 3]) @ T[:, :3, :3]` (losses.py:17-47) applied by the same launch (and undone by the backward launch); neither
    takes a gradient."""
    lg = _code_logits(logits)[0]  # validated here; consumed in its own element type and layout (no `.float()`, no `.contiguous()`)
    bits = _bits3(bit_cnt, lg.shape[1])
    gb = _as_u8("gt_raw_bits", gt_raw_bits)
    gm = None if gt_msk is None else _as_u8("gt_msk", gt_msk)
    sc = None if out_scale is None else _lib.require_hip_f32("out_scale", out_scale.detach().reshape(lg.shape[0], 3))
    xf = None if out_xform is None else _lib.require_hip_f32("out_xform", out_xform.detach().reshape(lg.shape[0], 4, 4))
    return _DecodeGtFn.apply(lg, gb, gm, bits, int(top_left[0]), int(top_left[1]), int(sample), sc, xf)


def nn_logits2noc_with_gt(logits: Tensor, gt_raw_bits: Tensor, bit_cnt: Union[int, List[int]], gt_msk: Tensor) -> Tensor:
    """floatbits.py:50-72: logits, gt_raw_bits (B,C,H,W), gt_msk (B,H,W) -> noc (B,H,W,3), differentiable w.r.t. logits."""
    B, _, H, W = logits.shape
    return decode_with_gt_strided(logits, gt_raw_bits, bit_cnt, gt_msk).reshape(B, H, W, 3)


@torch.no_grad()
def nn_logits2noc(logits: Tensor, bit_cnt: Union[int, List[int]], nearest_lut: Tensor = None) -> Tensor:
    """floatbits.py:35-48 (inference decode): (B,C,H,W) -> (B,H,W,3)."""
    if nearest_lut is not None:
        raise NotImplementedError("lc_amd.floatbits: nearest_lut is not used by any reference call site")
    lib = _lib.load()
    lg, bs, code = _code_logits(logits)
    B, C, H, W = lg.shape
    bits = _bits3(bit_cnt, C)
    noc = torch.empty(B, H, W, 3, device=lg.device, dtype=torch.float32)
    with _lib.on_device(lg.device):
        rc = lib.lc_bits_decode3(_lib.ptr(lg), None, None, code, bs, B, C, H, W, *bits, int(_black_background), 0, _lib.ptr(noc),
                                 _lib.stream_ptr(lg.device))
    _lib.check(rc, "lc_bits_decode3")
    return noc


@torch.no_grad()
def nn_logits2xyz_planes(logits: Tensor, bit_cnt: Union[int, List[int]], noc_scale: Tensor = None, model_transform: Tensor = None) -> Tensor:
    """Inference decode straight to object coordinates as (B,3,H,W) planes -- `nn_out_to_xyz(..., inference=True).permute(0, 3, 1, 2)`
    (losses.py:17-47) in one launch: Gray decode, `* noc_scale`, `(. - T[:, :3, 3]) @ T[:, :3, :3]`, channel-first layout (what the
    dense front end reads)."""
    lib = _lib.load()
    lg, bs, code = _code_logits(logits)
    B, C, H, W = lg.shape
    bits = _bits3(bit_cnt, C)
    sc = None if noc_scale is None else _lib.require_hip_f32("noc_scale", noc_scale.reshape(B, 3))
    xf = None if model_transform is None else _lib.require_hip_f32("model_transform", model_transform.reshape(B, 4, 4))
    out = torch.empty(B, 3, H, W, device=lg.device, dtype=torch.float32)
    with _lib.on_device(lg.device):
        rc = lib.lc_bits_decode3(_lib.ptr(lg), _lib.ptr(sc), _lib.ptr(xf), code, bs, B, C, H, W, *bits, int(_black_background), 1, _lib.ptr(out),
                                 _lib.stream_ptr(lg.device))
    _lib.check(rc, "lc_bits_decode3")
    return out


@torch.no_grad()
def decode_selected_rows(logits: Tensor, bit_cnt: Union[int, List[int]], index: Tensor, counts: Tensor, out_pts3d: Tensor, *, noc_scale: Tensor = None,
                         model_transform: Tensor = None, sample: int = 1, top_left=(0, 0)) -> Tensor:
    """Inference decode of the SELECTED pixels only: fills out_pts3d[b, :counts[b]] (B,N,3) with the object coordinates of the sampled pixels
    index[b, :counts[b]] (the `index` / `counts` a point selection returns) -- `nn_logits2xyz_planes(...)` gathered at those pixels, without
    decoding (writing, re-reading) the other four fifths of the map."""
    lib = _lib.load()
    lg, bs, code = _code_logits(logits)
    B, C, H, W = lg.shape
    bits = _bits3(bit_cnt, C)
    sc = None if noc_scale is None else _lib.require_hip_f32("noc_scale", noc_scale.reshape(B, 3))
    xf = None if model_transform is None else _lib.require_hip_f32("model_transform", model_transform.reshape(B, 4, 4))
    N = index.shape[1]
    if not (index.dtype == counts.dtype == torch.int32 and index.is_contiguous() and counts.is_contiguous() and index.shape == (B, N)
            and out_pts3d.shape == (B, N, 3) and out_pts3d.dtype == torch.float32 and out_pts3d.is_contiguous()
            and index.device == counts.device == out_pts3d.device == lg.device):
        raise ValueError("decode_selected_rows: index (B,N) / counts (B,) int32 and out_pts3d (B,N,3) float32, contiguous, on the logits' device")
    with _lib.on_device(lg.device):
        rc = lib.lc_bits_decode_rows(_lib.ptr(lg), _lib.ptr(sc), _lib.ptr(xf), code, bs, B, C, H, W, *bits, int(_black_background), int(top_left[0]),
                                     int(top_left[1]), int(sample), _lib.ptr(index), _lib.ptr(counts), N, _lib.ptr(out_pts3d), _lib.stream_ptr(lg.device))
    _lib.check(rc, "lc_bits_decode_rows")
    return out_pts3d


@torch.no_grad()
def mod_noc2bits_bb(numbers: Tensor, N: int, black_background=True):
    """floatbits.py:77-97 (label prep): normalised coordinate (-1,1) -> Gray-coded bits + raw bits, (*,N) bool."""
    max_num = 2 ** N - 1
    ints = torch.clamp((numbers + 1) * (max_num * 0.5), 0, max_num).round().to(torch.int32)
    mask = 2 ** torch.arange(N - 1, -1, -1, dtype=torch.int32, device=numbers.device)
    bits = ints.unsqueeze(-1).bitwise_and(mask).bool()
    mod_bits = bits.clone()
    mod_bits[..., 1:] = mod_bits[..., 1:].logical_xor(bits[..., 0:-1])
    if black_background:
        mod_bits[..., 0:2] = mod_bits[..., 0:2].logical_not()
    return mod_bits, bits


@torch.no_grad()
def nn_noc2target(noc: Tensor, bit_cnt: Union[int, List[int]]):
    """floatbits.py:13-33: noc (B,H,W,3) -> (mod_bits, raw_bits), each (B,C,H,W) bool."""
    bits = [int(b) for b in bit_cnt] if isinstance(bit_cnt, abc.Sequence) else [int(bit_cnt)] * 3
    outs = [mod_noc2bits_bb(noc[..., a], n, _black_background) for a, n in enumerate(bits)]
    mod = torch.cat([o[0] for o in outs], dim=-1).permute(0, 3, 1, 2)
    raw = torch.cat([o[1] for o in outs], dim=-1).permute(0, 3, 1, 2)
    return mod, raw


def calc_bit_count(sizes, max_bits=7, min_bits=2):
    """floatbits.py:259-263."""
    max_size = max(sizes)
    return [max(min_bits, round(max_bits + math.log2(size / max_size))) for size in sizes]
