"""The dense heads' auxiliary losses of `Loss_fn.forward` (`losses.py:281-316`) as ONE HIP launch each way
(`lc_amd/csrc/lc_dense_aux.hip`): loss_noc (L1 of the masked xyz head), loss_seg (visibility mask) and, during the pose loss's
warm-up, loss_weight_seg (the weight logits against the visibility mask) -- ~25 element-wise / reduction torch launches and their
autograd twins in the reference."""
from __future__ import annotations

import torch
from torch import Tensor

from . import _lib

SEG_TYPES = {"bce": 0, "l1": 1}
ARRIVAL_WORDS = 544  # include/lc_amd.h LC_ARRIVAL_WORDS: the forward kernels' sharded arrival counters
_WS = {}  # (device index, stream) -> (partials, ticket): the forward's reduction workspace (launches of one stream are ordered)


def _workspace(dev):
    if torch.cuda.is_current_stream_capturing():
        # a graph owns its workspace; only the arrival counters have to start from zero (their zero-fill is a node of the graph), the
        # partials are written before they are read
        return torch.empty(3 * 4096, device=dev, dtype=torch.float64), torch.zeros(ARRIVAL_WORDS, device=dev, dtype=torch.int32)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), _lib.raw_stream(dev))
    ws = _WS.get(key)
    if ws is None:
        if len(_WS) >= 64:
            _WS.clear()
        ws = _WS[key] = (torch.zeros(3 * 4096, device=dev, dtype=torch.float64), torch.zeros(ARRIVAL_WORDS, device=dev, dtype=torch.int32))
    return ws


def fused_path_ok(*tensors) -> bool:
    """Maps on the GPU take the fused launch -- fp32, or the 16-bit types of a mixed-precision backbone, read natively (`_lib.hip_maps`)."""
    return all(t is None or (t.is_cuda and t.dtype in _lib.MAP_DTYPES and t.numel() > 0) for t in tensors)


class _DenseAux(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, msk_noc, noc_tgt, seg_logits, msk_vis, wlogits, seg_type: int):
        lib = _lib.load()
        # the network's outputs in their own element type and layout (fp32 / fp16 / bf16, dense or channel slices: `_lib.hip_maps`);
        # targets and masks are labels: fp32 / bytes
        (x, seg, w), (xs, ss, wls), code = _lib.hip_maps(xyz_noc=xyz, msk_vis_logits=seg_logits, xyz_weight_logits=wlogits)
        B = seg.shape[0]
        HW = seg.numel() // B
        dev = seg.device
        vis = _lib.require_hip_f32("msk_vis", msk_vis.reshape(B, HW))
        tgt = m8 = mf = None
        if xyz is not None:
            tgt = _lib.require_hip_f32("xyz_noc_tgt", noc_tgt)
            m = msk_noc.reshape(B, HW).contiguous()
            if m.dtype == torch.bool:
                m8 = m.view(torch.uint8)
            elif m.dtype == torch.uint8:
                m8 = m
            else:
                mf = _lib.require_hip_f32("msk_noc", m)
        losses = torch.empty(3, device=dev, dtype=torch.float32)
        partials, ticket = _workspace(dev)
        P = _lib.ptr
        with _lib.on_device(dev):
            rc = lib.lc_dense_aux_fwd2(P(x), P(m8), P(mf), P(tgt), P(seg), P(vis), P(w), code, xs, ss, wls, B, HW, int(seg_type), P(losses),
                                       P(partials), P(ticket), _lib.stream_ptr(dev))
        _lib.check(rc, "lc_dense_aux_fwd2")
        ctx.map_args = (code, xs, ss, wls)
        ctx.save_for_backward(*(t for t in (x, m8, mf, tgt, seg, vis, w) if t is not None))
        ctx.have = tuple(t is not None for t in (x, m8, mf, tgt, seg, vis, w))
        ctx.seg_type, ctx.shapes = int(seg_type), (None if xyz is None else xyz.shape, seg_logits.shape, None if wlogits is None else wlogits.shape)
        l0, l1, l2 = losses.unbind(0)
        return l0, l1, l2

    @staticmethod
    def backward(ctx, g0, g1, g2):
        lib = _lib.load()
        it = iter(ctx.saved_tensors)
        x, m8, mf, tgt, seg, vis, w = (next(it) if h else None for h in ctx.have)
        B = seg.shape[0]
        HW = seg.numel() // B
        dev = seg.device
        need_x, _, _, need_seg, _, need_w, _ = ctx.needs_input_grad
        dense = lambda t: torch.empty(t.shape, device=dev, dtype=t.dtype)  # noqa: E731  (gradient maps: dense, in the map's own type)
        d_x = dense(x) if (need_x and x is not None) else None
        d_s = dense(seg) if need_seg else None
        d_w = dense(w) if (need_w and w is not None) else None
        gs = [None if g is None else g.to(dtype=torch.float32).contiguous() for g in (g0, g1, g2)]
        code, xs, ss, wls = ctx.map_args
        P = _lib.ptr
        with _lib.on_device(dev):
            rc = lib.lc_dense_aux_bwd2(P(x), P(m8), P(mf), P(tgt), P(seg), P(vis), P(w), code, xs, ss, wls, B, HW, ctx.seg_type, P(gs[0]), P(gs[1]),
                                       P(gs[2]), P(d_x), P(d_s), P(d_w), _lib.stream_ptr(dev))
        _lib.check(rc, "lc_dense_aux_bwd2")
        sx, ss, sw = ctx.shapes
        return (None if d_x is None else d_x.view(sx), None, None, None if d_s is None else d_s.view(ss), None,
                None if d_w is None else d_w.view(sw), None)


def dense_aux_losses(xyz_noc: Tensor, msk_noc: Tensor, xyz_noc_tgt: Tensor, msk_vis_logits: Tensor, msk_vis: Tensor,
                     xyz_weight_logits: Tensor = None, seg_loss_type: str = "bce"):
    """-> (loss_noc, loss_seg, loss_weight_seg) as 0-dim tensors (the first / last are zero constants when their input is None):
    `F.l1_loss(xyz_noc * msk_noc[:, None], xyz_noc_tgt)`, `seg(msk_vis_logits, msk_vis[:, None])` and
    `seg(xyz_weight_logits, msk_vis[:, None].expand_as(xyz_weight_logits))`, seg = BCE-with-logits or `Loss_seg_L1`."""
    return _DenseAux.apply(xyz_noc, msk_noc, xyz_noc_tgt, msk_vis_logits, msk_vis, xyz_weight_logits, SEG_TYPES[seg_loss_type.lower()])


class _XyzBinLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, gt_bits, msk_vis_logits, histogram, momentum: float, group=None):
        lib = _lib.load()
        B, C = logits.shape[:2]
        HW = logits.numel() // (B * C)
        (x, v), (ls, vs), code = _lib.hip_maps(xyz_noc_bin=logits, msk_vis_logits=msk_vis_logits.reshape(B, HW))  # a (B,1,H,W) channel slice reshapes without a copy
        dev = x.device
        t = gt_bits.contiguous()
        t = t.view(torch.uint8) if t.dtype == torch.bool else (t if t.dtype == torch.uint8 else (t != 0).view(torch.uint8))
        if not (histogram.is_cuda and histogram.dtype == torch.float32 and histogram.is_contiguous() and histogram.numel() == C):
            raise ValueError("Loss_xyz_bin: the histogram buffer is a contiguous float32 tensor of one entry per code bit on the GPU")
        loss = torch.empty(1, device=dev, dtype=torch.float32)
        weights = torch.empty(C, device=dev, dtype=torch.float32)
        partials, ticket = _workspace(dev)
        if partials.numel() < C * 32 * 3:
            raise ValueError("Loss_xyz_bin: more than 128 code bits")
        P = _lib.ptr
        if group is None:
            with _lib.on_device(dev):
                rc = lib.lc_xyz_bin_loss_fwd2(P(x), P(t), P(v), code, ls, vs, B, C, HW, float(momentum), P(histogram), P(loss), P(weights), P(partials),
                                              P(ticket), _lib.stream_ptr(dev))
            _lib.check(rc, "lc_xyz_bin_loss_fwd2")
        else:
            # The batch is sharded over `group`: the same pass stops after the counts, the C + 1 integers are summed over the ranks (int64: exact
            # at any batch size, as the reference's integer sums are -- losses.py:203-204), and a one-workgroup launch closes with the arithmetic
            # the one-launch kernel ends with.  Two launches + one latency-sized all-reduce instead of ~15 element-wise torch launches.
            import torch.distributed as dist

            counts = torch.empty(C + 1, device=dev, dtype=torch.int64)
            bce_mean = torch.empty(C, device=dev, dtype=torch.float32)
            with _lib.on_device(dev):
                rc = lib.lc_xyz_bin_loss_counts(P(x), P(t), P(v), code, ls, vs, B, C, HW, P(counts), P(bce_mean), P(partials), P(ticket),
                                                _lib.stream_ptr(dev))
            _lib.check(rc, "lc_xyz_bin_loss_counts")
            dist.all_reduce(counts, group=group)
            with _lib.on_device(dev):
                rc = lib.lc_xyz_bin_loss_finish(P(counts), P(bce_mean), C, float(momentum), P(histogram), P(loss), P(weights), _lib.stream_ptr(dev))
            _lib.check(rc, "lc_xyz_bin_loss_finish")
        ctx.map_args = (code, ls, vs)
        ctx.save_for_backward(x, t, v, weights)
        ctx.shape = logits.shape
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, t, v, weights = ctx.saved_tensors
        B, C = x.shape[:2]
        HW = x.numel() // (B * C)
        d = torch.empty(x.shape, device=x.device, dtype=x.dtype)  # dense, in the logits' own type
        g = g.to(dtype=torch.float32).contiguous()
        code, ls, vs = ctx.map_args
        P = _lib.ptr
        with _lib.on_device(x.device):
            rc = lib.lc_xyz_bin_loss_bwd2(P(x), P(t), P(v), P(weights), P(g), code, ls, vs, B, C, HW, P(d), _lib.stream_ptr(x.device))
        _lib.check(rc, "lc_xyz_bin_loss_bwd2")
        return d.view(ctx.shape), None, None, None, None, None


def xyz_bin_loss(noc_xyz_bin_logits: Tensor, noc_xyz_bin_gt: Tensor, msk_vis_logits: Tensor, histogram: Tensor, momentum: float, group=None) -> Tensor:
    """`Loss_xyz_bin.forward` (`losses.py:196-216`) as one launch each way; `histogram` (the module's buffer) is updated in place.
    `group`: the batch is sharded over this process group -- the histogram update takes the whole batch's counts (counts launch, int64
    all-reduce, finish launch); the returned loss is this rank's (DistributedDataParallel averages the ranks)."""
    return _XyzBinLoss.apply(noc_xyz_bin_logits, noc_xyz_bin_gt, msk_vis_logits, histogram, momentum, group)
