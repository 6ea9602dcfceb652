"""Test-time pose recovery (the job of `test.py:47-136` `solve_pnp` / `solve_pnp_dense`) with every stage on the GPU and no
host synchronisation between them:

    network output -> [code decode] -> dense front end (joint softmax x scale, stride sub-sampling)      lc_dense.hip
                   -> point selection + compaction into padded lists with device-side counts            lc_select.hip
                   -> RANSAC-P3P initialiser (in place of cv2.solvePnPRansac)                           lc_pnp_init.hip
                   -> weighted LM from that start                                                        lc_pnp_body.h
                   -> [second selection by the RANSAC inliers -> weighted LM again: 'weighted-filtered']

The reference walks Python lists here (one `nonzero()` = one device sync per sample, `np.random` padding, per-sample
re-batching, a multiprocessing pool around OpenCV); this module keeps one padded batch and an int32 `counts` vector on the
device from the first kernel to the last.  The returned dict has the reference's keys ('weighted', 'ransac',
'weighted-filtered'), each a (B,7) tensor of `w,x,y,z,tx,ty,tz`.
"""
from __future__ import annotations

import contextlib
import gc

import torch
from torch import Tensor

from . import floatbits, splitws
from .dense import FUSED_SELECT_MAX_POINTS, dense_front_end_select, dense_front_end_with_visibility, dense_select
from .losses import nn_out_to_xyz
from .pnp import gpu_solver, pnp_ceres


def quantile_msk(den_inv_std2d: Tensor, quantile) -> Tensor:
    """`test.py:39-45` as a mask (kept for callers that want the mask itself; the pipeline below uses the fused
    `dense_select`).  `quantile` is a float or a per-sample (B,) tensor."""
    w = den_inv_std2d.sum(dim=-1)
    if isinstance(quantile, Tensor):
        order = w.sort(dim=1).values
        rank = quantile.to(w.dtype).clamp(0, 1) * (w.shape[1] - 1)
        lo, hi = rank.floor().long(), rank.ceil().long()
        thr = torch.lerp(order.gather(1, lo[:, None]), order.gather(1, hi[:, None]), (rank - lo)[:, None])
    else:
        thr = torch.quantile(w, quantile, dim=1, keepdim=True)
    return w >= thr


def _reprojection_threshold(cfg, gt_dict, default_px):
    """-> keyword arguments of `gpu_solver.solve_device`; rel_reproj_err (test.py:56-57,115-116): 2 / out_pix_scale, divided by the kernel."""
    if cfg.get("rel_reproj_err", False):
        return dict(reprojectionError=2.0, reproj_divisor=gt_dict["out_pix_scale"])
    return dict(reprojectionError=default_px)


def _weighted(K, pts3d, pts2d, icov, start, counts=None):
    """`cer_solver.solve(..., filter_input_nan=True)[1]` (test.py:59,120,133) without the flag tensors nobody reads here: NaN filter,
    square root of the inverse variances and the fall-back to `start` all happen inside the one solver launch."""
    return pnp_ceres.solve_device(K, pts3d, pts2d, icov, start, counts, weights_are_icov=True, nan_to_num=True)[0]


@torch.no_grad()
def solve_pnp(cfg, out_dict, gt_dict):
    """Sparse heads (`test.py:47-64`): keypoints + predicted std -> RANSAC start -> weighted solve."""
    if "pts2d" not in out_dict:
        return solve_pnp_dense(cfg, out_dict, gt_dict)
    K, pts3d = gt_dict["out_K"], gt_dict["pts3d"]
    pts2d, std = out_dict["pts2d"], out_dict["pts2d_std"]
    # RANSAC, then its inlier refinement and the weighted solve on all keypoints as ONE launch (`lc_pnp_lm_chain2_f32`); `1 / std**2` (test.py:52), the NaN
    # filter and the square root of cer_solver.py:29-36 are formed at the solve's loads, not by element-wise launches in front of it
    _ransac, _inl, _bad, refine = gpu_solver.solve_device(K, pts3d, pts2d, refine="defer", **_reprojection_threshold(cfg, gt_dict, 2))
    (start, _, _), (weighted, _, _) = pnp_ceres.solve_chain_device(
        refine, dict(cam_mat=K, pts3d=pts3d, pts2d=pts2d, sqrtL=std, weights_are_std=True, nan_to_num=True, start="first"))
    return {"weighted": weighted, "ransac": start}


_SIDE_STREAMS = {}  # device index -> streams the sub-batches of a wide test-time batch run on


def _sub_batch_count(B: int, N: int) -> int:
    """How many sub-batches `solve_pnp_dense` cuts a batch into: 1 unless LC_AMD_TEST_TIME_STREAMS says otherwise.
    At zlmo's test-time shape (16 384 candidates per object) five of the chain's seven launches are one workgroup per object -- 64 objects keep a
    quarter of the chip busy -- and the other two (code decode, RANSAC scoring) fill it.  Cut into sub-batches on side streams the narrow
    launches of one sub-batch COULD run beside the wide ones of another; results do not change (every object is independent, and the RANSAC's
    hypothesis streams / padding draws are keyed by the object's index in the WHOLE batch: `pose_index_offset`,
    tests/test_gpu_test_time.py::test_sub_batches_on_streams_return_the_one_batch_result).  MEASURED (round 4, one MI355X, 64 objects, four
    sub-batches): the replayed graph takes 254 us against 225 us for the one batch -- the runtime does not overlap the graph's parallel
    branches enough to pay for 4 x 7 smaller launches -- and the eager call is host-bound (894 us).  Hence off by default; the switch stays for
    callers whose batches are larger than one round of the chip."""
    import os

    forced = os.environ.get("LC_AMD_TEST_TIME_STREAMS")
    return max(1, min(int(forced), B)) if forced else 1


@torch.no_grad()
def solve_pnp_dense(cfg, out_dict, gt_dict):
    """Dense heads (`test.py:67-136`)."""
    B, _, H, W = out_dict["xyz_weight_logits"].shape
    stride = cfg.get("dense_sample", 2)
    parts = _sub_batch_count(B, -(-H // stride) * -(-W // stride))
    if parts <= 1:
        return _solve_pnp_dense(cfg, out_dict, gt_dict, 0)
    dev = out_dict["xyz_weight_logits"].device
    cur = torch.cuda.current_stream(dev)
    pool = _SIDE_STREAMS.setdefault(dev.index if dev.index is not None else torch.cuda.current_device(), [])
    while len(pool) < parts:
        pool.append(torch.cuda.Stream(dev))
    cut = lambda d, b0, b1: {k: (v[b0:b1] if isinstance(v, Tensor) and v.dim() > 0 and v.shape[0] == B else v) for k, v in d.items()}  # noqa: E731
    bounds = [(B * k // parts, B * (k + 1) // parts) for k in range(parts)]
    results = []
    for (b0, b1), side in zip(bounds, pool):
        side.wait_stream(cur)  # the network's outputs are ready on the caller's stream
        with torch.cuda.stream(side), splitws.no_split():  # concurrent launches: forms whose workgroups wait for each other would stay correct (rescue launch) but crawl
            results.append(_solve_pnp_dense(cfg, cut(out_dict, b0, b1), cut(gt_dict, b0, b1), b0))
    for side in pool[:parts]:
        cur.wait_stream(side)
    for res in results:  # allocated on the side streams, read from here on by the caller's
        for t in res.values():
            t.record_stream(cur)
    return {k: torch.cat([res[k] for res in results]) for k in results[0]}


def _solve_pnp_dense(cfg, out_dict, gt_dict, pose0):
    K = gt_dict["out_K"]
    stride = cfg.get("dense_sample", 2)
    thr = cfg.get("seg_thresh", 0.5)
    mode = cfg.dense_point_select
    if mode not in ("mask", "quantile", "quantile_in_mask"):
        raise ValueError(f"unknown dense_point_select {mode!r}")
    if "xyz_noc" in out_dict and gt_dict.get("model_transform", None) is None and gt_dict.get("bit_cnt", None) is None:
        # continuous head: the kernel scales the normalised coordinates itself (losses.py:17-22 is that one multiply)
        xyz_map, noc_scale = out_dict["xyz_noc"], gt_dict["noc_scale"]
    elif "xyz_noc_bin" in out_dict:
        # binary-code head (fp32, fp16 or bf16 logits, read in their own type).  Up to FUSED_SELECT_MAX_POINTS candidates per object the
        # selection runs WITHOUT model points and the codes of the selected pixels only are decoded afterwards (Gray decode, noc_scale, model
        # transform: `decode_selected_rows` -- a fifth of the whole-map decode's work at zlmo's shape); beyond that, the whole map is decoded
        # into the (B,3,H,W) planes the two-launch front end reads.
        _, _, H_, W_ = out_dict["xyz_weight_logits"].shape
        if -(-H_ // stride) * -(-W_ // stride) <= FUSED_SELECT_MAX_POINTS:
            xyz_map = None
        else:
            xyz_map = floatbits.nn_logits2xyz_planes(out_dict["xyz_noc_bin"], gt_dict["bit_cnt"], gt_dict["noc_scale"],
                                                     gt_dict.get("model_transform", None))
        noc_scale = None
    else:
        head = out_dict["xyz_noc"] if "xyz_noc" in out_dict else out_dict["xyz_noc_bin"]
        xyz_map = nn_out_to_xyz(head, gt_dict["noc_scale"], model_transform=gt_dict.get("model_transform", None),
                                bit_cnt=gt_dict.get("bit_cnt", None), inference=True).permute(0, 3, 1, 2)  # (B,H,W,3), object frame
        noc_scale = None

    # survivors compacted to the front of each row; icov = inv_std^2 (test.py:92); counts stay on the device.
    # 'weighted' and 'weighted-filtered' solve the same objects from the same start on two selections: when both are wanted the two
    # selections are written into the halves of ONE (2B, N, .) batch and solved by ONE launch (K and start shared through
    # `shared_poses`) -- 64 objects fill a quarter of the chip, the two solves side by side cost what one does.
    wanted = cfg.solvers
    both = "weighted_filtered" in wanted and "weighted" in wanted
    B, _, H, W = out_dict["xyz_weight_logits"].shape
    N = -(-H // stride) * -(-W // stride)
    rows = 2 * B if both else B
    dev = out_dict["xyz_weight_logits"].device
    f32, i32 = dict(device=dev, dtype=torch.float32), dict(device=dev, dtype=torch.int32)
    U2, W2, X2 = torch.empty(rows, N, 2, **f32), torch.empty(rows, N, 2, **f32), torch.empty(rows, N, 3, **f32)
    C2, I2 = torch.empty(rows, **i32), torch.empty(rows, N, **i32)
    halves = list(zip(*(t.chunk(2) if both else (t,) for t in (U2, W2, X2, C2, I2))))  # [(u, w, x, counts, index) of each half]
    half = halves.__getitem__
    select_args = dict(quantile=float(cfg.get("quantile", 0.0)), square_weights=True, min_count=4, out=half(0))
    if N <= FUSED_SELECT_MAX_POINTS:
        # joint softmax x scale, the (0,0)-phase stride sub-sampling (test.py:85-92), the visibility mask of the sampled pixels
        # (test.py:88-90) AND the point selection (test.py:94-113) in one launch
        u, icov, x, counts, index = dense_front_end_select(xyz_map, out_dict["xyz_weight_logits"], out_dict["xyz_weights_scale"], noc_scale,
                                                           out_dict["msk_vis_logits"], mode, seg_thresh=thr, sample=stride, pose_index_offset=pose0,
                                                           **select_args)
        if xyz_map is None:  # code heads: the model points of the selected pixels (padding entries included: they carry source indices too)
            floatbits.decode_selected_rows(out_dict["xyz_noc_bin"], gt_dict["bit_cnt"], index, counts, x, noc_scale=gt_dict["noc_scale"],
                                           model_transform=gt_dict.get("model_transform", None), sample=stride)
    else:
        pts2d, inv_std, pts3d, visible = dense_front_end_with_visibility(xyz_map, out_dict["xyz_weight_logits"], out_dict["xyz_weights_scale"],
                                                                         noc_scale, out_dict["msk_vis_logits"], thr, sample=stride)
        u, icov, x, counts, index = dense_select(pts2d, inv_std, pts3d, mode, mask=visible, **select_args)
    # test.py:129-133: the selection intersected with the RANSAC inliers -- compacted by the RANSAC's own selection step (the workgroup
    # that writes the inlier mask), not by a `dense_select(..., 'mask', mask=inliers)` launch behind the refinement
    filtered = None
    if "weighted_filtered" in wanted:
        filtered = dict(weights=icov, index=index, min_count=4, out=half(1) if both else None)
    start, inliers, _bad, refine = gpu_solver.solve_device(K, x, u, counts, select=filtered, refine="defer", pose_index_offset=pose0,
                                                           **_reprojection_threshold(cfg, gt_dict, 3))
    # The RANSAC's inlier refinement and the weighted solve(s) that start from its result: ONE launch (`lc_pnp_lm_chain2_f32`), each
    # workgroup refines its object's pose and goes on with its own weighted solve.
    weighted = dict(weights_are_icov=True, nan_to_num=True, start="first")  # `_weighted` above, chained
    out = {}
    if both:  # the two selections side by side: 2B poses
        (start, _, _), (states, _, _) = pnp_ceres.solve_chain_device(
            refine, dict(cam_mat=K, pts3d=X2, pts2d=U2, sqrtL=W2, n_points=C2, shared_poses=B, **weighted))
        out["weighted"], out["weighted-filtered"] = states.chunk(2)
    elif "weighted_filtered" in wanted:
        fu, ficov, fx, fcounts, _ = filtered["result"]
        (start, _, _), (out["weighted-filtered"], _, _) = pnp_ceres.solve_chain_device(
            refine, dict(cam_mat=K, pts3d=fx, pts2d=fu, sqrtL=ficov, n_points=fcounts, **weighted))
    elif "weighted" in wanted:
        (start, _, _), (out["weighted"], _, _) = pnp_ceres.solve_chain_device(
            refine, dict(cam_mat=K, pts3d=x, pts2d=u, sqrtL=icov, n_points=counts, **weighted))
    else:
        start = pnp_ceres.solve_device(**refine)[0]
    if "ransac" in wanted:
        out["ransac"] = start
    return {k: out[k] for k in ("ransac", "weighted-filtered", "weighted") if k in out}  # key order of test.py:129-135


@contextlib.contextmanager
def quiet_capture():
    """No Python garbage collection while a stream is capturing: a collected `CUDAGraph` / graph-pool tensor of an earlier
    capture would run its HIP destructor inside the capture, which the runtime answers with an abort."""
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was_enabled:
            gc.enable()


class GraphedSolvePnP:
    """`solve_pnp` captured once as a hipGraph and replayed (fixed shapes): the pipeline above is a chain of 5 short
    launches with no host synchronisation, so a replay removes the per-launch host cost (64 objects of 64x64 maps: 104 us
    eager -> 64 us replayed on one MI355X, identical results; `scripts/ubench/graph_inference.py`).

        solver = GraphedSolvePnP(cfg, out_dict, gt_dict)      # example inputs fix the shapes; captured on a side stream
        poses = solver(out_dict, gt_dict)                     # copies the tensors into the static buffers, replays

    Every tensor entry of the two dicts is treated as an input; non-tensor entries (bit counts, ...) are frozen at capture.
    """

    def __init__(self, cfg, out_dict, gt_dict, warmup: int = 2):
        self.cfg = cfg
        # static input buffers, contiguous whatever the example's layout: a strided network output would otherwise be copied into
        # shape by a torch launch INSIDE every replay
        own = lambda v: v.detach().clone(memory_format=torch.contiguous_format) if isinstance(v, Tensor) else v  # noqa: E731
        self._out = {k: own(v) for k, v in out_dict.items()}
        self._gt = {k: own(v) for k, v in gt_dict.items()}
        dev = next(v.device for v in self._out.values() if isinstance(v, Tensor))
        if dev.type != "cuda":
            raise RuntimeError("lc_amd: GraphedSolvePnP needs tensors on the MI355X (there is no CPU fallback in the product path)")
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # warm-up off the capture: module loading, allocator pools, LDS attributes
            for _ in range(max(1, warmup)):
                solve_pnp(cfg, self._out, self._gt)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        # the workspaces of the split launches (few objects x thousands of points: selection, solves): zeroed once here, kept consistent by the launches themselves -- no fill node in the graph
        self._split_ws = (torch.zeros(splitws.max_bytes("pnp", dev), device=dev, dtype=torch.uint8), torch.zeros(splitws.max_bytes("select", dev), device=dev, dtype=torch.uint8))
        torch.cuda.synchronize(dev)
        with quiet_capture(), torch.cuda.graph(self.graph), splitws.owned(pnp=self._split_ws[0], select=self._split_ws[1]):
            self._res = solve_pnp(cfg, self._out, self._gt)

    @torch.no_grad()
    def __call__(self, out_dict, gt_dict):
        for static, new in ((self._out, out_dict), (self._gt, gt_dict)):
            for k, buf in static.items():
                if isinstance(buf, Tensor):
                    src = new[k]
                    if src.shape != buf.shape:
                        raise ValueError(f"GraphedSolvePnP: {k} has shape {tuple(src.shape)}, captured with {tuple(buf.shape)}")
                    if src.data_ptr() != buf.data_ptr():
                        buf.copy_(src, non_blocking=True)
        self.graph.replay()
        return {k: v.clone() for k, v in self._res.items()}
