"""GPU initialiser with the call surface of `lib.pnp.cv2_solver.solve` (`lib/pnp/cv2_solver.py:8-60`).

    invalids, states, inliers = gpu_solver.solve(K, pts3d, pts2d, reprojectionError=2)

RANSAC over P3P minimal samples (one wavefront per pose, >= 150 hypotheses like the reference's iterationsCount),
optionally polished by a few unweighted LM iterations on the inliers (the role EPnP-on-inliers plays inside
cv2.solvePnPRansac).  Everything stays on the device; `inliers` are per-pose index tensors like the reference returns.
"""
from __future__ import annotations

import torch

from .. import _lib
from . import cer_solver, pnp_ceres


def workspace_views(ws, B, N, iterations):
    """Diagnostics (the oracle tests): the split form's workspace as tensors -- hyp64 (B,H,12) double and hyp32 (B,H,12) float, every
    hypothesis' [R row-major | t], and the chunk partials (B,C,H) as (count int32, error float32).  The section offsets come from the
    library (`lc_pnp_ransac_workspace_layout`: what lc_pnp_init.hip's carve_workspace does), not from a copy of its arithmetic."""
    import ctypes

    lay = (ctypes.c_size_t * 6)()
    _lib.check(_lib.load().lc_pnp_ransac_workspace_layout(int(B), int(N), int(iterations), lay), "lc_pnp_ransac_workspace_layout")
    o64, o32, opart, H, C, total = (int(v) for v in lay)
    raw = ws.view(torch.uint8)
    assert raw.numel() >= total
    hyp64 = raw[o64:o64 + 8 * 12 * B * H].view(torch.float64).view(B, H, 12)
    hyp32 = raw[o32:o32 + 4 * 12 * B * H].view(torch.float32).view(B, H, 12)
    part = raw[opart:opart + 8 * B * C * H].view(torch.int32).view(B, C, H, 2)
    return hyp64, hyp32, part[..., 0], part[..., 1].view(torch.float32)


def solve_device(cam_mat, coord_3d, coord_2d, n_points=None, *, reprojectionError=3.0, iterations=150, seed=0, refine=True,
                 return_hypothesis=False, split=None, ticketed=False, select=None, reproj_divisor=None, pose_index_offset=0, workspace_out=None):
    """Batched tensors (B,3,3), (B,N,3), (B,N,2) [+ n_points (B)] -> states (B,7), inlier_mask (B,N) bool, invalid (B) bool
    [+ best_hyp (B) int32, n_inliers (B) int32 with return_hypothesis: the integer outputs the oracle test compares exactly].
    pose_index_offset: this batch is the slice [offset, offset + B) of a larger one -- hypothesis streams and padding draws are those of the
    larger batch's poses, so sub-batches solved on several streams return what one call over the whole batch returns.
    reprojectionError: pixels, a scalar or a (B,) tensor; with reproj_divisor (B,) the threshold of pose b is reprojectionError /
    reproj_divisor[b], divided inside the launch (test.py:56-57,115-116: `2 / gt_dict['out_pix_scale']`).

    refine: True runs the inlier refinement (an unweighted LM solve on the inliers, the role EPnP-on-inliers plays inside
    cv2.solvePnPRansac); 'defer' returns (ransac states, inlier_mask, invalid, job) with `job` the keyword arguments of that solve for
    `pnp_ceres.solve_chain_device` -- a caller that continues with a weighted solve runs both as one launch.
    select: None, or a dict(weights=(B,N,2), index=(B,N) int32 | None, min_count=4, seed=0, out=None) -- the 'weighted-filtered'
    re-selection of test.py:129-133 done by the workgroup that writes the inlier mask; the compacted rows
    (pts2d, weights, pts3d, counts, index), exactly `dense.dense_select(..., 'mask', mask=inliers)`'s, come back as
    select['result'].  ticketed: the split form with the selection inside the scoring launch (two launches instead of three, same
    outputs, tests compare the two) -- measured 2.9 us SLOWER per call on MI355X (profiles/r03/test_time/ransac_forms.txt), hence off.

    workspace_out: a list that receives the split form's workspace tensor (see `workspace_views`; diagnostics).
    split: None picks the launch form from the shape -- the single launch keeps a pose on one compute unit (its scoring loop costs
    ~0.07 us per point, times ceil(B/256) when the poses outnumber the compute units), the split form pays ~17 us of extra launches
    and fixed work but spreads the points of a pose over the chip (scripts/ubench/ransac_forms.py: (64, 1024) 83 -> 37 us,
    (256, 64) 27 vs 33 us); True / False force one."""
    lib = _lib.load()
    K = _lib.require_hip_f32("cam_mat", cam_mat)
    X = _lib.require_hip_f32("coord_3d", coord_3d)
    U = _lib.require_hip_f32("coord_2d", coord_2d)
    B, N = X.shape[:2]
    dev = X.device
    if split is None:
        split = N * ((B + 255) // 256) > 256
    counts = None if n_points is None else torch.as_tensor(n_points).to(device=dev, dtype=torch.int32).contiguous()
    per_pose = None
    if reproj_divisor is not None:  # threshold of pose b = reprojectionError / reproj_divisor[b], formed by the kernel (rel_reproj_err)
        if isinstance(reprojectionError, torch.Tensor) or not reprojectionError > 0:
            raise ValueError("reproj_divisor goes with a positive scalar reprojectionError")
        per_pose = reproj_divisor.to(device=dev, dtype=torch.float32).reshape(B).contiguous()
    elif isinstance(reprojectionError, torch.Tensor):
        per_pose = reprojectionError.to(device=dev, dtype=torch.float32).reshape(B).contiguous()
        reprojectionError = 0.0
    states = torch.empty(B, 7, device=dev, dtype=torch.float32)
    mask = torch.empty(B, N, device=dev, dtype=torch.uint8)
    n_in = torch.empty(B, device=dev, dtype=torch.int32)
    invalid = torch.empty(B, device=dev, dtype=torch.int32)
    hyp = torch.empty(B, device=dev, dtype=torch.int32) if return_hypothesis else None
    rows = torch.empty(B, device=dev, dtype=torch.int32) if refine else None  # point count, 0 for the poses RANSAC gave up on
    ws, nbytes = None, 0
    sel_w = sel_idx = None
    sel_out = (None,) * 5
    if select is not None:
        sel_w = _lib.require_hip_f32("select['weights']", select["weights"])
        if sel_w.shape != (B, N, 2):
            raise ValueError(f"select['weights'] must be (B, N, 2) = {(B, N, 2)}, got {tuple(sel_w.shape)}")
        if select.get("index") is not None:
            sel_idx = select["index"].to(device=dev, dtype=torch.int32).contiguous()
        sel_out = select.get("out")
        if sel_out is None:
            sel_out = (torch.empty_like(U), torch.empty_like(sel_w), torch.empty_like(X), torch.empty(B, device=dev, dtype=torch.int32),
                       torch.empty(B, N, device=dev, dtype=torch.int32))
        o_u, o_w, o_x, o_c, o_i = sel_out
        if not (o_u.shape == U.shape and o_w.shape == sel_w.shape and o_x.shape == X.shape and o_c.shape == (B,) and o_i.shape == (B, N)
                and o_u.dtype == o_w.dtype == o_x.dtype == torch.float32 and o_c.dtype == o_i.dtype == torch.int32
                and all(t.is_contiguous() and t.device == dev for t in sel_out)):
            raise ValueError("solve_device: select['out'] buffers must be contiguous tensors of the result shapes on the input's device")
    with _lib.on_device(dev):
        if split:  # hypotheses / scoring / selection as launches over a workspace: spreads one pose over many compute units
            nbytes = int(lib.lc_pnp_ransac_workspace_bytes(B, N, int(iterations)))
            ws = torch.empty((nbytes + 7) // 8, device=dev, dtype=torch.int64)
            if workspace_out is not None:
                workspace_out.append(ws)
        head = (_lib.ptr(K), _lib.ptr(X), _lib.ptr(U), _lib.ptr(counts), B, N, float(reprojectionError), _lib.ptr(per_pose),
                int(iterations), int(seed) & 0xFFFFFFFF, _lib.ptr(states), _lib.ptr(mask), _lib.ptr(n_in), _lib.ptr(invalid), _lib.ptr(hyp),
                _lib.ptr(rows), _lib.ptr(ws), nbytes)
        # (a tensor threshold went in as per-pose values beside a zero scalar: the values ARE the thresholds; beside a positive scalar they divide it)
        rc = lib.lc_pnp_ransac_init5_f32(*head, int(bool(ticketed)), _lib.ptr(sel_w), _lib.ptr(sel_idx), int(select.get("min_count", 4)) if select else 0,
                                         (int(select.get("seed", 0)) if select else 0) & 0xFFFFFFFF, *(_lib.ptr(sel_out[k]) for k in (0, 1, 2, 4, 3)),  # C order: rows, index, counts
                                         int(pose_index_offset), _lib.stream_ptr(dev))
    _lib.check(rc, "lc_pnp_ransac_init5_f32")
    if select is not None:
        select["result"] = sel_out
    inl = mask.view(torch.bool)  # the kernel writes 0 / 1: same bytes, no launch
    bad = invalid.view(torch.bool).view(B, 4)[:, 0]  # the flag is 0 / 1: its low byte as bool, no launch
    if refine:
        # unit information on the inliers (weight_mask); the poses RANSAC gave up on are skipped through their zero point count: the
        # solver returns its start for them and for the solves it flags invalid -- no element-wise launches around the solve
        job = dict(cam_mat=K, pts3d=X, pts2d=U, sqrtL=None, start=states, n_points=rows, max_iter_count=20, weight_mask=mask)
        if refine == "defer":  # the caller chains the refinement with its own solve (pnp_ceres.solve_chain_device: one launch)
            return states, inl, bad, job
        states, _, _ = pnp_ceres.solve_device(**job)
    if return_hypothesis:
        return states, inl, bad, hyp, n_in
    return states, inl, bad


def solve(cam_mat, coord_3d, coord_2d, *, reprojectionError=3.0, confidence=0.99, num_workers=1, **kwargs):
    """`cv2_solver.solve` surface: tensors or per-pose lists -> (invalids, states, inliers) as tuples of per-pose items."""
    single = not isinstance(coord_2d, (list, tuple)) and coord_2d.dim() == 2
    if single:
        cam_mat, coord_3d, coord_2d = cam_mat[None], coord_3d[None], coord_2d[None]
    n_points = None
    if isinstance(coord_3d, (list, tuple)):
        n_points = [len(c) for c in coord_3d]
        dev = coord_3d[0].device
        cam_mat, coord_3d, coord_2d, n_points = cer_solver._batch_tensors(cam_mat, coord_3d, coord_2d, n_points, device=dev)
    states, inl, bad = solve_device(cam_mat, coord_3d, coord_2d, n_points, reprojectionError=reprojectionError,
                                    iterations=kwargs.get("iterationsCount", 150), seed=kwargs.get("seed", 0),
                                    refine=kwargs.get("refine", True))
    invalids = tuple(bool(v) for v in bad.tolist())
    st = tuple(states.unbind(0))
    inliers = tuple(torch.nonzero(m, as_tuple=False)[:, 0] if not b else m.new_zeros(0, dtype=torch.int64) for m, b in zip(inl, invalids))
    if single:
        return invalids[0], st[0], inliers[0]
    return invalids, st, inliers
