"""`lib.pnp.pnp_ceres` call surface (`lib/pnp/pnp_ceres.py:6-140`) on the HIP batched LM solver.

Two routes, same results:
  * tensors already on the GPU  -> `lc_pnp_lm3_f32` (the solve + the callers' load-time options) on device-resident zero-padded batches (no host round trip);
  * CPU tensors / numpy arrays  -> the reference's own ABI `pnp_ceres_f32_omp` (arrays of host pointers), whose body
    in liblc_amd.so stages the jobs to the GPU.  This is exactly what the reference's cffi marshaller calls.
Returns `(state, result_tr, invalid_flags)` like the reference (`:61`): float32 (B,7), float32 (B,), int32 (B,).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from .. import _lib, splitws


def _to_np(val):
    if not isinstance(val, np.ndarray):
        val = val.detach().to(device="cpu").numpy()
    return val


LC_PNP_WEIGHTS_ARE_ICOV, LC_PNP_NAN_TO_NUM, LC_PNP_WEIGHTS_ARE_STD = 1, 2, 4  # include/lc_amd.h


SPLIT_WORKSPACE_MAX_BYTES = splitws.PNP_MAX_BYTES
no_split = splitws.no_split
split_is_off = splitws.is_off


def owned_split_workspace(ws):
    """`splitws.owned(pnp=ws)`: solves inside the block use `ws` (zeroed uint8 tensor allocated before a stream capture) as their workspace."""
    return splitws.owned(pnp=ws)


def split_workspace(dev, *shapes, split=None):
    """Workspace for solves of few poses x thousands of correspondences (`lc_pnp_lm_workspace_bytes`: several workgroups per pose), sized for
    the largest of `shapes` = (B, N) pairs; None when none of them takes that form, or with split=False / LC_AMD_PNP_SPLIT=0 / `no_split()`."""
    lib = _lib.load()
    return splitws.get("pnp", dev, max(int(lib.lc_pnp_lm_workspace_bytes(int(B), int(N))) for B, N in shapes), split)


def solve_device(cam_mat, pts3d, pts2d, sqrtL, start, n_points=None, *, max_iter_count=50, function_tolerance=1e-6,
                 return_iters=False, trace_rows=0, weights_are_icov=False, nan_to_num=False, weight_mask=None, shared_poses=0, split=None,
                 weights_are_std=False):
    """Device route. cam_mat (B,3,3) pts3d (B,N,3) pts2d (B,N,2) start (B,7); sqrtL (B,N,2,2) lower factor or
    (B,N,2) diagonal; n_points (B,) int or None.  trace_rows > 0 runs the diagnostic twin of the kernel and appends the
    (B,trace_rows,8) float64 per-iteration schedule (`lc_pnp_lm_trace_f32`, include/lc_amd.h) to the returned tuple.

    Folded into the kernel's loads instead of separate element-wise launches (`lc_pnp_lm3_f32`): weights_are_icov (the diagonal
    tensor holds inverse variances), weights_are_std (it holds standard deviations: `1/(std**2)` of test.py:52 formed at the load), nan_to_num (torch.nan_to_num on every input; an invalid job returns the filtered start),
    weight_mask (B,N) uint8/bool in place of sqrtL: unit information where set; shared_poses = P > 0: cam_mat and start have P rows
    and pose b of the B = k P correspondence sets reads row b % P (several selections of the same objects in one launch).
    split (default: on, LC_AMD_PNP_SPLIT=0 turns it off): batches of at most 128 poses with rows wider than 2048 are solved by several
    workgroups per pose (`lc_pnp_lm3_f32`) -- the same solve up to the order of the fp64 sums.  The result never depends on whether those
    workgroups got to run together: a rescue launch behind the split launch re-solves, bit for bit, whatever did not meet (include/lc_amd.h)."""
    lib = _lib.load()
    K = _lib.require_hip_f32("cam_mat", cam_mat)
    X = _lib.require_hip_f32("pts3d", pts3d)
    U = _lib.require_hip_f32("pts2d", pts2d)
    B, N = X.shape[:2]
    dev = X.device
    L = M = None
    if weight_mask is not None:
        M = weight_mask.view(torch.uint8) if weight_mask.dtype == torch.bool else weight_mask
        if M.dtype != torch.uint8 or not M.is_cuda or tuple(M.shape) != (B, N):
            raise TypeError("weight_mask must be a (B,N) uint8/bool tensor on the GPU")
        M = M.contiguous()
    else:
        L = _lib.require_hip_f32("pts2d_icov_sqrtL", sqrtL)
    full = L is not None and L.dim() == 4
    start = _lib.require_hip_f32("start", start)
    if shared_poses and (start.shape[0] != shared_poses or K.shape[0] != shared_poses or B % shared_poses):
        raise ValueError("shared_poses: cam_mat and start need that many rows and B must be a multiple of it")
    state = torch.empty(B, 7, device=dev, dtype=torch.float32)
    counts = None if n_points is None else torch.as_tensor(n_points).to(device=dev, dtype=torch.int32).contiguous()
    tr = torch.empty(B, device=dev, dtype=torch.float32)
    ret = torch.empty(B, device=dev, dtype=torch.int32)
    iters = torch.empty(B, device=dev, dtype=torch.int32) if return_iters else None
    opts = (LC_PNP_WEIGHTS_ARE_ICOV if weights_are_icov or weights_are_std else 0) | (LC_PNP_NAN_TO_NUM if nan_to_num else 0) | (LC_PNP_WEIGHTS_ARE_STD if weights_are_std else 0)
    if trace_rows > 0:
        if opts or M is not None or shared_poses:
            raise ValueError("the diagnostic trace takes plain inputs")
        trace = torch.zeros(B, int(trace_rows), 8, device=dev, dtype=torch.float64)
        with _lib.on_device(dev):
            rc = lib.lc_pnp_lm_trace_f32(_lib.ptr(K), _lib.ptr(X), _lib.ptr(U), _lib.ptr(L) if full else None,
                                         None if full else _lib.ptr(L), _lib.ptr(counts), _lib.ptr(start), _lib.ptr(state), _lib.ptr(tr),
                                         _lib.ptr(ret), _lib.ptr(iters), B, N, int(max_iter_count), float(function_tolerance),
                                         _lib.ptr(trace), int(trace_rows), _lib.stream_ptr(dev))
        _lib.check(rc, "lc_pnp_lm_trace_f32")
        return (state, tr, ret, iters, trace) if return_iters else (state, tr, ret, trace)
    with _lib.on_device(dev):  # options = 0 without a mask and without a workspace: the plain solve (same kernel instantiation)
        ws = split_workspace(dev, (B, N), split=split)
        rc = lib.lc_pnp_lm3_f32(_lib.ptr(K), _lib.ptr(X), _lib.ptr(U), _lib.ptr(L) if full else None,
                                _lib.ptr(L) if (L is not None and not full) else None, _lib.ptr(M), _lib.ptr(counts), _lib.ptr(start),
                                _lib.ptr(state), _lib.ptr(tr), _lib.ptr(ret), _lib.ptr(iters), B, N, int(max_iter_count),
                                float(function_tolerance), opts, int(shared_poses), _lib.ptr(ws), 0 if ws is None else ws.numel(),
                                _lib.stream_ptr(dev))
    _lib.check(rc, "lc_pnp_lm3_f32")
    return (state, tr, ret, iters) if return_iters else (state, tr, ret)


class _Job(ctypes.Structure):  # include/lc_amd.h: lc_pnp_lm_job
    _fields_ = [(n, ctypes.c_void_p) for n in ("K", "pts3d", "pts2d", "sqrtL", "weights_diag", "weight_mask", "counts", "start", "states",
                                               "result_tr", "rets", "iters")] + \
               [("B", ctypes.c_int), ("Nmax", ctypes.c_int), ("max_iter", ctypes.c_int), ("function_tolerance", ctypes.c_float),
                ("options", ctypes.c_int), ("pose_mod", ctypes.c_int)]


def _job(cam_mat, pts3d, pts2d, sqrtL, start, n_points=None, *, max_iter_count=50, function_tolerance=1e-6, weights_are_icov=False,
         nan_to_num=False, weight_mask=None, shared_poses=0, weights_are_std=False):
    """The arguments of `solve_device` as an lc_pnp_lm_job + its output tensors (state, tr, ret) + the tensors the job points into."""
    K = _lib.require_hip_f32("cam_mat", cam_mat)
    X = _lib.require_hip_f32("pts3d", pts3d)
    U = _lib.require_hip_f32("pts2d", pts2d)
    B, N = X.shape[:2]
    dev = X.device
    L = M = None
    if weight_mask is not None:
        M = weight_mask.view(torch.uint8) if weight_mask.dtype == torch.bool else weight_mask
        if M.dtype != torch.uint8 or not M.is_cuda or tuple(M.shape) != (B, N):
            raise TypeError("weight_mask must be a (B,N) uint8/bool tensor on the GPU")
        M = M.contiguous()
    else:
        L = _lib.require_hip_f32("pts2d_icov_sqrtL", sqrtL)
    full = L is not None and L.dim() == 4
    start = _lib.require_hip_f32("start", start)
    if shared_poses and (start.shape[0] != shared_poses or K.shape[0] != shared_poses or B % shared_poses):
        raise ValueError("shared_poses: cam_mat and start need that many rows and B must be a multiple of it")
    state = torch.empty(B, 7, device=dev, dtype=torch.float32)
    counts = None if n_points is None else torch.as_tensor(n_points).to(device=dev, dtype=torch.int32).contiguous()
    tr = torch.empty(B, device=dev, dtype=torch.float32)
    ret = torch.empty(B, device=dev, dtype=torch.int32)
    opts = (LC_PNP_WEIGHTS_ARE_ICOV if weights_are_icov or weights_are_std else 0) | (LC_PNP_NAN_TO_NUM if nan_to_num else 0) | (LC_PNP_WEIGHTS_ARE_STD if weights_are_std else 0)
    P = _lib.ptr
    job = _Job(P(K), P(X), P(U), P(L) if full else None, P(L) if (L is not None and not full) else None, P(M), P(counts), P(start), P(state),
               P(tr), P(ret), None, B, N, int(max_iter_count), float(function_tolerance), opts, int(shared_poses))
    return job, (state, tr, ret), (K, X, U, L, M, counts, start), dev


def solve_chain_device(first: dict, second: dict, split=None):
    """Two solves as one call (`lc_pnp_lm_chain2_f32`): `first` and `second` are keyword arguments of `solve_device`; second['start'] may be
    the string 'first' -- the states the first solve returns (with shared_poses = the first job's batch when the second holds several
    selections of the same objects).  One launch where the shapes allow (both 256 < N <= 1024), else the two launches; the same
    results as `solve_device(**first)` followed by `solve_device(**second)` bit for bit.  Returns ((state, tr, ret), (state, tr, ret))."""
    lib = _lib.load()
    j1, out1, keep1, dev = _job(**first)
    if isinstance(second.get("start"), str):
        if second["start"] != "first":
            raise ValueError("second['start'] is a tensor or the string 'first'")
        second = dict(second, start=out1[0])
    j2, out2, keep2, dev2 = _job(**second)
    if dev2 != dev:
        raise ValueError("solve_chain_device: both jobs on one device")
    with _lib.on_device(dev):
        ws = split_workspace(dev, (j1.B, j1.Nmax), (j2.B, j2.Nmax), split=split)
        rc = lib.lc_pnp_lm_chain2_f32(ctypes.byref(j1), ctypes.byref(j2), _lib.ptr(ws), 0 if ws is None else ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "lc_pnp_lm_chain2_f32")
    del keep1, keep2
    return out1, out2


def _solve_host_lists(state, pts3d, pts2d, sqrtL, cam_mat, point_counts, worker_count=1, max_iter_count=300, **kwargs):
    """`_pnp_ceres_omp_f32` (pnp_ceres.py:74-140): per-job contiguous float32 arrays -> `pnp_ceres_f32_omp`."""
    lib = _lib.load()
    job_cnt = len(state)
    state = np.stack([np.ascontiguousarray(s, np.float32) for s in state]).astype(np.float32)
    keep = [[np.ascontiguousarray(a, np.float32) for a in lst] for lst in (cam_mat, pts2d, pts3d, sqrtL)]
    fp = ctypes.POINTER(ctypes.c_float)
    PA = fp * job_cnt
    as_ptr = lambda a: a.ctypes.data_as(fp)
    states = PA(*[as_ptr(state[i]) for i in range(job_cnt)])
    ptrs = [PA(*[as_ptr(a) for a in lst]) for lst in keep]
    result_tr = np.zeros(job_cnt, np.float32)
    flags = np.zeros(job_cnt, np.int32)
    counts = np.asarray(point_counts, dtype=np.int32)
    lib.pnp_ceres_f32_omp(states, ptrs[0], ptrs[1], ptrs[2], ptrs[3], counts.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                          int(max_iter_count), float(kwargs.get("function_tolerance", 1e-6)), int(kwargs.get("print_summary", 0)),
                          as_ptr(result_tr), flags.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), job_cnt, int(worker_count))
    return state, result_tr, flags


def solve(cam_mat, pts3d, pts2d, pts2d_icov_sqrtL, start, n_points=None, *, max_iter_count=50, num_workers=1, **kwargs):
    """Drop-in for `lib.pnp.pnp_ceres.solve` (pnp_ceres.py:6-61); batched tensors or per-job lists, or one un-batched job."""
    single = False
    state = start
    on_gpu = isinstance(pts3d, torch.Tensor) and pts3d.is_cuda and isinstance(state, torch.Tensor) and state.dim() == 2
    if on_gpu:
        return solve_device(cam_mat, pts3d, pts2d, pts2d_icov_sqrtL, state, n_points, max_iter_count=max_iter_count,
                            function_tolerance=kwargs.get("function_tolerance", 1e-6))
    if not isinstance(state, (list, tuple)) and len(state.shape) == 1:
        single = True
        if tuple(pts2d.shape) == tuple(pts2d_icov_sqrtL.shape):
            pts2d_icov_sqrtL = torch.diag_embed(torch.as_tensor(pts2d_icov_sqrtL))
        n_points = [int(n_points)] if n_points is not None else [pts2d.shape[0]]
        state, pts3d, pts2d, pts2d_icov_sqrtL, cam_mat = ([_to_np(v)] for v in (state, pts3d, pts2d, pts2d_icov_sqrtL, cam_mat))
    else:
        if tuple(pts2d[0].shape) == tuple(pts2d_icov_sqrtL[0].shape):
            pts2d_icov_sqrtL = [torch.diag_embed(torch.as_tensor(c)) for c in pts2d_icov_sqrtL]
        state, pts3d, pts2d, pts2d_icov_sqrtL, cam_mat = ([_to_np(v) for v in lst] for lst in
                                                          (state, pts3d, pts2d, pts2d_icov_sqrtL, cam_mat))
        n_points = [int(n) for n in n_points] if n_points is not None else [int(c.shape[0]) for c in pts2d]
    outs = _solve_host_lists(state, pts3d, pts2d, pts2d_icov_sqrtL, cam_mat, n_points, worker_count=num_workers,
                             max_iter_count=max_iter_count, **kwargs)
    outs = [torch.from_numpy(o) for o in outs]
    return outs if not single else [o[0] for o in outs]
