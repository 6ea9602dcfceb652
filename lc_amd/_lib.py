"""ctypes binding of liblc_amd.so (the C ABI declared in include/lc_amd.h).

The product path has NO CPU fallback: if the library is missing, or a tensor is not a float32 HIP tensor,
the callers below raise.  (The CPU oracle under oracle/ is test infrastructure and is never imported here.)
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_float, c_int, c_void_p

import torch

from . import build as _build

_LIB = None

_F = ctypes.POINTER(c_float)
_FP = ctypes.POINTER(_F)
_I = ctypes.POINTER(c_int)

_SIGNATURES = {
    "lc_amd_source_hash": (ctypes.c_char_p, []),
    "lc_amd_version": (c_int, []),
    "lc_amd_last_error": (ctypes.c_char_p, []),
    "pnp_ceres_f32_omp": (None, [_FP, _FP, _FP, _FP, _FP, _I, c_int, c_float, c_int, _F, _I, c_int, c_int]),
    "lc_pnp_lm3_f32": (c_int, [c_void_p] * 12 + [c_int, c_int, c_int, c_float, c_int, c_int, c_void_p, ctypes.c_size_t, c_void_p]),
    "lc_pnp_lm_workspace_bytes": (ctypes.c_size_t, [c_int, c_int]),
    "lc_split_workspace_rescues": (ctypes.c_longlong, [c_void_p, ctypes.c_size_t, c_int, c_void_p]),
    "lc_pnp_lm_chain2_f32": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_size_t, c_void_p]),
    "lc_pnp_lm_trace_f32": (c_int, [c_void_p] * 11 + [c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p]),
    "lc_cov_loss3_fwd_bwd_f32": (c_int, [c_void_p] * 8 + [c_int, c_int, c_float, c_float, c_float, c_int] + [c_void_p] * 6 + [ctypes.c_size_t, c_void_p]),
    "lc_cov_loss_workspace_bytes": (ctypes.c_size_t, [c_int, c_int]),
    "lc_pose_unit2_f32": (c_int, [c_void_p] * 8 + [c_int, c_int, c_float, c_float, c_float] + [c_void_p] * 10 +
                          [c_int, c_float, c_void_p, ctypes.c_size_t, c_void_p]),
    "lc_scale_rows_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                  c_int, c_void_p]),
    "lc_pnp_ransac_init5_f32": (c_int, [c_void_p] * 4 + [c_int, c_int, c_float, c_void_p, c_int, ctypes.c_uint] + [c_void_p] * 7 +
                                [ctypes.c_size_t, c_int, c_void_p, c_void_p, c_int, ctypes.c_uint] + [c_void_p] * 5 + [c_int, c_void_p]),
    "lc_pnp_ransac_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "lc_pnp_ransac_workspace_layout": (c_int, [c_int, c_int, c_int, ctypes.POINTER(ctypes.c_size_t)]),
    "lc_pose_errors_f32": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_void_p, c_void_p]),
    "lc_kpt_nll_fwd_bwd_f32": (c_int, [c_void_p] * 5 + [c_int, c_int] + [c_void_p] * 4),
    "lc_dense_select_f32": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, ctypes.c_double, c_int, c_int, ctypes.c_uint] + [c_void_p] * 6),
    "lc_softargmax2d_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lc_softargmax2d_bwd": (c_int, [c_void_p, c_int] + [c_void_p] * 5 + [c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "lc_sqnorm": (c_int, [c_void_p, c_int, ctypes.c_longlong, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "lc_norm_clip_apply": (c_int, [c_void_p, c_int, ctypes.c_longlong, c_void_p, c_void_p, c_float, c_float, ctypes.c_double,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    # the f1 / f3 entry points: maps of any element type (map_dtype) lying anywhere a sample is contiguous (batch strides)
    "lc_dense_frontend_fwd3": (c_int, [c_void_p] * 5 + [c_float, c_int, c_int, c_int] + [ctypes.c_longlong] * 3 + [c_int] * 6 + [c_void_p] * 6),
    "lc_dense_frontend_bwd2": (c_int, [c_void_p] * 6 + [c_int, c_int, ctypes.c_longlong] + [c_int] * 6 + [c_void_p] * 4),
    "lc_dense_frontend_select3": (c_int, [c_void_p] * 5 + [c_float, c_int, c_int, c_int] + [ctypes.c_longlong] * 3 + [c_int] * 7 + [ctypes.c_double, c_int, c_int, ctypes.c_uint, c_int] +
                                  [c_void_p] * 5 + [c_void_p, ctypes.c_size_t, c_void_p]),
    "lc_dense_frontend_select_workspace_bytes": (ctypes.c_size_t, [c_int] * 6),
    "lc_bits_decode_gt_fwd3": (c_int, [c_void_p] * 5 + [c_int, ctypes.c_longlong] + [c_int] * 11 + [c_void_p, c_void_p]),
    "lc_bits_decode_gt_bwd3": (c_int, [c_void_p] * 6 + [c_int, ctypes.c_longlong] + [c_int] * 11 + [c_void_p, c_void_p]),
    "lc_bits_decode3": (c_int, [c_void_p] * 3 + [c_int, ctypes.c_longlong] + [c_int] * 9 + [c_void_p, c_void_p]),
    "lc_bits_decode_rows": (c_int, [c_void_p] * 3 + [c_int, ctypes.c_longlong] + [c_int] * 11 + [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "lc_dense_aux_fwd2": (c_int, [c_void_p] * 7 + [c_int] + [ctypes.c_longlong] * 3 + [c_int] * 3 + [c_void_p] * 4),
    "lc_dense_aux_bwd2": (c_int, [c_void_p] * 7 + [c_int] + [ctypes.c_longlong] * 3 + [c_int] * 3 + [c_void_p] * 7),
    "lc_xyz_bin_loss_fwd2": (c_int, [c_void_p] * 3 + [c_int, ctypes.c_longlong, ctypes.c_longlong] + [c_int] * 3 + [c_float] + [c_void_p] * 6),
    "lc_xyz_bin_loss_counts": (c_int, [c_void_p] * 3 + [c_int, ctypes.c_longlong, ctypes.c_longlong] + [c_int] * 3 + [c_void_p] * 5),
    "lc_xyz_bin_loss_finish": (c_int, [c_void_p, c_void_p, c_int, c_float] + [c_void_p] * 4),
    "lc_xyz_bin_loss_bwd2": (c_int, [c_void_p] * 5 + [c_int, ctypes.c_longlong, ctypes.c_longlong] + [c_int] * 3 + [c_void_p] * 2),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def lib_path() -> str:
    return os.environ.get("LC_AMD_LIB", _build.SO_PATH)


def load(build_if_missing: bool = True):
    """Load (building first if the .so is absent and hipcc is available). Raises if it cannot be loaded."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if path == _build.SO_PATH and _build.is_stale():
        # missing, or built from other .hip/.h contents than the ones on disk (the library carries the hash of its sources):
        # rebuild (hipcc cross-compiles without a GPU); never silently run a library that ignores edited sources
        if os.path.exists(path) and not _build.hipcc_available():
            # a deployed copy on a box without the compiler: it cannot be rebuilt.  A library whose struct or argument layouts may differ
            # from the sources next to it is refused unless the caller says so explicitly (a warning is lost wherever stderr is discarded)
            if os.environ.get("LC_AMD_ALLOW_STALE") != "1":
                raise RuntimeError(f"lc_amd: {path} was built from other sources than the ones next to it (embedded hash "
                                   f"{_build.embedded_hash(path)}, sources {_build.source_hash()}) and hipcc is not available to rebuild it; "
                                   f"set LC_AMD_ALLOW_STALE=1 to load it as it is")
            import warnings

            warnings.warn(f"lc_amd: loading {path} although it was built from other sources than the ones next to it (LC_AMD_ALLOW_STALE=1)")
        elif not build_if_missing:
            raise RuntimeError(f"lc_amd: {path} is missing or stale; run `python __graft_entry__.py build`")
        else:
            try:
                _build.build()
            except Exception as e:  # noqa: BLE001
                raise RuntimeError(f"lc_amd: {path} is missing or older than lc_amd/csrc and could not be rebuilt ({e}); "
                                   f"run `python __graft_entry__.py build` where hipcc is available") from e
    elif not os.path.exists(path):
        raise RuntimeError(f"lc_amd: LC_AMD_LIB={path} does not exist")
    lib = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _LIB = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().lc_amd_last_error().decode(errors="replace")
        raise RuntimeError(f"lc_amd.{what} failed (code {rc}): {msg}")


def require_hip_f32(name: str, t: torch.Tensor) -> torch.Tensor:
    """The kernels take contiguous float32 device tensors.  Half-precision network outputs (fp16 / bf16 heads under mixed
    precision: BASELINE.json configs 3 and 5) are up-cast here, differentiably, so the loss itself always runs at the
    reference's fp32 I/O precision; anything else (CPU tensors, float64, integers) is an error, never a silent fallback."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"lc_amd: {name} must be a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"lc_amd: {name} is on {t.device}; the HIP path needs tensors on the MI355X "
                           f"(there is no CPU fallback in the product path)")
    if t.dtype in (torch.float16, torch.bfloat16):
        t = t.float()
    if t.dtype != torch.float32:
        raise TypeError(f"lc_amd: {name} must be float32 (reference I/O precision), got {t.dtype}")
    return t.contiguous()


MAP_DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}  # LC_F32 / LC_F16 / LC_BF16 (include/lc_amd.h)


def require_hip_map(name: str, t: torch.Tensor) -> torch.Tensor:
    """(…,H,W) maps of the keypoint head: fp32, fp16 or bf16 device tensors are consumed in their own type (no up-cast copy)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"lc_amd: {name} must be a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise RuntimeError(f"lc_amd: {name} is on {t.device}; the HIP path needs tensors on the MI355X "
                           f"(there is no CPU fallback in the product path)")
    if t.dtype not in MAP_DTYPES:
        raise TypeError(f"lc_amd: {name} must be float32, float16 or bfloat16, got {t.dtype}")
    return t.contiguous()


def hip_maps(**named):
    """Network-output maps of ONE call (`xyz_noc`, `xyz_weight_logits`, `msk_vis_logits`, code logits ...; None entries pass through):
    -> (tensors, batch strides in elements, map_dtype code of include/lc_amd.h).

    The kernels read fp32, fp16 and bf16 maps in their own type (no up-cast copy) from wherever every sample is contiguous: a dense
    batch, or a channel slice `out_raw[:, a:b]` of the network's (B,C_all,H,W) output (what `ptnet.py:56` hands over) -- its batch
    stride is passed along and the slice is consumed where it lies.  Only two cases copy: a sample that is not contiguous itself
    (copied into shape), and maps of DIFFERENT element types in one call (the 16-bit ones are up-cast to fp32)."""
    for name, t in named.items():
        if t is None:
            continue
        if not isinstance(t, torch.Tensor):
            raise TypeError(f"lc_amd: {name} must be a torch.Tensor, got {type(t)}")
        if not t.is_cuda:
            raise RuntimeError(f"lc_amd: {name} is on {t.device}; the HIP path needs tensors on the MI355X "
                               f"(there is no CPU fallback in the product path)")
        if t.dtype not in MAP_DTYPES:
            raise TypeError(f"lc_amd: {name} must be float32, float16 or bfloat16, got {t.dtype}")
    dtypes = {t.dtype for t in named.values() if t is not None}
    mixed = len(dtypes) > 1
    out, strides = [], []
    for t in named.values():
        if t is None:
            out.append(None)
            strides.append(0)
            continue
        if mixed and t.dtype != torch.float32:
            t = t.float()
        per_sample = t[0].numel() if t.shape[0] else 0
        if t.shape[0] > 1 and not (t[0].is_contiguous() and t.stride(0) >= per_sample):
            t = t.contiguous()
        elif t.shape[0] <= 1 and not t.is_contiguous():
            t = t.contiguous()
        out.append(t)
        strides.append(int(t.stride(0)) if t.shape[0] > 1 else 0)
    code = MAP_DTYPES[torch.float32 if (mixed or not dtypes) else next(iter(dtypes))]
    return out, strides, code


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def on_device(device):
    """`torch.cuda.device(device)` for the launch, at no cost in the usual case that `device` already is the current one
    (the context manager itself is ~3 us of the ~15 us a wrapper call costs on the host)."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # the current stream's handle without building a Stream object (~0.3 us against ~2)


def raw_stream(device=None) -> int:
    """Handle of torch's current stream on `device` as an integer."""
    if _raw_stream is not None:
        idx = None if device is None else getattr(device, "index", device)
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


def stream_ptr(device=None):
    return c_void_p(raw_stream(device))
