"""ctypes front-end of oracle/pnp_lm_oracle.c (TEST INFRASTRUCTURE ONLY).

Mirrors `lib/pnp/pnp_ceres.py:74-140 _pnp_ceres_omp_f32` (pointer-array marshalling onto
`pnp_ceres_f32_omp`) plus a contiguous-batch helper used by tests and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libpnp_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pnp_lm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        fpp = ctypes.POINTER(ctypes.POINTER(ctypes.c_float))
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int)
        _lib.pnp_ceres_f32_omp.argtypes = [fpp, fpp, fpp, fpp, fpp, ip, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                           fp, ip, ctypes.c_int, ctypes.c_int]
        _lib.pnp_ceres_f32_omp.restype = None
        _lib.pnp_oracle_batched_f32.argtypes = [fp, fp, fp, fp, fp, ip, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                fp, ip, ctypes.c_int, ctypes.c_int]
        _lib.pnp_oracle_batched_f32.restype = None
        _lib.pnp_oracle_batched_trace_f32.argtypes = _lib.pnp_oracle_batched_f32.argtypes + [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ip]
        _lib.pnp_oracle_batched_trace_f32.restype = None
        dp = ctypes.POINTER(ctypes.c_double)
        _lib.oracle_powell_trace.argtypes = [dp, ctypes.c_int, ctypes.c_double, dp, ctypes.c_int, ip, dp, dp]
        _lib.oracle_powell_trace.restype = ctypes.c_int
        _lib.oracle_hello_trace.argtypes = [dp, ctypes.c_int, ctypes.c_double, dp, ctypes.c_int, ip, dp]
        _lib.oracle_hello_trace.restype = ctypes.c_int
        _lib.oracle_radius_schedule.argtypes = [dp, ctypes.c_int, dp]
        _lib.oracle_radius_schedule.restype = None
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def solve_pointer_arrays(states, Ks, pts2d, pts3d, sqrtL, counts, max_iter=50, ftol=1e-6, num_threads=1, symbol_lib=None):
    """Call `pnp_ceres_f32_omp` exactly the way the reference's cffi marshaller does (lists of per-job arrays).

    `symbol_lib` lets the tests drive ANOTHER library exporting the same symbol (the HIP drop-in) with the same code.
    """
    L = symbol_lib or lib()
    B = len(states)
    states = [np.ascontiguousarray(s, np.float32).copy() for s in states]
    arrs = [[np.ascontiguousarray(a, np.float32) for a in lst] for lst in (Ks, pts2d, pts3d, sqrtL)]
    PA = ctypes.POINTER(ctypes.c_float) * B
    sp = PA(*[_fp(s) for s in states])
    ptrs = [PA(*[_fp(a) for a in lst]) for lst in arrs]
    counts = np.asarray(counts, np.int32)
    tr = np.zeros(B, np.float32)
    ret = np.zeros(B, np.int32)
    fn = L.pnp_ceres_f32_omp
    fn.restype = None
    fn(sp, ptrs[0], ptrs[1], ptrs[2], ptrs[3], counts.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
       ctypes.c_int(int(max_iter)), ctypes.c_float(float(ftol)), ctypes.c_int(0), _fp(tr),
       ret.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), ctypes.c_int(B), ctypes.c_int(int(num_threads)))
    return np.stack(states), tr, ret


def solve_batched(states, Ks, pts2d, pts3d, sqrtL, counts=None, max_iter=50, ftol=1e-6, num_threads=1):
    """Contiguous (B,7),(B,3,3),(B,N,2),(B,N,3),(B,N,2,2) float32 arrays -> (states, trust_radius, invalid)."""
    L = lib()
    states = np.ascontiguousarray(states, np.float32).copy()
    Ks, pts2d, pts3d, sqrtL = (np.ascontiguousarray(a, np.float32) for a in (Ks, pts2d, pts3d, sqrtL))
    B, N = pts3d.shape[:2]
    counts = np.full(B, N, np.int32) if counts is None else np.ascontiguousarray(counts, np.int32)
    tr = np.zeros(B, np.float32)
    ret = np.zeros(B, np.int32)
    ip = ctypes.POINTER(ctypes.c_int)
    L.pnp_oracle_batched_f32(_fp(states), _fp(Ks), _fp(pts2d), _fp(pts3d), _fp(sqrtL), counts.ctypes.data_as(ip), N,
                             int(max_iter), float(ftol), _fp(tr), ret.ctypes.data_as(ip), B, int(num_threads))
    return states, tr, ret


TRACE_COLS = 8  # PNP_TRACE_COLS: kind, cost at x, candidate cost, model cost change, rho, ||step||, radius after, max|g| after


def solve_batched_trace(states, Ks, pts2d, pts3d, sqrtL, counts=None, max_iter=50, ftol=1e-6, num_threads=1, trace_rows=50):
    """`solve_batched` + the per-iteration trust-region schedule: returns (states, trust_radius, invalid, iters, trace)
    with trace (B, trace_rows, TRACE_COLS) float64 (rows beyond a job's iteration count stay zero)."""
    L = lib()
    states = np.ascontiguousarray(states, np.float32).copy()
    Ks, pts2d, pts3d, sqrtL = (np.ascontiguousarray(a, np.float32) for a in (Ks, pts2d, pts3d, sqrtL))
    B, N = pts3d.shape[:2]
    counts = np.full(B, N, np.int32) if counts is None else np.ascontiguousarray(counts, np.int32)
    tr = np.zeros(B, np.float32)
    ret = np.zeros(B, np.int32)
    iters = np.zeros(B, np.int32)
    trace = np.zeros((B, trace_rows, TRACE_COLS), np.float64)
    ip = ctypes.POINTER(ctypes.c_int)
    L.pnp_oracle_batched_trace_f32(_fp(states), _fp(Ks), _fp(pts2d), _fp(pts3d), _fp(sqrtL), counts.ctypes.data_as(ip), N,
                                   int(max_iter), float(ftol), _fp(tr), ret.ctypes.data_as(ip), B, int(num_threads),
                                   trace.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), int(trace_rows), iters.ctypes.data_as(ip))
    return states, tr, ret, iters, trace


def powell_trace(x0=(3.0, -1.0, 0.0, 1.0), max_iter=100, ftol=1e-6, trace_rows=100):
    """The oracle's minimiser (the same `lm_minimize` the PnP solve runs through) on Powell's function, the example whose
    per-iteration log the Ceres documentation prints.  -> (converged, x (4,), iterations, final radius, (cost0, max|g0|), trace)."""
    L = lib()
    dp = ctypes.POINTER(ctypes.c_double)
    x = np.array(x0, np.float64)
    trace = np.zeros((trace_rows, TRACE_COLS), np.float64)
    n = ctypes.c_int(0)
    radius, initial = np.zeros(1), np.zeros(2)
    ok = L.oracle_powell_trace(x.ctypes.data_as(dp), int(max_iter), float(ftol), trace.ctypes.data_as(dp), int(trace_rows), ctypes.byref(n),
                               radius.ctypes.data_as(dp), initial.ctypes.data_as(dp))
    return bool(ok), x, n.value, float(radius[0]), (float(initial[0]), float(initial[1])), trace[:n.value]


def hello_trace(x0=0.5, max_iter=100, ftol=1e-6, trace_rows=10):
    """`lm_minimize` on f = 10 - x (the tutorial's first example).  -> (converged, x, iterations, final radius, trace)."""
    L = lib()
    dp = ctypes.POINTER(ctypes.c_double)
    x = np.array([x0], np.float64)
    trace = np.zeros((trace_rows, TRACE_COLS), np.float64)
    n = ctypes.c_int(0)
    radius = np.zeros(1)
    ok = L.oracle_hello_trace(x.ctypes.data_as(dp), int(max_iter), float(ftol), trace.ctypes.data_as(dp), int(trace_rows), ctypes.byref(n),
                              radius.ctypes.data_as(dp))
    return bool(ok), float(x[0]), n.value, float(radius[0]), trace[:n.value]


def radius_schedule(quality):
    """The minimiser's own radius updates (`radius_step_accepted` / `radius_step_rejected`) driven by a sequence of step qualities
    (<= 1e-3: rejected), starting from the initial radius 1e4.  -> radii after every step."""
    L = lib()
    dp = ctypes.POINTER(ctypes.c_double)
    q = np.asarray(quality, np.float64)
    out = np.zeros(len(q), np.float64)
    L.oracle_radius_schedule(q.ctypes.data_as(dp), len(q), out.ctypes.data_as(dp))
    return out
