"""The hand-off for the one pin this image cannot produce: tests/golden/gen_golden_pnp_ceres.py runs the REFERENCE's own
`lib.pnp.pnp_ceres.solve` (ceres.cpp behind cffi) wherever that extension exists.  Here the extension is absent, so the generator is
exercised end to end against a stand-in `lib.pnp._ext` -- a ctypes imitation of the two cffi objects the reference's binding uses
(`ffi.new("float*[n]")`, `ffi.cast`, `lib.pnp_ceres_f32_omp`), backed by the CPU oracle's build of the same C symbol -- so that the
generator (argument marshalling through the reference's unmodified pnp_ceres.py, the seven problem sets, the .npz layout the
skip-unless-present tests read) cannot rot.  Output goes to a temporary directory: nothing produced here is a golden vector."""
import ctypes
import importlib.util
import os
import re
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("LC_REFERENCE", "/root/reference")


class _FakeFFI:
    """The three cffi calls of lib/pnp/pnp_ceres.py:93-125, on ctypes."""

    def new(self, decl):
        m = re.fullmatch(r"float\*\[(\d+)\]", decl.replace(" ", ""))
        assert m, decl
        return (ctypes.POINTER(ctypes.c_float) * int(m.group(1)))()

    def cast(self, decl, addr):
        t = {"float*": ctypes.c_float, "int*": ctypes.c_int}[decl.replace(" ", "")]
        return ctypes.cast(ctypes.c_void_p(addr), ctypes.POINTER(t))


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "lib", "pnp")), reason="needs a reference checkout (its own pnp_ceres.py)")
def test_ceres_golden_generator_runs_end_to_end(tmp_path, monkeypatch):
    from oracle import pnp_oracle

    pnp_oracle.build()
    so = ctypes.CDLL(pnp_oracle._SO)
    fn = so.pnp_ceres_f32_omp
    fn.restype = None
    PP = ctypes.POINTER(ctypes.POINTER(ctypes.c_float))
    fn.argtypes = [PP] * 5 + [ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.POINTER(ctypes.c_float),
                   ctypes.POINTER(ctypes.c_int), ctypes.c_int, ctypes.c_int]

    class _Lib:
        @staticmethod
        def pnp_ceres_f32_omp(*a):
            fn(*[ctypes.cast(x, PP) if isinstance(x, ctypes.Array) else x for x in a])

    ext = types.ModuleType("lib.pnp._ext")
    ext.ffi, ext.lib = _FakeFFI(), _Lib()
    monkeypatch.setitem(sys.modules, "lib.pnp._ext", ext)
    for m in [k for k in sys.modules if k == "lib.pnp.pnp_ceres"]:
        monkeypatch.delitem(sys.modules, m)
    spec = importlib.util.spec_from_file_location("gen_golden_pnp_ceres", os.path.join(ROOT, "tests", "golden", "gen_golden_pnp_ceres.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gen.main(out_dir=str(tmp_path))

    from tests.pnp_cases import PNP_CASES, pnp_case

    files = sorted(os.listdir(tmp_path))
    assert files == sorted(f"pnp_ceres_{n}.npz" for n in PNP_CASES)
    for name in PNP_CASES:
        z = np.load(tmp_path / f"pnp_ceres_{name}.npz")
        c = pnp_case(name)
        B = len(c["start"])
        assert z["states"].shape == (B, 7) and z["result_tr"].shape == (B,) and z["rets"].shape == (B,)
        assert np.array_equal(z["in_start"], c["start"]) and np.array_equal(z["in_pts2d"], c["pts2d"])
        # what went through the reference's marshalling is what the oracle's own batched entry computes
        st, tr, ret = pnp_oracle.solve_batched(c["start"], c["K"], c["pts2d"], c["pts3d"], c["sqrtL"], counts=c["counts"],
                                               max_iter=c["max_iter"], ftol=c["ftol"])
        assert np.array_equal(z["rets"], ret) and np.array_equal(z["states"], st) and np.array_equal(z["result_tr"], tr)
    assert not [f for f in os.listdir(os.path.join(ROOT, "tests", "golden")) if f.startswith("pnp_ceres_")], "stand-in output must never land in tests/golden"
