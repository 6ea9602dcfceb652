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


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "lib", "pnp", "cv2_solver.py")), reason="needs a reference checkout (its own cv2_solver.py)")
def test_cv2_golden_generator_and_role_check_run_end_to_end(tmp_path, monkeypatch):
    """The OpenCV hand-off (tests/golden/gen_golden_ransac_cv2.py), exercised against a stand-in `cv2` module -- `solvePnPRansac` played by the
    float64 RANSAC oracle with another seed, so poses and inlier sets differ from ours the way another RANSAC's do -- through the
    reference's UNMODIFIED `lib/pnp/cv2_solver.py` (list marshalling, the axis-angle -> quaternion conversion, the exception path of rows
    with fewer than four points), then the role-contract check the skip-unless-present tests run (tests/ransac_role.py), on the oracle.
    Output goes to a temporary directory: nothing produced here is a golden vector."""
    from scipy.spatial.transform import Rotation

    from oracle import p3p_ransac_oracle as O

    calls = []

    def solvePnPRansac(coord_3d, coord_2d, cam_mat, dist, flags=None, confidence=0.99, iterationsCount=100, reprojectionError=8.0):
        assert dist is None and flags == 1 and iterationsCount == 150 and coord_3d.dtype == coord_2d.dtype == cam_mat.dtype == np.float32
        if len(coord_3d) < 4:
            raise RuntimeError("cv2.error: solvePnPRansac needs at least 4 points")  # OpenCV throws; cv2_solver.py:77-81 turns it into `invalid`
        calls.append(len(coord_3d))
        r = O.ransac(cam_mat, coord_3d, coord_2d, len(coord_3d), reprojectionError, iterationsCount, 777, len(calls))
        if r["invalid"]:
            return False, np.zeros((3, 1)), np.zeros((3, 1)), None
        return True, Rotation.from_matrix(r["R"]).as_rotvec().reshape(3, 1), np.asarray(r["t"]).reshape(3, 1), r["inliers"].astype(np.int32).reshape(-1, 1)

    fake = types.ModuleType("cv2")
    fake.solvePnPRansac, fake.SOLVEPNP_EPNP, fake.__version__ = solvePnPRansac, 1, "stand-in"
    monkeypatch.setitem(sys.modules, "cv2", fake)
    for m in [k for k in sys.modules if k == "lib.pnp.cv2_solver"]:
        monkeypatch.delitem(sys.modules, m)
    spec = importlib.util.spec_from_file_location("gen_golden_ransac_cv2", os.path.join(ROOT, "tests", "golden", "gen_golden_ransac_cv2.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    only = [p for p in gen.problem_sets() if any(t in p for t in ("clean_B16_N32", "ragged_B8_N40", "outliers_B24_N64"))]  # the small sets: the oracle walks every pair in Python
    monkeypatch.setattr(gen, "problem_sets", lambda: only)
    gen.main(out_dir=str(tmp_path))
    assert sorted(os.listdir(tmp_path)) == sorted("ransac_cv2_" + os.path.basename(p)[7:] for p in only) and len(calls) >= 40

    from tests import ransac_role
    from tests.test_oracle_ransac_cv2_golden import oracle_ransac, oracle_refine

    for f in sorted(os.listdir(tmp_path)):
        z = np.load(tmp_path / f)
        src = np.load(os.path.join(ROOT, "tests", "golden", str(z["problem_set"])))
        assert z["states"].shape == (len(src["in_K"]), 7) and z["inlier_mask"].shape == src["in_pts3d"].shape[:2]
        assert z["invalid"][src["in_counts"] < 4].all()  # rows with fewer than four points went through the exception path of cv2_solver.py:77-81
        out = ransac_role.check_role(str(tmp_path / f), oracle_ransac, oracle_refine)
        assert out["poses"] >= 4 and out["dq"] <= 1e-4 and out["iou_min"] >= 0.6, out
    assert not [f for f in os.listdir(os.path.join(ROOT, "tests", "golden")) if f.startswith("ransac_cv2_")], "stand-in output must never land in tests/golden"
