"""The torch-free C/C++ consumer of the drop-in boundary: builds tests/native/harness.cpp against include/lc_amd.h +
liblc_amd.so with hipcc and runs it (device API, the reference's pnp_ceres_f32_omp symbol, the error channel)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_harness_builds_and_passes(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from lc_amd import _lib

    libdir = os.path.dirname(_lib.lib_path())
    exe = str(tmp_path / "harness")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "native", "harness.cpp"), f"-L{libdir}",
                    "-llc_amd", f"-Wl,-rpath,{libdir}", "-o", exe], check=True, capture_output=True, timeout=600)
    run = subprocess.run([exe, "all"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "harness done, 0 check(s) failed" in run.stdout
