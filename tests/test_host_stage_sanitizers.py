"""The host side of the reference ABI (`pnp_ceres_f32_omp`'s gather of caller pointer arrays into the staging buffer and the
in-place scatter of the results: lc_amd/csrc/lc_host_stage.h, the one piece of host C++ that walks caller memory) under
AddressSanitizer + UndefinedBehaviorSanitizer on the CPU: ragged, zero and negative counts, null point arrays of empty jobs,
> 4096 x 64 points.  (GPU ASan is not available on the pool; the kernel side is covered by the C-ABI harness on the GPU.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_staging_is_clean_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "host_stage_sanitize")
    build = subprocess.run([gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            os.path.join(ROOT, "tests", "native", "host_stage_sanitize.cpp"), "-o", exe], capture_output=True, text=True, timeout=300)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-3000:]
    assert "0 contract violations" in run.stdout and run.stdout.count(" ok") == 6
