"""The C oracle under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (sanitizers run on the
CPU build only; GPU ASan is not available on the pool)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_oracle_is_clean_under_asan_ubsan(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not available")
    exe = str(tmp_path / "oracle_sanitize")
    build = subprocess.run([gcc, "-O1", "-g", "-std=c99", "-fopenmp", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            os.path.join(ROOT, "oracle", "pnp_lm_oracle.c"), os.path.join(ROOT, "tests", "native", "oracle_sanitize.c"),
                            "-lm", "-o", exe], capture_output=True, text=True, timeout=300)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, run.stdout + run.stderr[-3000:]
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr, run.stderr[-3000:]
    assert "0 contract violations" in run.stdout
