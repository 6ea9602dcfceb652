"""Builds lc_amd/_C/liblc_amd.so from lc_amd/csrc/*.hip with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import glob
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OUT_DIR = os.path.join(PKG, "_C")
SO_PATH = os.path.join(OUT_DIR, "liblc_amd.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale() -> bool:
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(PKG), "include", "lc_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return SO_PATH
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = SO_PATH + ".tmp"
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
           *sources(), "-o", tmp]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, SO_PATH)
    return SO_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
