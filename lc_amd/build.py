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


HASH_PATH = SO_PATH + ".srchash"


def _deps():
    return sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(os.path.dirname(PKG), "include", "lc_amd.h")]


def source_hash() -> str:
    """sha256 over the names and contents of every source the library is built from (mtimes do not survive the copy onto a
    GPU box; contents do)."""
    import hashlib

    h = hashlib.sha256()
    for d in _deps():
        if os.path.exists(d):
            h.update(os.path.basename(d).encode())
            h.update(open(d, "rb").read())
    return h.hexdigest()


def is_stale() -> bool:
    """True when liblc_amd.so is missing or was built from other source contents than the ones on disk now."""
    if not os.path.exists(SO_PATH) or not os.path.exists(HASH_PATH):
        return True
    return open(HASH_PATH).read().strip() != source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return SO_PATH
    os.makedirs(OUT_DIR, exist_ok=True)
    import fcntl

    with open(SO_PATH + ".lock", "w") as lock:  # one builder at a time (pytest-xdist workers, torchrun ranks)
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not is_stale():
            return SO_PATH
        return _build_locked(verbose)


def _build_locked(verbose: bool) -> str:
    tmp = SO_PATH + f".tmp{os.getpid()}"
    digest = source_hash()
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
           *sources(), "-o", tmp]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, SO_PATH)
    with open(HASH_PATH, "w") as f:
        f.write(digest + "\n")
    return SO_PATH


def build_variant(name: str, flags, verbose: bool = False) -> str:
    """An experiment build of the same sources with extra compiler flags (-D switches) next to the shipped library:
    lc_amd/_C/liblc_amd_<name>.so; select it with LC_AMD_LIB=<path>.  Used by the A/B scripts under scripts/ only."""
    os.makedirs(OUT_DIR, exist_ok=True)
    out = os.path.join(OUT_DIR, f"liblc_amd_{name}.so")
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wno-unused-function",
           *flags, *sources(), "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
