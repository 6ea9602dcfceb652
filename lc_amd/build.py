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


HASH_MARKER = b"LC_AMD_SRC_HASH:"  # the library carries the hash of its own sources (lc_capi.hip: lc_amd_source_hash)


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
    h.update(repr(sorted(PER_FILE_FLAGS.items())).encode())  # a change of the compiler flags is a change of the build
    h.update(repr(COMMON_FLAGS).encode())
    return h.hexdigest()


def embedded_hash(path: str = SO_PATH):
    """The source hash a built library was compiled from, read from its bytes (no dlopen: a stale library must not get loaded
    just to be asked), or None for a file without the marker."""
    try:
        blob = open(path, "rb").read()
    except OSError:
        return None
    i = blob.find(HASH_MARKER)
    if i < 0:
        return None
    j = i + len(HASH_MARKER)
    return blob[j:j + 64].decode("ascii", "replace")


def is_stale() -> bool:
    """True when liblc_amd.so is missing or was built from other source contents than the ones on disk now.  The hash lives INSIDE
    the library, so a copied library keeps it and there is no window in which library and hash disagree."""
    return embedded_hash(SO_PATH) != source_hash()


def hipcc_available() -> bool:
    try:
        _hipcc()
        return True
    except RuntimeError:
        return False


def _locked(fn):
    import fcntl

    os.makedirs(OUT_DIR, exist_ok=True)
    with open(SO_PATH + ".lock", "w") as lock:  # one builder at a time (pytest-xdist workers, torchrun ranks, A/B scripts)
        fcntl.flock(lock, fcntl.LOCK_EX)
        return fn()


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return SO_PATH
    return _locked(lambda: SO_PATH if (not force and not is_stale()) else _compile(SO_PATH, [], verbose))


# Extra compiler flags of single translation units.  The latency builds of the one-wave pose kernels are scheduled for the shortest
# dependent chains (a lone wave per SIMD has nobody to hide its latencies behind): -2.8 % on the metric's launch, same bits; every
# other kernel is faster (or equal) with the default strategy (lc_amd/csrc/lc_pnp_latency.hip has the measurements).
_LATENCY = ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-amdgpu-use-amdgpu-trackers=1"]  # (the register-pressure trackers: -1.3 % more on the fused launch)
_MAX_ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
PER_FILE_FLAGS = {
    "lc_pnp_latency.hip": _LATENCY,
    "lc_fused_latency.hip": _LATENCY,
    # few-waves-per-unit latency chains as well: head forward -2.6 % (fp32) / -2.8 % (bf16), wide solve -1 %, the test-time pipeline
    # 81.1 -> 79.8 us (scripts/ubench: bench_head.py, pnp_wide_ab.py, graph_inference.py; outputs unchanged)
    "lc_head.hip": _MAX_ILP,
    "lc_pnp.hip": _MAX_ILP,
    "lc_pnp_init.hip": _MAX_ILP,
    "lc_select.hip": _MAX_ILP,
    "lc_dense.hip": _MAX_ILP,
    # NOT lc_loss.hip (the loss kernel alone: +4 %) and NOT lc_fused.hip (the large-grid pose unit: +1.4 %)
}
# -ffp-contract=on: a multiply and an add are fused where the SOURCE writes them in one expression, and nowhere else.  hipcc's default
# (fast) also fuses across statements, and decides that per inlined copy of a function: the LC loss's walk body exists in a one-tile
# and a many-tiles copy, and under `fast` the two contracted differently -- two slicings of the same sample then differed in the last
# bit of an fp32 gradient for about 4 outputs in 10^8 (scripts/ubench/forms_ulp.py).  With `on` every form, slicing and translation
# unit evaluates the same expressions the same way: "bit for bit" is a property of the build, not of the inputs tried.  Cost on the
# headline launch: within the run-to-run noise (13.13 vs 13.07 us).
COMMON_FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-ffp-contract=on"]


def _compile(out: str, flags, verbose: bool) -> str:
    """One object per .hip source (in parallel; per-file flags from PER_FILE_FLAGS), then one link."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor

    hipcc, digest = _hipcc(), source_hash()
    tmp = out + f".tmp{os.getpid()}"
    with tempfile.TemporaryDirectory(prefix="lc_amd_build_") as objdir:
        def one(src):
            obj = os.path.join(objdir, os.path.basename(src) + ".o")
            cmd = [hipcc, *COMMON_FLAGS, f'-DLC_AMD_SRC_HASH="{digest}"', *flags, *PER_FILE_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            return obj

        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
            objs = list(pool.map(one, sources()))
        link = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fvisibility=hidden", *objs, "-o", tmp]
        if verbose:
            print(" ".join(link))
        try:
            subprocess.check_call(link)
            os.replace(tmp, out)  # atomic: a reader sees the old library or the new one, each with its own hash inside
        finally:
            if os.path.exists(tmp):
                os.unlink(tmp)
    return out


def _variant_dir() -> str:
    """Where experiment builds go: LC_AMD_BUILD_DIR if set; next to a source checkout (build/variants, git-ignored) when the package sits in
    one; a per-user cache otherwise (an installed package's parent directory is site-packages: nothing is written there)."""
    env = os.environ.get("LC_AMD_BUILD_DIR")
    if env:
        return env
    parent = os.path.dirname(PKG)
    if os.path.exists(os.path.join(parent, "include", "lc_amd.h")) and os.access(parent, os.W_OK):
        return os.path.join(parent, "build", "variants")
    return os.path.join(os.environ.get("XDG_CACHE_HOME", os.path.join(os.path.expanduser("~"), ".cache")), "lc_amd", "variants")


VARIANT_DIR = _variant_dir()  # diagnostics stay out of the package


def variant_path(name: str) -> str:
    return os.path.join(VARIANT_DIR, f"liblc_amd_{name}.so")


def build_variant(name: str, flags, verbose: bool = False) -> str:
    """An experiment build of the same sources with extra compiler flags (-D switches), OUTSIDE the package:
    build/variants/liblc_amd_<name>.so; select it with LC_AMD_LIB=<path>.  Used by the A/B scripts under scripts/ and by the tests
    that need a diagnostic build only."""
    os.makedirs(VARIANT_DIR, exist_ok=True)
    return _locked(lambda: _compile(variant_path(name), list(flags), verbose))


if __name__ == "__main__":
    print(build(force=True, verbose=True))
