#!/usr/bin/env python3
"""Register / LDS / scratch allocation of every kernel in liblc_amd.so, read from the code object itself (the AMDGPU metadata note of
the gfx950 ELF inside the library's offload bundle): what the hardware is told to allocate, which is what bounds the occupancy --
rocprofv3's per-dispatch `VGPR_Count` column is not that number.

    python scripts/kernel_resources.py [path/to/liblc_amd.so]      -> one line per kernel
    from scripts.kernel_resources import kernel_resources          -> {demangled-ish name: dict}
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _device_elfs(blob: bytes):
    """Every gfx950 code object in the library (one offload bundle per .hip translation unit)."""
    out, i = [], blob.find(MAGIC)
    while i >= 0:
        n, = struct.unpack_from("<Q", blob, i + len(MAGIC))
        p = i + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode(errors="replace")
            p += 24 + tlen
            if "gfx950" in triple and size:
                out.append(blob[i + off:i + off + size])
        i = blob.find(MAGIC, i + len(MAGIC))
    if not out:
        raise RuntimeError("no gfx950 code object found in the library's offload bundles")
    return out


def kernel_resources(so_path=None):
    so_path = so_path or os.path.join(ROOT, "lc_amd", "_C", "liblc_amd.so")
    out = {}
    for elf in _device_elfs(open(so_path, "rb").read()):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(elf)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for ln in txt.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", ln)
            if not m:
                continue
            k, v = m.group(1), m.group(2).strip().strip("'")
            if k == "agpr_count":  # first key of a kernel record (keys are sorted)
                cur = {"agpr_count": int(v)}
            elif cur is not None and k in ("vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "max_flat_workgroup_size",
                                           "vgpr_spill_count", "sgpr_spill_count"):
                cur[k] = int(v)
            elif cur is not None and k == "name":
                cur["name"] = v
            elif cur is not None and k == "wavefront_size":  # last key of the record
                # gfx90a+: .vgpr_count is the UNIFIED allocation (architectural VGPRs + AGPRs) out of 512 per SIMD lane, granule 8
                total = (cur.get("vgpr_count", 0) + 7) // 8 * 8
                cur["arch_vgpr_count"] = cur.get("vgpr_count", 0) - cur.get("agpr_count", 0)
                cur["waves_per_simd_by_registers"] = min(8, 512 // max(total, 8))
                out[cur["name"]] = cur
                cur = None
    return out


def short_name(mangled: str) -> str:
    m = re.search(r"\d+(lc_\w+?)(I|E)", mangled)
    return m.group(1) if m else mangled


if __name__ == "__main__":
    res = kernel_resources(sys.argv[1] if len(sys.argv) > 1 else None)
    print(f"{'kernel (mangled)':100s} VGPR(arch) AGPR  SGPR  LDS B  scratch B/lane  waves/SIMD (registers)")
    for name, d in sorted(res.items()):
        print(f"{name[:100]:100s} {d['arch_vgpr_count']:10d} {d.get('agpr_count', 0):4d} {d.get('sgpr_count', 0):5d} {d.get('group_segment_fixed_size', 0):6d} "
              f"{d.get('private_segment_fixed_size', 0):8d} {d['waves_per_simd_by_registers']:10d}")
