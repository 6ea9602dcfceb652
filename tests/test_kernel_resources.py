"""What the code object tells the hardware to allocate for the hot kernels (read from liblc_amd.so with scripts/kernel_resources.py;
no GPU needed).  A kernel of the path that starts spilling to scratch is 2-3x slower on this chip (profiles/r03/occupancy.txt) and no
functional test notices: this one does."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


@pytest.fixture(scope="module")
def res():
    from lc_amd import _lib
    from kernel_resources import kernel_resources

    _lib.load()  # builds the library if the sources changed
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("llvm-readelf not available")
    return kernel_resources()


def _find(res, *parts):
    hits = [d for n, d in res.items() if all(p in n for p in parts)]
    assert hits, parts
    return hits


def test_pose_kernels_do_not_spill_and_keep_their_occupancy(res):
    # large-grid builds (more than 1024 one-wave workgroups): two waves per SIMD, eight workgroups per CU
    for parts in (("lc_pose_unit_kernelILi2E",), ("lc_pnp_lm_kernelILb1ELi2E",), ("lc_cov_loss_kernelILb1ELb0E",)):
        for d in _find(res, *parts):
            assert d.get("private_segment_fixed_size", 0) == 0, (d["name"], "scratch")
            assert d["vgpr_count"] <= 256 and d["waves_per_simd_by_registers"] >= 2, d["name"]
            assert d.get("group_segment_fixed_size", 0) <= 16 * 1024, d["name"]  # eight one-wave workgroups per CU fit the 160 KB of LDS
    # latency builds (at most 1024 workgroups = one per SIMD): one wave per SIMD is all they need, AGPRs may serve the max-ILP schedule,
    # scratch may not
    for parts in (("lc_pose_unit_kernelILi1E",), ("lc_pnp_lm_kernelILb1ELi1E",)):
        for d in _find(res, *parts):
            assert d.get("private_segment_fixed_size", 0) == 0, (d["name"], "scratch")
            assert d["vgpr_count"] <= 512 and d.get("group_segment_fixed_size", 0) <= 16 * 1024, d["name"]


def test_dense_kernels_do_not_spill(res):
    # 256-thread workgroups, one per compute unit: AGPRs may serve as spill space (cheap), scratch memory may not
    for parts in (("lc_cov_loss_tiled_kernelILb0E",), ("lc_cov_loss_kernelILb0ELb0E",), ("lc_pnp_lm_wide_kernel", "Lb0ELi4E"), ("lc_pnp_lm_wide_kernel", "Lb0ELi8E"),
                  ("lc_pnp_lm_wide_kernelILb1E",), ("lc_pnp_lm_chain_kernel",), ("lc_pose_unit_dense_kernel",)):
        for d in _find(res, *parts):
            assert d.get("private_segment_fixed_size", 0) == 0, (d["name"], "scratch")
            assert d["vgpr_count"] <= 512 and d.get("group_segment_fixed_size", 0) <= 64 * 1024, d["name"]


def test_split_solve_fits_one_workgroup_per_compute_unit(res):
    # lc_pnp_lm_split_kernel: its workgroups wait for each other, so every one of a launch must be resident -- pnp_split_parts() sizes the grid to
    # at most 256 workgroups, one per compute unit: 256 threads, registers of one wave per SIMD (VGPRs + AGPRs <= 512), a few KB of LDS, no scratch
    for d in _find(res, "lc_pnp_lm_split_kernel"):
        assert d.get("private_segment_fixed_size", 0) == 0, (d["name"], "scratch")
        assert d["vgpr_count"] <= 512 and d.get("group_segment_fixed_size", 0) <= 8 * 1024, d["name"]
    # the rescue launch behind it (round 5): it must be startable while the chip is held by others -- one 256-thread workgroup per pose, no scratch,
    # a few KB of LDS; and the scoring kernels of the RANSAC, whose division-free form must not have cost them their occupancy
    for d in _find(res, "lc_pnp_lm_split_rescue_kernel"):
        assert d.get("private_segment_fixed_size", 0) == 0 and d["vgpr_count"] <= 512 and d.get("group_segment_fixed_size", 0) <= 8 * 1024, d["name"]
    for name in ("lc_ransac_score_kernel", "lc_ransac_score_live_kernel", "lc_ransac_score_wide_kernel"):
        for d in _find(res, name):
            assert d.get("private_segment_fixed_size", 0) == 0 and d["waves_per_simd_by_registers"] >= 4, d["name"]


def test_head_kernels_keep_their_occupancy(res):
    # HBM-bound streams: the backward needs many waves in flight; the one-wave-per-map forward holds 64 values per lane and is
    # faster under the max-ILP schedule at three waves per SIMD than under the default one at four (bench_head.py: 42.9 -> 41.8 us)
    for d in _find(res, "lc_head_bwd_kernel"):
        assert d.get("private_segment_fixed_size", 0) == 0 and d["waves_per_simd_by_registers"] >= 4, d["name"]
    for d in _find(res, "lc_head_fwd_wave64_kernel"):
        assert d.get("private_segment_fixed_size", 0) == 0 and d["waves_per_simd_by_registers"] >= 3, d["name"]
