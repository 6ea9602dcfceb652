"""The launch forms whose workgroups wait for each other (several workgroups per pose: `lc_pnp_lm3_f32`; per object:
`lc_dense_frontend_select3`) while SOMETHING ELSE holds compute units: a second stream, another process, a DDP evaluation's RCCL kernels.

Contract: a valid pose never depends on scheduling.  A part that has waited in vain stops, the launch that
always follows (the rescue launch) re-zeroes the unit's exchange region and computes the unit with one workgroup that plays the parts in
turn -- the SAME bits the parts produce when they meet -- so results under contention are bit-identical to the undisturbed call, `rets`
keeps meaning "did not converge" and nothing else (ceres.cpp:134-138, cer_solver.py:51-52), and the workspace is good for the next launch.

`tests/native/occupy.hip` (a test helper, not part of the product) holds compute units for a chosen time, in one of two ways: "waves" --
1024-thread workgroups, 16 of a compute unit's 32 wave slots each, two per held unit (what is left over admits some 256-thread workgroups
of a launch and not the others); "lds" -- one workgroup with all 160 KB of LDS per held unit (nothing that uses LDS starts there; the
1024-thread selection workgroups, which do not fit next to a "waves" holder at all, get exactly the free units).  Workgroups are dispatched
in grid order, round-robin over the XCDs, so what a launch sees under contention is a sliding window of its grid: the parts of a unit
meet when the window holds them all, and time out when it cannot."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

from lc_amd import _lib, synth
from lc_amd.pnp import pnp_ceres

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"
PNP_POSE_BYTES = 2 * 8 * 64 * 8 + 128   # lc_common.h kSplitPoseBytes
SEL_POSE_BYTES = 2 * 8 * 256 * 8 + 128  # lc_select.hip kSelSplitPoseBytes


@pytest.fixture(scope="module")
def occupy():
    so = os.path.join(ROOT, "build", "tests", "liboccupy.so")
    src = os.path.join(ROOT, "tests", "native", "occupy.hip")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
        if not os.path.exists(hipcc):
            pytest.skip("hipcc not available and build/tests/liboccupy.so not prebuilt")
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so], check=True, capture_output=True, timeout=600)
    lib = ctypes.CDLL(so)
    lib.occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_longlong, ctypes.c_void_p]
    lib.occupy.restype = ctypes.c_int
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    side = torch.cuda.Stream()

    def hold(free_cus: int, ms: float, how: str = "waves"):
        """All of the chip but `free_cus` compute units, for `ms` milliseconds, on a side stream."""
        if how == "waves":
            rc = lib.occupy(2 * (cus - free_cus), 1024, 0, int(ms * 1e5), ctypes.c_void_p(side.cuda_stream))
        else:
            rc = lib.occupy(cus - free_cus, 256, 160 * 1024, int(ms * 1e5), ctypes.c_void_p(side.cuda_stream))
        assert rc == 0
        return side

    hold.cus = cus
    return hold


def _tail(ws, B, pose_bytes):
    """(epoch, dirty, rescues) words of every unit's region."""
    t = ws[:B * pose_bytes].view(torch.int32).view(B, pose_bytes // 4)[:, -32:-29]
    return t[:, 0].clone(), t[:, 1].clone(), t[:, 2].clone()


def _pnp_batch(B, N, seed):
    b = {k: v.to(DEV) for k, v in synth.make_batch(B, N, seed=seed, outlier_frac=0.05, noise_px=0.7).items()}
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(2100, N + 1, (B,), generator=g, dtype=torch.int32)
    counts[1] = 2  # fewer than three correspondences: invalid without a solve
    return b, counts.to(DEV)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("B,N", [(64, 4096), (24, 3000), (128, 2600)])  # 4, 8 and 2 workgroups per pose
def test_split_solve_is_bit_identical_when_its_parts_cannot_all_be_resident(occupy, B, N):
    b, counts = _pnp_batch(B, N, seed=B)
    args = (b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"], counts)
    ws = pnp_ceres.split_workspace(torch.device(DEV), (B, N))
    want = pnp_ceres.solve_device(*args, return_iters=True, split=True)
    torch.cuda.synchronize()
    _, dirty0, resc0 = _tail(ws, B, PNP_POSE_BYTES)
    assert int(dirty0.sum()) == 0
    # room for a quarter of the launch's workgroups only: the others cannot start before the first have given up
    parts = next(p for p in (8, 4, 2) if (B + 7) // 8 * 8 * p <= occupy.cus)  # lc_pnp.hip: pnp_split_parts
    # the parts of a pose sit 8 workgroups apart in the grid (one XCD per pose): the window the free units admit must be narrower than 8 x parts
    free = max(2, (B * parts) // 4 // 8) if parts > 2 else 1  # (a free compute unit takes two of these 256-register workgroups)
    rescued = 0
    for attempt in range(4):  # (the helper and the launch under test race for the dispatcher: if the launch got in first, once more)
        side = occupy(free, 60.0)
        got = pnp_ceres.solve_device(*args, return_iters=True, split=True)
        torch.cuda.synchronize()
        for name, x, y in zip(("states", "result_tr", "rets", "iters"), got, want):
            assert torch.equal(x, y), (name, attempt)
        assert int(got[2].max()) <= 1, "the internal 'a part never arrived' status never leaves the launch pair"
        _, dirty, resc = _tail(ws, B, PNP_POSE_BYTES)
        assert int(dirty.sum()) == 0, "the rescue launch leaves every region clean"
        rescued = int((resc - resc0).sum())
        if rescued > 0:
            break
        side.synchronize()
    print(f"B={B} N={N}: {rescued} of {B} poses solved by the rescue launch while {occupy.cus - free} compute units were held")
    assert rescued > 0, "the contention did not bite: the test does not exercise the rescue path"
    # and the workspace is good for the next launch, which meets again (no rescue)
    again = pnp_ceres.solve_device(*args, return_iters=True, split=True)
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(again, want))
    side.synchronize()
    resc1 = _tail(ws, B, PNP_POSE_BYTES)[2]
    again = pnp_ceres.solve_device(*args, return_iters=True, split=True)
    torch.cuda.synchronize()
    _, dirty2, resc2 = _tail(ws, B, PNP_POSE_BYTES)
    assert all(torch.equal(x, y) for x, y in zip(again, want)) and int(dirty2.sum()) == 0
    assert torch.equal(resc2, resc1), "with the chip to itself the launch needs no rescue"


def _select_inputs(B, H, W, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    blob = (((yy - H / 2) / (0.3 * H)) ** 2 + ((xx - W / 2) / (0.25 * W)) ** 2 < 1).float()
    xyz = torch.randn(B, 3, H, W, generator=g).to(DEV, dtype)
    wl = (torch.randn(B, 2, H, W, generator=g) * 1.5 + 4 * blob).to(DEV, dtype)
    vl = ((blob * 2 - 1) * 3 + torch.randn(B, 1, H, W, generator=g))
    vl[1] = -5.0
    return xyz, wl, (torch.rand(B, generator=g) + 0.5).to(DEV), (torch.rand(B, 3, generator=g) + 0.5).to(DEV), vl.to(DEV, dtype)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode,q", [("quantile_in_mask", 0.2), ("quantile", 0.3)])
def test_split_selection_is_bit_identical_when_its_parts_cannot_all_be_resident(occupy, mode, q):
    from lc_amd import splitws
    from lc_amd.dense import dense_front_end_select

    B, H, W = 64, 128, 128
    xyz, wl, ws_, ns, vl = _select_inputs(B, H, W, seed=5)
    kw = dict(seg_thresh=0.5, sample=1, quantile=q, min_count=6, seed=3)
    want = dense_front_end_select(xyz, wl, ws_, ns, vl, mode, split=True, **kw)
    torch.cuda.synchronize()
    work = splitws.get("select", torch.device(DEV), _lib.load().lc_dense_frontend_select_workspace_bytes(B, H, W, 0, 0, 1), True)
    _, _, resc0 = _tail(work, B, SEL_POSE_BYTES)
    cnt = want[3]
    live = torch.arange(want[0].shape[1], device=DEV)[None, :] < cnt[:, None]
    rescued = 0
    for attempt in range(4):  # (the helper and the launch under test race for the dispatcher: if the launch got in first, once more)
        side = occupy(2, 100.0, "lds")  # two compute units left: never the four parts of an object at once
        got = dense_front_end_select(xyz, wl, ws_, ns, vl, mode, split=True, **kw)
        torch.cuda.synchronize()
        assert torch.equal(got[3], cnt)
        for x, y in zip(got, want):
            if x.dim() >= 2:
                m = live if x.dim() == 2 else live[..., None].expand_as(x)
                assert torch.equal(x[m], y[m])
        _, dirty, resc = _tail(work, B, SEL_POSE_BYTES)
        assert int(dirty.sum()) == 0
        rescued = int((resc - resc0).sum())
        if rescued > 0:
            break
        side.synchronize()
    print(f"{mode}: {rescued} of {B} objects selected by the rescue launch")
    assert rescued > 0
    side.synchronize()
    again = dense_front_end_select(xyz, wl, ws_, ns, vl, mode, split=True, **kw)
    torch.cuda.synchronize()
    assert torch.equal(again[3], cnt) and torch.equal(again[4][live], want[4][live])
    assert torch.equal(_tail(work, B, SEL_POSE_BYTES)[2], resc), "with the chip to itself the launch needs no rescue"


def _to(d, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in d.items()}


@pytest.mark.timeout(600)
@pytest.mark.parametrize("filler", ["held-units-waves", "held-units-lds", "matmul-stream", "head-backward-stream"])
def test_zlmo_chain_loses_no_pose_under_contention(occupy, filler):
    """zlmo's whole test-time chain (64 objects x 16 384 candidates: the split selection and two split solves are on its path) eager and as a
    replayed hipGraph, while (a) all but 24 / all but 3 compute units are held for 80 ms, (b) a stream of bf16 matrix products or (c) the
    keypoint head's backward in a loop competes for the chip: poses bit-identical to the undisturbed call, every time."""
    from lc_amd.config import AttrDict
    from lc_amd.inference import GraphedSolvePnP, solve_pnp

    cfg, gt_c, out_c = synth.test_time_inputs("zlmo", B=64, seed=5)
    cfg = AttrDict(cfg)
    gt, out = _to(gt_c, DEV), _to(out_c, DEV)
    want = solve_pnp(cfg, out, gt)["weighted-filtered"].clone()
    solver = GraphedSolvePnP(cfg, out, gt)
    assert torch.equal(solver(out, gt)["weighted-filtered"], want)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    if filler == "matmul-stream":
        a = torch.randn(8192, 8192, device=DEV, dtype=torch.bfloat16)
    elif filler == "head-backward-stream":
        from lc_amd.ptnet import spatial_softargmax_2d_std
        logits = synth.make_head_logits(64, 64, 64, 64, seed=1).to(DEV).requires_grad_(True)

    def disturb():
        if filler == "held-units-waves":
            return occupy(24, 80.0)
        if filler == "held-units-lds":
            return occupy(3, 80.0, "lds")
        with torch.cuda.stream(side):
            for _ in range(40):
                if filler == "matmul-stream":
                    a @ a
                else:
                    mean, std = spatial_softargmax_2d_std(logits)
                    torch.autograd.grad((mean.sum() + std.sum()), logits)
        return side

    for rep in range(3):
        s = disturb()
        eager = solve_pnp(cfg, out, gt)["weighted-filtered"]
        replay = solver(out, gt)["weighted-filtered"].clone()
        torch.cuda.synchronize()
        assert torch.equal(eager, want), (filler, rep, "eager", int((eager != want).any(1).sum()))
        assert torch.equal(replay, want), (filler, rep, "graph", int((replay != want).any(1).sum()))
        s.synchronize()
    assert torch.equal(solve_pnp(cfg, out, gt)["weighted-filtered"], want)
