"""Edge cases of the HIP path: empty batches, single-point problems, non-contiguous / expanded inputs, NaN handling,
independent streams, and the host-pointer ABI called from several threads."""
import ctypes
import threading

import numpy as np
import pytest
import torch

from lc_amd import synth
from tests.util import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_empty_batches_are_no_ops():
    from lc_amd.cov_mixed import loss_cov_mixed_fused
    from lc_amd.pnp import pnp_ceres
    from lc_amd.ptnet import spatial_softargmax_2d_std

    dev = torch.device(DEV)
    b = {k: v.to(dev)[:0] for k, v in synth.make_batch(2, 8, seed=0).items()}
    loss, du, ds, dx, _ = loss_cov_mixed_fused(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], None, b["bbox_3d"])
    assert loss.shape == (0,) and du.shape == (0, 8, 2)
    st, tr, ret = pnp_ceres.solve_device(b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"])
    assert st.shape == (0, 7) and ret.shape == (0,)
    m, s = spatial_softargmax_2d_std(torch.empty(0, 4, 16, 16, device=dev))
    assert m.shape == (0, 4, 2)


def test_single_point_loss_matches_oracle():
    from lc_amd.cov_mixed import loss_cov_mixed_fused
    from oracle import lc_loss_oracle as orc

    b = synth.make_batch(4, 1, seed=3)
    d = {k: v.to(DEV) for k, v in b.items()}
    loss, du, ds, dx, _ = loss_cov_mixed_fused(d["K"], d["pose"], d["pts3d"], d["pts2d"], d["inv_std"], None, d["bbox_3d"])
    b64 = {k: v.double() for k, v in b.items()}
    rl, ru, rs, rx = orc.loss_and_grads(b64["K"], b64["pose"], b64["pts3d"], b64["pts2d"], b64["inv_std"], None, b64["bbox_3d"])
    # N=1: H has rank 2 -> the SPD fallback H := I is taken by both (first pivot > 0, a later one is not)
    assert torch.isfinite(loss).all()
    assert ((loss.cpu().double() - rl).abs() / rl.abs().clamp_min(1)).max() <= 1e-5
    assert rel_err(ds.cpu(), rs) <= 1e-4


def test_non_contiguous_and_expanded_inputs():
    """Loss_cov_mixed receives expand()ed grids and .mT views from the dense front end of the reference (losses.py:142-161)."""
    from lc_amd.cov_mixed import Loss_cov_mixed
    from oracle import lc_loss_oracle as orc

    b = synth.make_batch(3, 12, seed=4)
    d = {k: v.to(DEV) for k, v in b.items()}
    grid = d["pts2d"][0].clone()                       # one grid shared by the batch -> expand (stride 0)
    u = grid.unsqueeze(0).expand(3, 12, 2)
    s_t = d["inv_std"].mT.contiguous().mT.requires_grad_(True)   # (B,N,2) view of a (B,2,N) buffer
    X_t = d["pts3d"].mT.contiguous().mT.requires_grad_(True)
    K1 = d["K"][:1].expand(3, 3, 3)
    loss = Loss_cov_mixed(K1, d["pose"], X_t, u, s_t, torch.ones(3, 12, device=DEV), bbox_3d=d["bbox_3d"][:1].expand(3, 8, 3))
    gs, gx = torch.autograd.grad(loss.sum(), [s_t, X_t])
    b64 = {k: v.double() for k, v in b.items()}
    rl, _, rs, rx = orc.loss_and_grads(b64["K"][:1].expand(3, 3, 3), b64["pose"], b64["pts3d"], b64["pts2d"][:1].expand(3, 12, 2),
                                       b64["inv_std"], torch.ones(3, 12, dtype=torch.float64), b64["bbox_3d"][:1].expand(3, 8, 3))
    assert ((loss.detach().cpu().double() - rl).abs() / rl.abs().clamp_min(1)).max() <= 1e-5
    assert rel_err(gs.cpu(), rs) <= 1e-4 and rel_err(gx.cpu(), rx) <= 1e-4


def test_nan_inputs_follow_the_reference_contract():
    """cer_solver(filter_input_nan=True) zeroes NaNs (cer_solver.py:27-29); without filtering a NaN job comes back invalid and
    keeps its start pose (ceres would fail to evaluate: FAILURE != CONVERGENCE, ceres.cpp:134-138)."""
    from lc_amd.pnp import cer_solver

    b = synth.make_batch(4, 32, seed=5)
    d = {k: v.to(DEV) for k, v in b.items()}
    u = d["pts2d"].clone()
    u[1, 3, 0] = float("nan")
    inv, st = cer_solver.solve(d["K"], d["pts3d"], u, d["inv_std"] ** 2, d["start"], filter_input_nan=False)
    assert inv["invalids"].tolist() == [False, True, False, False] and torch.equal(st[1], d["start"][1])
    inv2, st2 = cer_solver.solve(d["K"], d["pts3d"], u, d["inv_std"] ** 2, d["start"], filter_input_nan=True)
    assert not inv2["invalids"].any() and torch.isfinite(st2).all()


def test_kernels_follow_the_current_stream():
    """Launches go to torch's current stream: two side streams produce the same results as the default stream."""
    from lc_amd.cov_mixed import loss_cov_mixed_fused

    dev = torch.device(DEV)
    b = {k: v.to(dev) for k, v in synth.make_batch(64, 64, seed=6).items()}
    ref = loss_cov_mixed_fused(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], None, b["bbox_3d"])[0]
    torch.cuda.synchronize()
    outs = []
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st):
            outs.append(loss_cov_mixed_fused(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], None, b["bbox_3d"])[0])
    for st in streams:
        st.synchronize()
    assert all(torch.equal(o, ref) for o in outs)


def test_host_abi_is_thread_safe():
    """`pnp_ceres_f32_omp` shares one staging workspace behind a mutex: concurrent callers get independent, correct answers."""
    from lc_amd import _lib
    from oracle import pnp_oracle

    lib = ctypes.CDLL(_lib.lib_path())
    results = {}

    def work(tid):
        b = synth.make_batch(16 + tid, 24 + 8 * tid, seed=10 + tid)
        L = torch.diag_embed(b["inv_std"]).numpy()
        n = b["pts3d"].shape[1]
        results[tid] = (pnp_oracle.solve_pointer_arrays(list(b["start"].numpy()), list(b["K"].numpy()), list(b["pts2d"].numpy()),
                                                        list(b["pts3d"].numpy()), list(L), [n] * len(L), symbol_lib=lib),
                        pnp_oracle.solve_batched(b["start"].numpy(), b["K"].numpy(), b["pts2d"].numpy(), b["pts3d"].numpy(), L))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    [t.start() for t in threads]
    [t.join() for t in threads]
    for tid, ((s_gpu, _, r_gpu), (s_cpu, _, r_cpu)) in results.items():
        np.testing.assert_array_equal(r_gpu, r_cpu)
        ok = r_cpu == 0
        assert np.abs(s_gpu[ok][:, :4] - s_cpu[ok][:, :4]).max() <= 1e-4


def test_pnp_dense_sizes_vs_oracle():
    """N = 1849 (zlmo: 128^2 / 3^2 rounded up) and ragged counts through the wave-stride path."""
    from lc_amd.pnp import pnp_ceres
    from oracle import pnp_oracle

    b = synth.make_batch(3, 1849, seed=8)
    counts = torch.tensor([1849, 1000, 65], dtype=torch.int32)
    L = torch.diag_embed(b["inv_std"])
    st, tr, ret = pnp_ceres.solve_device(b["K"].to(DEV), b["pts3d"].to(DEV), b["pts2d"].to(DEV), L.to(DEV), b["start"].to(DEV), counts)
    so, _, ro = pnp_oracle.solve_batched(b["start"].numpy(), b["K"].numpy(), b["pts2d"].numpy(), b["pts3d"].numpy(), L.numpy(),
                                         counts=counts.numpy())
    np.testing.assert_array_equal(ret.cpu().numpy(), ro)
    assert np.abs(st.cpu().numpy()[:, :4] - so[:, :4]).max() <= 1e-4


def test_misaligned_pointers_are_rejected():
    """The C ABI refuses a float2-row pointer at an odd element offset instead of faulting in a vector load."""
    from lc_amd import _lib, synth

    lib = _lib.load()
    dev = torch.device("cuda:0")
    b = {k: v.to(dev) for k, v in synth.make_batch(2, 8, seed=0).items()}
    flat = torch.zeros(2 * 8 * 2 + 1, device=dev)
    odd = flat[1:]  # 4-byte offset: not 8-byte aligned
    st, tr, ret = torch.empty_like(b["start"]), torch.empty(2, device=dev), torch.empty(2, device=dev, dtype=torch.int32)
    P = _lib.ptr
    rc = lib.lc_pnp_lm3_f32(P(b["K"]), P(b["pts3d"]), P(odd), None, P(b["inv_std"]), None, None, P(b["start"]), P(st), P(tr), P(ret), None, 2, 8,
                            50, 1e-6, 0, 0, None, 0, _lib.stream_ptr(dev))
    assert rc == 1 and b"aligned" in lib.lc_amd_last_error()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_half_precision_keypoint_tensors_get_gradients_in_their_own_dtype(dtype):
    """bf16 / fp16 heads (mixed-precision training): same loss as fp32 on the rounded values, gradients return in the head's dtype."""
    from lc_amd import synth
    from lc_amd.cov_mixed import Loss_cov_mixed

    dev = torch.device("cuda:0")
    b = {k: v.to(dev) for k, v in synth.make_batch(4, 32, seed=5).items()}
    u16 = b["pts2d"].to(dtype).requires_grad_(True)
    s16 = b["inv_std"].to(dtype).requires_grad_(True)
    l16 = Loss_cov_mixed(b["K"], b["pose"], b["pts3d"], u16, s16, None, bbox_3d=b["bbox_3d"])
    gu16, gs16 = torch.autograd.grad(l16.sum(), (u16, s16))
    u32 = u16.detach().float().requires_grad_(True)
    s32 = s16.detach().float().requires_grad_(True)
    l32 = Loss_cov_mixed(b["K"], b["pose"], b["pts3d"], u32, s32, None, bbox_3d=b["bbox_3d"])
    gu32, gs32 = torch.autograd.grad(l32.sum(), (u32, s32))
    assert l16.dtype == torch.float32 and torch.equal(l16, l32)
    assert gu16.dtype == dtype and gs16.dtype == dtype
    assert torch.equal(gu16, gu32.to(dtype)) and torch.equal(gs16, gs32.to(dtype))
