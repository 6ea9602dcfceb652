"""GPU parity of the batched LM weighted-PnP kernel against oracle/pnp_lm_oracle.c, through both C-ABI routes.

Tolerance (north_star / BASELINE.md 3.6): max|dq| <= 1e-4 after sign alignment and ||dt||/||t|| <= 1e-4; both solvers follow
the same schedule, so trust radius and invalid flags must agree too.
"""
import ctypes

import numpy as np
import pytest
import torch

from lc_amd import synth
from oracle import pnp_oracle

pytestmark = pytest.mark.gpu


def pose_err(a, b):
    qa = a[:, :4] / np.linalg.norm(a[:, :4], axis=1, keepdims=True)
    qb = b[:, :4] / np.linalg.norm(b[:, :4], axis=1, keepdims=True)
    sgn = np.sign((qa * qb).sum(1, keepdims=True))
    dq = np.abs(qa - sgn * qb).max(1)
    dt = np.linalg.norm(a[:, 4:] - b[:, 4:], axis=1) / np.linalg.norm(b[:, 4:], axis=1)
    return dq, dt


def make(B, N, seed, **kw):
    b = synth.make_batch(B, N, seed=seed, **kw)
    return b, torch.diag_embed(b["inv_std"])


@pytest.mark.parametrize("B,N,seed", [(256, 64, 0), (16, 16, 1), (8, 4, 2), (5, 100, 3), (3, 1024, 4), (3, 1849, 5), (2, 2500, 6), (2, 4096, 7),
                                      (4, 65, 8), (3, 255, 9), (16, 256, 10), (3, 257, 11)])
def test_pnp_device_route_vs_oracle(B, N, seed):
    """(the last four: either side of the one-wavefront / four-wavefront switch at N = 64 and of four correspondences per thread at N = 256;
    B=16, N=256 is BASELINE configs[0]'s shape: 16 crops, 32x32 maps, stride 2)"""
    from lc_amd.pnp import pnp_ceres

    b, L = make(B, N, seed)
    dev = torch.device("cuda:0")
    st, tr, ret, iters = pnp_ceres.solve_device(b["K"].to(dev), b["pts3d"].to(dev), b["pts2d"].to(dev), L.to(dev), b["start"].to(dev),
                                                return_iters=True)
    so, tro, reto = pnp_oracle.solve_batched(b["start"].numpy(), b["K"].numpy(), b["pts2d"].numpy(), b["pts3d"].numpy(), L.numpy())
    np.testing.assert_array_equal(ret.cpu().numpy(), reto)
    dq, dt = pose_err(st.cpu().numpy(), so)
    assert dq.max() <= 1e-4 and dt.max() <= 1e-4, (dq.max(), dt.max())
    assert np.allclose(tr.cpu().numpy(), tro, rtol=1e-3)
    assert (iters.cpu().numpy() >= 1).all()
    # diagonal-factor entry point == full-factor entry point
    st2, _, _ = pnp_ceres.solve_device(b["K"].to(dev), b["pts3d"].to(dev), b["pts2d"].to(dev), b["inv_std"].to(dev), b["start"].to(dev))
    assert torch.equal(st2, st)


def test_reference_abi_symbol_vs_oracle():
    """Drive liblc_amd.so's `pnp_ceres_f32_omp` with the same pointer-array marshalling the reference's cffi code uses
    (ragged counts, a <3-point job), and compare with the oracle's implementation of the same symbol."""
    from lc_amd import _lib

    b, L = make(12, 48, 5)
    counts = [48, 40, 2, 17, 48, 3, 48, 48, 0, 48, 31, 48]
    lists = lambda t: [t[i, :max(counts[i], 1)].numpy() for i in range(12)]
    args = ([s for s in b["start"].numpy()], [k for k in b["K"].numpy()], lists(b["pts2d"]), lists(b["pts3d"]), lists(L), counts)
    s_gpu, tr_gpu, ret_gpu = pnp_oracle.solve_pointer_arrays(*args, num_threads=4, symbol_lib=ctypes.CDLL(_lib.lib_path()))
    s_cpu, tr_cpu, ret_cpu = pnp_oracle.solve_pointer_arrays(*args, num_threads=4)
    np.testing.assert_array_equal(ret_gpu, ret_cpu)
    assert ret_gpu[2] == 1 and ret_gpu[8] == 1 and tr_gpu[2] == 1 and tr_gpu[8] == 1
    np.testing.assert_array_equal(s_gpu[[2, 8]], b["start"].numpy()[[2, 8]])  # untouched
    ok = ret_cpu == 0
    dq, dt = pose_err(s_gpu[ok], s_cpu[ok])
    assert dq.max() <= 1e-4 and dt.max() <= 1e-4


def test_cer_solver_surface():
    """lib.pnp.cer_solver.solve semantics: ragged lists, NaN filtering, invalid -> start, full 2x2 information."""
    from lc_amd.pnp import cer_solver

    dev = torch.device("cuda:0")
    b, L = make(6, 32, 6)
    d = {k: v.to(dev) for k, v in b.items()}
    n = [32, 20, 32, 2, 32, 9]
    pts3d = [d["pts3d"][i, :n[i]] for i in range(6)]
    pts2d = [d["pts2d"][i, :n[i]] for i in range(6)]
    icov = [(d["inv_std"][i, :n[i]] ** 2) for i in range(6)]
    pts2d[0] = pts2d[0].clone()
    pts2d[0][3, 0] = float("nan")
    inv, states = cer_solver.solve(d["K"], pts3d, pts2d, icov, list(d["start"]), num_workers=4, filter_input_nan=True)
    assert states.shape == (6, 7) and states.device.type == "cuda"
    assert inv["invalids"].tolist()[3] is True and torch.equal(states[3], d["start"][3])
    # oracle on the same padded problem
    P = 32
    pad = lambda lst, w: torch.stack([torch.cat((t, t.new_zeros((P - len(t),) + t.shape[1:]))) for t in lst])
    u_p = torch.nan_to_num(pad(pts2d, 2)).cpu()
    so, _, reto = pnp_oracle.solve_batched(b["start"].numpy(), b["K"].numpy(), u_p.numpy(), pad(pts3d, 3).cpu().numpy(),
                                           torch.diag_embed(pad(icov, 2).sqrt()).cpu().numpy(), counts=np.array(n, np.int32))
    np.testing.assert_array_equal(inv["solver_invalids"].cpu().numpy(), reto.astype(bool))
    ok = reto == 0
    dq, dt = pose_err(states.cpu().numpy()[ok], so[ok])
    assert dq.max() <= 1e-4 and dt.max() <= 1e-4
    # full 2x2 information matrices
    g = torch.Generator().manual_seed(1)
    M = torch.randn(6, 32, 2, 2, generator=g)
    icov_full = (M @ M.mT + 0.5 * torch.eye(2))
    inv2, st2 = cer_solver.solve(d["K"], d["pts3d"], d["pts2d"], icov_full.to(dev), d["start"])
    so2, _, ret2 = pnp_oracle.solve_batched(b["start"].numpy(), b["K"].numpy(), b["pts2d"].numpy(), b["pts3d"].numpy(),
                                            torch.linalg.cholesky(icov_full).numpy())
    np.testing.assert_array_equal(inv2["invalids"].cpu().numpy(), ret2.astype(bool))
    dq, dt = pose_err(st2.cpu().numpy()[ret2 == 0], so2[ret2 == 0])
    assert dq.max() <= 1e-4 and dt.max() <= 1e-4
    # optimal_start short-circuit (cer_solver.py:33-34)
    inv3, st3 = cer_solver.solve(d["K"], d["pts3d"], d["pts2d"], d["inv_std"] ** 2, d["start"], optimal_start=True)
    assert torch.equal(st3, d["start"]) and not inv3["invalids"].any()


def test_pnp_full_size_properties():
    """B=4096 x N=64: (a) noise-free correspondences recover the GT pose from the perturbed start; (b) re-solving from
    the solution is idempotent (converges immediately, pose moves < 1e-5); (c) max_iter=1 -> every job invalid, untouched."""
    from lc_amd.pnp import pnp_ceres

    dev = torch.device("cuda:0")
    b = synth.make_batch(4096, 64, seed=21, outlier_frac=0.0, noise_px=0.0)
    d = {k: v.to(dev) for k, v in b.items()}
    st, tr, ret = pnp_ceres.solve_device(d["K"], d["pts3d"], d["pts2d"], d["inv_std"], d["start"])
    assert int(ret.sum()) == 0
    dq, dt = pose_err(st.cpu().numpy(), b["pose"].numpy())
    assert dq.max() <= 5e-5 and dt.max() <= 5e-5
    st2, _, ret2 = pnp_ceres.solve_device(d["K"], d["pts3d"], d["pts2d"], d["inv_std"], st)
    assert int(ret2.sum()) == 0
    dq, dt = pose_err(st2.cpu().numpy(), st.cpu().numpy())
    assert dq.max() <= 1e-5 and dt.max() <= 1e-5
    b = synth.make_batch(512, 64, seed=22)
    d = {k: v.to(dev) for k, v in b.items()}
    st3, _, ret3 = pnp_ceres.solve_device(d["K"], d["pts3d"], d["pts2d"], d["inv_std"], d["start"], max_iter_count=1)
    assert int(ret3.sum()) == 512 and torch.equal(st3, d["start"])


def test_pnp_stress_hard_starts_vs_oracle():
    """2048 hard problems (12 points, 2 px noise, 20 % outliers, starts 0.3 rad / 10 % off): many LM iterations, rejected
    steps, NO_CONVERGENCE exits.  Trajectories of up to 50 iterations through ill-conditioned steps amplify last-bit
    differences: ANY two correct implementations differ in `rets` on ~0.04 % of such jobs -- the DENSE_QR oracle against
    itself with the column sums accumulated in reverse order flips 0.038 %, normal equations 0.042 %, double-double normal
    equations 0.040 % (profiles/r02/pnp_flip_rates.txt) -- so exact equality is not a property of the algorithm here.
    Bound: at most 4 of 2048 flags differ (0.2 %), >= 99.8 % of the jointly accepted poses within the 1e-4 tolerance;
    tests/test_gpu_pnp_trace.py pins the schedule itself row by row."""
    from lc_amd.pnp import pnp_ceres

    B, N = 2048, 12
    b = synth.make_batch(B, N, seed=31, outlier_frac=0.2, noise_px=2.0)
    g = torch.Generator().manual_seed(32)
    rv = torch.randn(B, 3, generator=g) * 0.3
    ang = rv.norm(dim=-1, keepdim=True)
    dq = torch.cat(((ang / 2).cos(), rv / ang * (ang / 2).sin()), -1)
    q = b["pose"][:, :4]
    qs = torch.stack((q[:, 0] * dq[:, 0] - (q[:, 1:] * dq[:, 1:]).sum(-1),
                      q[:, 0] * dq[:, 1] + q[:, 1] * dq[:, 0] + q[:, 2] * dq[:, 3] - q[:, 3] * dq[:, 2],
                      q[:, 0] * dq[:, 2] - q[:, 1] * dq[:, 3] + q[:, 2] * dq[:, 0] + q[:, 3] * dq[:, 1],
                      q[:, 0] * dq[:, 3] + q[:, 1] * dq[:, 2] - q[:, 2] * dq[:, 1] + q[:, 3] * dq[:, 0]), -1)
    start = torch.cat((qs, b["pose"][:, 4:] * (1 + 0.1 * torch.randn(B, 3, generator=g))), -1).float().contiguous()
    L = torch.diag_embed(b["inv_std"])
    dev = torch.device("cuda:0")
    st, tr, ret, iters = pnp_ceres.solve_device(b["K"].to(dev), b["pts3d"].to(dev), b["pts2d"].to(dev), L.to(dev), start.to(dev),
                                                return_iters=True)
    so, tro, reto = pnp_oracle.solve_batched(start.numpy(), b["K"].numpy(), b["pts2d"].numpy(), b["pts3d"].numpy(), L.numpy(), num_threads=8)
    ret = ret.cpu().numpy()
    assert iters.max().item() > 10  # the batch does contain long solves
    print(f"stress: {int((ret != reto).sum())} of {B} flags differ")
    assert (ret != reto).sum() <= 4, (ret != reto).sum()
    both = (ret == 0) & (reto == 0)
    dq_, dt_ = pose_err(st.cpu().numpy()[both], so[both])
    ok = (dq_ <= 1e-4) & (dt_ <= 1e-4)
    assert ok.mean() >= 0.998, (1 - ok.mean(), np.sort(dq_)[-5:])
    # invalid jobs keep their start pose bit for bit
    assert np.array_equal(st.cpu().numpy()[ret == 1], start.numpy()[ret == 1])


@pytest.mark.parametrize("scale", [0.0, 1e-12, 1e-9, 1e-6, 1e-3])
def test_pnp_identity_and_tiny_rotations_vs_oracle(scale):
    """Starts and optima at / near the identity rotation: the small-angle branches of QuaternionToAngleAxis,
    AngleAxisRotatePoint and AngleAxisToQuaternion (ceres.cpp:37,96,131) agree with the oracle to the last bit or two."""
    from lc_amd.pnp import pnp_ceres

    B, N = 256, 24
    g = torch.Generator().manual_seed(5)
    b = synth.make_batch(B, N, seed=3, noise_px=0.5, outlier_frac=0.0, rotate_K=False)
    q = torch.cat((torch.ones(B, 1), torch.randn(B, 3, generator=g) * scale), -1)
    q = q / q.norm(dim=-1, keepdim=True)
    pose = torch.cat((q, b["pose"][:, 4:]), -1)
    w, x, y, z = q.double().unbind(-1)
    R = torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                     2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)), -1).reshape(B, 3, 3)
    xf = (b["pts3d"].double() @ R.mT + pose[:, None, 4:].double()) @ b["K"].double().mT
    u = (xf[..., :2] / xf[..., 2:3] + 0.3 * torch.randn(B, N, 2, generator=g).double()).float()
    start = pose.clone()
    start[:, 4:] *= 1.01
    dev = torch.device("cuda:0")
    st, tr, ret = pnp_ceres.solve_device(b["K"].to(dev), b["pts3d"].to(dev), u.to(dev), b["inv_std"].to(dev), start.to(dev))
    so, tro, reto = pnp_oracle.solve_batched(start.numpy(), b["K"].numpy(), u.numpy(), b["pts3d"].numpy(),
                                             torch.diag_embed(b["inv_std"]).numpy(), num_threads=8)
    assert np.array_equal(ret.cpu().numpy(), reto) and reto.sum() == 0
    d = np.abs(st.cpu().numpy() - so)
    assert np.isfinite(st.cpu().numpy()).all() and d[:, :4].max() <= 1e-9 and d[:, 4:].max() <= 1e-4


def test_pnp_degenerate_jobs_vs_oracle():
    """Zero weights, an object behind the camera, coincident / collinear points, 500 px outliers, a NaN coordinate: the kernel
    and the oracle agree on which jobs are invalid; invalid jobs keep their start; valid ones agree within the tolerance."""
    from lc_amd.pnp import pnp_ceres

    B, N = 12, 20
    b = synth.make_batch(B, N, seed=77, noise_px=0.5, outlier_frac=0.0)
    K, X, u, s, start = (b[k].clone() for k in ("K", "pts3d", "pts2d", "inv_std", "start"))
    s[0] = 0.0                                   # zero information: singular normal equations
    start[1, 6] = -start[1, 6]                   # start behind the camera
    X[2] = X[2, :1]                              # all points coincide
    X[3] = X[3, :1] + torch.linspace(0, 1, N)[:, None] * torch.tensor([30.0, 10.0, -20.0])  # collinear points
    u[4, ::2] += 500.0                           # half of the points are gross outliers
    u[5, 3, 0] = float("nan")                    # a NaN measurement (cer_solver filters these; the raw solver must not crash)
    s[6, :, 1] = 0.0                             # information on u only
    X[7] = 0.0                                   # every point at the origin
    L = torch.diag_embed(s)
    dev = torch.device("cuda:0")
    st, tr, ret = pnp_ceres.solve_device(K.to(dev), X.to(dev), u.to(dev), L.to(dev), start.to(dev))
    so, tro, reto = pnp_oracle.solve_batched(start.numpy(), K.numpy(), u.numpy(), X.numpy(), L.numpy(), num_threads=4)
    ret, st = ret.cpu().numpy(), st.cpu().numpy()
    assert np.array_equal(ret, reto), (ret, reto)
    assert ret[0] == 0 and ret[5] == 1  # zero information: zero gradient -> CONVERGENCE at the start; NaN residual -> failure
    assert np.array_equal(st[ret == 1], start.numpy()[ret == 1])
    ok = ret == 0
    assert np.isfinite(st[ok]).all()
    dq, dt = pose_err(st[ok], so[ok])
    assert dq.max() <= 1e-4 and dt.max() <= 1e-4, (dq, dt)


def test_pnp_full_size_permutation_and_batch_independence():
    """B = 4096 x N = 64: (a) the solve does not depend on the order of the correspondences beyond round-off (sums over points);
    (b) jobs are independent: a slice of the batch alone returns the same rows bit for bit; (c) the in-place form (`start` aliasing
    `states`, the reference's contract) equals the out-of-place form."""
    from lc_amd import _lib
    from lc_amd.pnp import pnp_ceres

    dev = torch.device("cuda:0")
    B, N = 4096, 64
    b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=77).items()}
    st, tr, ret = pnp_ceres.solve_device(b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"])
    assert int(ret.sum()) == 0
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(2)).to(dev)
    stp, trp, retp = pnp_ceres.solve_device(b["K"], b["pts3d"][:, perm].contiguous(), b["pts2d"][:, perm].contiguous(),
                                            b["inv_std"][:, perm].contiguous(), b["start"])
    assert torch.equal(retp, ret)
    dq, dt = pose_err(stp.cpu().numpy(), st.cpu().numpy())
    assert dq.max() <= 1e-5 and dt.max() <= 1e-5, (dq.max(), dt.max())
    sl = slice(2000, 2049)
    sts, trs, rets = pnp_ceres.solve_device(*(b[k][sl].contiguous() for k in ("K", "pts3d", "pts2d", "inv_std", "start")))
    assert torch.equal(sts, st[sl]) and torch.equal(trs, tr[sl]) and torch.equal(rets, ret[sl])
    # in place: states holds the start poses on entry (lc_pnp_lm3_f32 with start == NULL)
    lib = _lib.load()
    inplace = b["start"].clone()
    tr2 = torch.empty(B, device=dev)
    ret2 = torch.empty(B, device=dev, dtype=torch.int32)
    rc = lib.lc_pnp_lm3_f32(_lib.ptr(b["K"]), _lib.ptr(b["pts3d"]), _lib.ptr(b["pts2d"]), None, _lib.ptr(b["inv_std"]), None, None, None, _lib.ptr(inplace),
                            _lib.ptr(tr2), _lib.ptr(ret2), None, B, N, 50, 1e-6, 0, 0, None, 0, _lib.stream_ptr(dev))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(inplace, st) and torch.equal(tr2, tr) and torch.equal(ret2, ret)


@pytest.mark.parametrize("B,N", [(37, 48), (9, 300), (5, 1500)])
def test_fused_input_handling_equals_the_elementwise_route(B, N):
    """lc_pnp_lm3_f32's options: nan_to_num, the square root of a diagonal inverse covariance and inlier-mask weights are applied at the
    kernel's loads; results must be the bits of the route that does the same with torch element-wise ops in front of the plain solve
    (cer_solver.py:29-36), including the (filtered) start returned for invalid jobs."""
    from lc_amd import synth
    from lc_amd.pnp import cer_solver, pnp_ceres

    dev = torch.device("cuda:0")
    d = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=B + N, outlier_frac=0.1).items()}
    g = torch.Generator().manual_seed(5)
    icov = d["inv_std"] ** 2
    pts2d, pts3d, start, K = d["pts2d"].clone(), d["pts3d"].clone(), d["start"].clone(), d["K"].clone()
    # poison: NaN / inf in every input kind
    pts2d[0, 1, 0] = float("nan"); pts3d[1, 2, 1] = float("inf"); icov[2, 0, 1] = float("nan"); icov[3, 1, 0] = float("inf")
    start[4, 5] = float("nan")
    counts = torch.randint(3, N + 1, (B,), generator=g).to(torch.int32).to(dev)
    counts[-1] = 2  # too few points: invalid, returns the start
    # (a) the reference-surface call, now on the fused route
    inv_a, st_a = cer_solver.solve(K, pts3d, pts2d, icov, start, counts, filter_input_nan=True)
    # (b) the element-wise route spelled out
    Kf, Xf, Uf, Wf, Sf = (torch.nan_to_num(t) for t in (K, pts3d, pts2d, icov, start))
    st_b, _tr, ret_b = pnp_ceres.solve_device(Kf, Xf, Uf, Wf.sqrt(), Sf, counts)
    st_b = torch.where((ret_b != 0)[:, None], Sf, st_b)
    assert torch.equal(inv_a["invalids"], ret_b != 0) and bool(inv_a["invalids"][-1])
    assert torch.equal(st_a, st_b)
    # without the filter the poisoned jobs fail and hand back their unfiltered start
    inv_c, st_c = cer_solver.solve(K, pts3d, pts2d, icov, start, counts)
    st_d, _tr, ret_d = pnp_ceres.solve_device(K, pts3d, pts2d, icov.sqrt(), start, counts)
    assert torch.equal(inv_c["invalids"], ret_d != 0)
    same = ~torch.isnan(st_d).any(-1)
    assert torch.equal(st_c[same], st_d[same]) and bool(inv_c["invalids"][0])
    # inlier-mask weights == float weights of ones and zeros
    m = (torch.rand(B, N, generator=g) > 0.3).to(dev)
    st_m, tr_m, ret_m = pnp_ceres.solve_device(d["K"], d["pts3d"], d["pts2d"], None, d["start"], counts, max_iter_count=20, weight_mask=m)
    w = m.float().unsqueeze(-1).expand(B, N, 2).contiguous()
    st_w, tr_w, ret_w = pnp_ceres.solve_device(d["K"], d["pts3d"], d["pts2d"], w, d["start"], counts, max_iter_count=20)
    assert torch.equal(st_m, st_w) and torch.equal(tr_m, tr_w) and torch.equal(ret_m, ret_w)


def test_shared_poses_batch_equals_separate_solves():
    """shared_poses: two correspondence selections of the same P objects solved as one launch of 2P poses (K and start given once)
    return the bits of the two separate solves."""
    from lc_amd import synth
    from lc_amd.pnp import pnp_ceres

    dev = torch.device("cuda:0")
    P, N = 23, 400
    d = {k: v.to(dev) for k, v in synth.make_batch(P, N, seed=3, outlier_frac=0.1).items()}
    g = torch.Generator().manual_seed(1)
    icov_a = d["inv_std"] ** 2
    icov_b = icov_a * (torch.rand(P, N, 1, generator=g).to(dev) > 0.4)
    cnt_a = torch.randint(50, N + 1, (P,), generator=g).to(torch.int32).to(dev)
    cnt_b = torch.randint(3, N + 1, (P,), generator=g).to(torch.int32).to(dev)
    kw = dict(weights_are_icov=True, nan_to_num=True)
    st_a, tr_a, ret_a = pnp_ceres.solve_device(d["K"], d["pts3d"], d["pts2d"], icov_a, d["start"], cnt_a, **kw)
    st_b, tr_b, ret_b = pnp_ceres.solve_device(d["K"], d["pts3d"], d["pts2d"], icov_b, d["start"], cnt_b, **kw)
    X2, U2 = torch.cat([d["pts3d"], d["pts3d"]]), torch.cat([d["pts2d"], d["pts2d"]])
    st, tr, ret = pnp_ceres.solve_device(d["K"], X2, U2, torch.cat([icov_a, icov_b]), d["start"], torch.cat([cnt_a, cnt_b]), shared_poses=P, **kw)
    assert torch.equal(st, torch.cat([st_a, st_b])) and torch.equal(tr, torch.cat([tr_a, tr_b])) and torch.equal(ret, torch.cat([ret_a, ret_b]))
    with pytest.raises(ValueError):
        pnp_ceres.solve_device(d["K"], X2[:P + 1], U2[:P + 1], torch.cat([icov_a, icov_b])[:P + 1], d["start"], None, shared_poses=P, **kw)


def test_fused_nan_filter_with_full_information_factor():
    """LC_PNP_NAN_TO_NUM with a full 2x2 lower factor (B,N,2,2): same bits as torch.nan_to_num on every input followed by the plain
    solve; a weight mask without point counts; the filter on the in-place form (states = start on entry)."""
    from lc_amd import _lib, synth
    from lc_amd.pnp import pnp_ceres

    dev = torch.device("cuda:0")
    B, N = 19, 96
    d = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=21, outlier_frac=0.1).items()}
    g = torch.Generator().manual_seed(2)
    L = torch.zeros(B, N, 2, 2)
    L[..., 0, 0] = torch.rand(B, N, generator=g) + 0.5
    L[..., 1, 1] = torch.rand(B, N, generator=g) + 0.5
    L[..., 1, 0] = torch.randn(B, N, generator=g) * 0.2
    L = L.to(dev)
    pts2d, K = d["pts2d"].clone(), d["K"].clone()
    pts2d[3, 7, 1] = float("nan"); L[5, 2, 1, 0] = float("inf"); K[6, 0, 2] = float("nan")
    a = pnp_ceres.solve_device(K, d["pts3d"], pts2d, L, d["start"], nan_to_num=True)
    Kf, Uf, Lf = torch.nan_to_num(K), torch.nan_to_num(pts2d), torch.nan_to_num(L)
    b = pnp_ceres.solve_device(Kf, d["pts3d"], Uf, Lf, d["start"])
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # mask weights, every row full
    m = (torch.rand(B, N, generator=g) > 0.2).to(dev)
    am = pnp_ceres.solve_device(d["K"], d["pts3d"], d["pts2d"], None, d["start"], weight_mask=m)
    bm = pnp_ceres.solve_device(d["K"], d["pts3d"], d["pts2d"], m.float().unsqueeze(-1).expand(B, N, 2).contiguous(), d["start"])
    for x, y in zip(am, bm):
        assert torch.equal(x, y)
    # in-place form through the C ABI: an invalid job must end with its FILTERED start in `states`
    lib = _lib.load()
    st = d["start"].clone()
    st[0, 4] = float("nan")
    counts = torch.full((B,), N, dtype=torch.int32, device=dev)
    counts[0] = 2  # too few points: invalid
    tr = torch.empty(B, device=dev); ret = torch.empty(B, device=dev, dtype=torch.int32)
    P = _lib.ptr
    rc = lib.lc_pnp_lm3_f32(P(d["K"]), P(d["pts3d"]), P(d["pts2d"]), None, P(d["inv_std"]), None, P(counts), None, P(st), P(tr), P(ret), None,
                            B, N, 50, 1e-6, pnp_ceres.LC_PNP_NAN_TO_NUM, 0, None, 0, _lib.stream_ptr(dev))
    assert rc == 0
    torch.cuda.synchronize()
    assert int(ret[0]) == 1 and float(st[0, 4]) == 0.0 and bool(torch.isfinite(st).all())


@pytest.mark.parametrize("hard", [False, True], ids=["metric-like", "hard starts + ragged + outliers"])
def test_large_grid_build_equals_the_latency_build(hard):
    """Batches of more than 1024 poses run the two-waves-per-SIMD build of the solve (default instruction schedule; with
    -DLC_BIG_LOWREG=1 the low-register form of lc_pnp_body.h), smaller ones the latency build (its own translation unit, max-ILP
    schedule, one wave per SIMD).  Same arithmetic, expression for expression: a pose solved inside a 3000-pose batch returns the
    same state, trust radius, flag and iteration count, bit for bit, as the same pose solved in a batch of 500 -- through the plain
    entry point and the one with the input handling fused in."""
    from lc_amd.pnp import pnp_ceres

    dev = torch.device("cuda:0")
    B, N = 3000, 64
    b = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=77, outlier_frac=0.3 if hard else 0.05, noise_px=2.0 if hard else 1.0).items()}
    counts = None
    if hard:
        g = torch.Generator().manual_seed(5)
        counts = torch.randint(0, N + 1, (B,), generator=g).to(torch.int32).to(dev)
        b["start"] = b["start"] + 0.2 * torch.randn(B, 7, generator=g).to(dev)  # far starts: many iterations, rejected steps, failures
    big = pnp_ceres.solve_device(b["K"], b["pts3d"], b["pts2d"], b["inv_std"], b["start"], counts, return_iters=True)
    for lo in (0, 500, 2500):
        sl = slice(lo, lo + 500)
        small = pnp_ceres.solve_device(b["K"][sl], b["pts3d"][sl], b["pts2d"][sl], b["inv_std"][sl], b["start"][sl],
                                       None if counts is None else counts[sl], return_iters=True)
        for a, c in zip(big, small):
            assert torch.equal(a[sl], c)
    assert int(big[3].max()) > 3 and (not hard or int(big[2].sum()) > 0)
    big2 = pnp_ceres.solve_device(b["K"], b["pts3d"], b["pts2d"], b["inv_std"] ** 2, b["start"], counts, weights_are_icov=True, nan_to_num=True)
    small2 = pnp_ceres.solve_device(b["K"][:500], b["pts3d"][:500], b["pts2d"][:500], b["inv_std"][:500] ** 2, b["start"][:500],
                                    None if counts is None else counts[:500], weights_are_icov=True, nan_to_num=True)
    for a, c in zip(big2, small2):
        assert torch.equal(a[:500], c)


@pytest.mark.parametrize("B1,k,N,used", [(16, 2, 700, 400), (24, 1, 1024, 1024), (8, 3, 300, 260), (16, 2, 64, 64), (16, 2, 1500, 900)])
def test_chained_solves_equal_the_two_calls(B1, k, N, used):
    """lc_pnp_lm_chain2_f32 (refinement on a mask, then k weighted solves per object that start from its result -- one launch where
    256 < N <= 1024) against lc_pnp_lm3_f32 twice: states, radii and flags of BOTH jobs bit for bit; incl. an object the first job skips
    (zero point count: the second starts from the first's start) and shapes that fall back to two launches."""
    from lc_amd import synth
    from lc_amd.pnp import pnp_ceres

    dev = torch.device("cuda:0")
    B2 = B1 * k
    a = {key: v.to(dev) for key, v in synth.make_batch(B1, N, seed=N + B1, outlier_frac=0.1, noise_px=1.0).items()}
    g = torch.Generator().manual_seed(B2)
    mask = (torch.rand(B1, N, generator=g) > 0.3).to(torch.uint8).to(dev)
    rows = torch.full((B1,), used, dtype=torch.int32, device=dev)
    rows[1] = 0  # a pose RANSAC gave up on
    first = dict(cam_mat=a["K"], pts3d=a["pts3d"], pts2d=a["pts2d"], sqrtL=None, start=a["start"], n_points=rows, max_iter_count=20,
                 weight_mask=mask)
    X2, U2 = a["pts3d"].repeat(k, 1, 1), a["pts2d"].repeat(k, 1, 1)
    W2 = (torch.rand(B2, N, 2, generator=g) * 4 + 0.05).to(dev)
    C2 = torch.randint(max(4, used // 2), used + 1, (B2,), generator=g).to(torch.int32).to(dev)
    second = dict(cam_mat=a["K"], pts3d=X2, pts2d=U2, sqrtL=W2, n_points=C2, weights_are_icov=True, nan_to_num=True,
                  shared_poses=B1 if k > 1 else 0)
    want1 = pnp_ceres.solve_device(**first)
    want2 = pnp_ceres.solve_device(**dict(second, start=want1[0]))
    got1, got2 = pnp_ceres.solve_chain_device(first, dict(second, start="first"))
    for x, y in zip(got1 + got2, want1 + want2):
        assert torch.equal(x, y)
    assert int(want2[2].sum()) < B2  # most second-stage solves converge
    # a second job that does not depend on the first
    got1, got2 = pnp_ceres.solve_chain_device(first, dict(second, start=a["start"]))
    want2 = pnp_ceres.solve_device(**dict(second, start=a["start"]))
    for x, y in zip(got1 + got2, want1 + want2):
        assert torch.equal(x, y)
