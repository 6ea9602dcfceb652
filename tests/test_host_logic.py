"""CPU tests of the host layer: the C-ABI library's exports, the reference-surface semantics of cer_solver / Loss_fn /
NormClipper (against golden trajectories from the reference), and the drop-in installer.  Compute runs on the oracle
through tests/cpu_backend.py -- no HIP kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from tests.cpu_backend import oracle_backend  # noqa: F401
from tests.util import GOLDEN, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    from lc_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "lc_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lc_[a-z0-9_]+|pnp_ceres_f32_omp)\s*\(", hdr))
    assert {"pnp_ceres_f32_omp", "lc_pnp_lm3_f32", "lc_cov_loss3_fwd_bwd_f32", "lc_pose_unit2_f32", "lc_softargmax2d_fwd", "lc_xyz_bin_loss_counts"} <= declared
    # one entry point per operation (LC_AMD_VERSION 2): no superseded generation is declared or exported any more
    retired = {"lc_pnp_lm_f32", "lc_pnp_lm2_f32", "lc_pnp_lm_chain_f32", "lc_cov_loss_fwd_bwd_f32", "lc_cov_loss2_fwd_bwd_f32", "lc_pose_unit_f32",
               "lc_softargmax2d_fwd_f32", "lc_dense_frontend_fwd_f32", "lc_dense_frontend_fwd2_f32", "lc_dense_frontend_select_f32",
               "lc_dense_frontend_select2", "lc_pnp_ransac_init_f32", "lc_pnp_ransac_init2_f32", "lc_pnp_ransac_init3_f32", "lc_pnp_ransac_init4_f32",
               "lc_bits_decode_f32", "lc_bits_decode2_f32", "lc_bits_decode_gt_fwd_f32", "lc_bits_decode_gt_fwd2_f32", "lc_sqnorm_f32",
               "lc_norm_clip_apply_f32", "lc_xyz_bin_loss_fwd_f32", "lc_dense_aux_fwd_f32"}
    assert not (retired & declared)
    lib = ctypes.CDLL(_lib.lib_path())
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/lc_amd.h but not exported"
    for name in retired:
        assert not hasattr(lib, name), f"{name} was retired but is still exported"
    assert set(_lib.EXPORTED_SYMBOLS) == declared
    assert _lib.load().lc_amd_version() == 2


def test_ransac_workspace_contract():
    """lc_pnp_ransac_workspace_bytes: B arrival counters (padded to 16 bytes), (B, H, 12) doubles + floats for the hypotheses and
    (B, chunks of 64 points, H) count/error partials, H = iterations rounded up to 64; a smaller workspace is refused before anything
    is launched (no GPU needed), by both split-form entry points."""
    from lc_amd import _lib

    lib = _lib.load()
    assert lib.lc_pnp_ransac_workspace_bytes(0, 10, 150) == 0 and lib.lc_pnp_ransac_workspace_bytes(4, 10, 0) == 0
    B, N, H = 5, 1000, 192
    chunks = (N + 63) // 64
    ctr = 16 * ((B + 3) // 4)  # the doubles behind the counters are read as 16-byte pairs
    assert lib.lc_pnp_ransac_workspace_bytes(B, N, 150) == ctr + B * H * 12 * (8 + 4) + B * chunks * H * 8
    assert lib.lc_pnp_ransac_workspace_bytes(B, 16384, 150) == ctr + B * H * 12 * 12 + B * 256 * H * 8  # every point of a row is scored
    buf = ctypes.create_string_buffer(64)
    p = ctypes.addressof(buf)
    rc = lib.lc_pnp_ransac_init5_f32(p, p, p, None, B, N, 2.0, None, 150, 0, p, p, p, p, None, None, p, 64, 0, None, None, 4, 0, None, None,
                                     None, None, None, 0, None)
    assert rc != 0 and b"workspace" in lib.lc_amd_last_error()
    rc = lib.lc_pnp_ransac_init5_f32(p, p, p, None, B, N, 2.0, None, 150, 0, p, p, p, p, None, None, None, 0, 0, p, None, 4, 0, p, p, p, None,
                                     None, 0, None)  # selection asked for, no counts output
    assert rc != 0 and b"selection" in lib.lc_amd_last_error()
    # the section offsets the diagnostics read the workspace through come from the library (lc_pnp_init.hip: carve_workspace)
    lay = (ctypes.c_size_t * 6)()
    assert lib.lc_pnp_ransac_workspace_layout(B, N, 150, lay) == 0
    assert list(lay) == [ctr, ctr + B * H * 12 * 8, ctr + B * H * 12 * 12, H, chunks, lib.lc_pnp_ransac_workspace_bytes(B, N, 150)]
    assert lib.lc_pnp_ransac_workspace_layout(0, N, 150, lay) != 0


def test_extended_entry_points_reject_bad_arguments_before_launching():
    """lc_pnp_lm3_f32 / lc_dense_frontend_fwd3: argument errors are reported by return code + lc_amd_last_error, nothing is
    launched (runs without a GPU)."""
    from lc_amd import _lib

    lib = _lib.load()
    buf = ctypes.create_string_buffer(256)
    p = ctypes.addressof(buf)
    err = lambda: lib.lc_amd_last_error().decode()
    # two weight forms at once / none at all
    assert lib.lc_pnp_lm3_f32(p, p, p, p, p, None, None, p, p + 64, p, p, None, 2, 4, 5, 1e-6, 0, 0, None, 0, None) != 0 and "exactly one" in err()
    assert lib.lc_pnp_lm3_f32(p, p, p, None, None, None, None, p, p + 64, p, p, None, 2, 4, 5, 1e-6, 0, 0, None, 0, None) != 0 and "exactly one" in err()
    # unknown option bit, icov flag without the diagonal weights, shared poses without a separate start
    assert lib.lc_pnp_lm3_f32(p, p, p, None, p, None, None, p, p + 64, p, p, None, 2, 4, 5, 1e-6, 8, 0, None, 0, None) != 0 and "option" in err()
    assert lib.lc_pnp_lm3_f32(p, p, p, None, None, p, None, p, p + 64, p, p, None, 2, 4, 5, 1e-6, 1, 0, None, 0, None) != 0 and "weights_diag" in err()
    assert lib.lc_pnp_lm3_f32(p, p, p, None, p, None, None, None, p + 64, p, p, None, 2, 4, 5, 1e-6, 0, 1, None, 0, None) != 0 and "start" in err()
    # visibility logits without a mask buffer
    assert lib.lc_dense_frontend_fwd3(p, p, p, None, p, 0.5, 0, 0, 0, 0, 0, 0, 1, 4, 4, 0, 0, 2, p, p, p, p, None, None) != 0 and "vis_" in err()
    # empty batches are fine
    assert lib.lc_pnp_lm3_f32(p, p, p, None, p, None, None, p, p + 64, p, p, None, 0, 4, 5, 1e-6, 0, 0, None, 0, None) == 0


def test_product_path_has_no_cpu_fallback():
    from lc_amd import synth
    from lc_amd.cov_mixed import Loss_cov_mixed
    from lc_amd.pnp import cer_solver
    from lc_amd.ptnet import softargmax_2d_std

    b = synth.make_batch(2, 8, seed=0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Loss_cov_mixed(b["K"], b["pose"], b["pts3d"], b["pts2d"], b["inv_std"], None, bbox_3d=b["bbox_3d"])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        softargmax_2d_std(torch.rand(2, 3, 8, 8))
    with pytest.raises(TypeError):
        from lc_amd import _lib
        _lib.require_hip_f32("x", [1, 2, 3])


def test_ragged_jobs_become_one_padded_batch():
    from lc_amd.pnp.cer_solver import _batch_tensors

    a = [torch.ones(3, 2), torch.ones(5, 2) * 2, torch.ones(1, 2) * 3]
    (bt, cnt, none, same) = _batch_tensors(a, [3, 5, 1], None, [torch.zeros(7), torch.ones(7), torch.zeros(7)])
    assert bt.shape == (3, 5, 2) and none is None and cnt.tolist() == [3, 5, 1] and same.shape == (3, 7)
    assert bt[0, 3:].abs().sum() == 0 and bt[2, 1:].abs().sum() == 0 and (bt[1] == 2).all()
    already = torch.rand(3, 4, 2)
    assert _batch_tensors(already)[0] is already or torch.equal(_batch_tensors(already)[0], already)


def test_cer_solver_surface_on_oracle_backend(oracle_backend):
    """Ragged lists + NaN filtering + invalid -> start + optimal_start, with the LM solve supplied by the oracle."""
    from lc_amd import synth
    from lc_amd.pnp import cer_solver

    b = synth.make_batch(5, 24, seed=3)
    n = [24, 10, 2, 24, 7]
    pts3d = [b["pts3d"][i, :n[i]] for i in range(5)]
    pts2d = [b["pts2d"][i, :n[i]].clone() for i in range(5)]
    pts2d[1][0, 1] = float("nan")
    icov = [b["inv_std"][i, :n[i]] ** 2 for i in range(5)]
    inv, st = cer_solver.solve(b["K"], pts3d, pts2d, icov, list(b["start"]), num_workers=4, filter_input_nan=True)
    assert st.shape == (5, 7) and inv["invalids"].tolist()[2] is True
    assert torch.equal(st[2], b["start"][2])
    ok = ~inv["invalids"]
    assert (st[ok] - b["start"][ok]).abs().max() > 1e-3  # moved away from the start
    inv2, st2 = cer_solver.solve(b["K"], b["pts3d"], b["pts2d"], b["inv_std"] ** 2, b["start"], optimal_start=True)
    assert torch.equal(st2, b["start"]) and not inv2["invalids"].any()


@pytest.mark.parametrize("kind", ["sparse", "dense", "bin"])
def test_loss_fn_matches_reference_trajectory(oracle_backend, kind):
    """lc_amd.losses.Loss_fn (host logic) + oracle loss == the reference's Loss_fn over the warm-up ramp, incl. the
    NormClipper.max_norm buffer trajectory and the gradients on the network outputs."""
    from lc_amd.losses import Loss_fn
    from tests.golden.gen_golden_lossfn import run

    z = np.load(os.path.join(GOLDEN, f"lossfn_{kind}_f64.npz"))
    rec = run(Loss_fn, kind, list(z["steps"]), torch.float64)
    assert set(rec) == set(z.files)
    for k in z.files:
        if k == "steps":
            continue
        # losses/state to 1e-8; gradients pass through the EMA-clipped hooks, where 1e-10 oracle-vs-reference
        # differences of tiny LC gradients are amplified relative to the tensor max -> 1e-6
        assert rel_err(rec[k], z[k]) <= (1e-6 if "_grad_" in k else 1e-8), k
    z32 = np.load(os.path.join(GOLDEN, f"lossfn_{kind}_f32.npz"))
    rec32 = run(Loss_fn, kind, list(z32["steps"]), torch.float32)
    for k in z32.files:
        if re.match(r"s\d+_w?loss_", k):
            assert abs(float(rec32[k]) - float(z32[k])) <= 1e-4 * max(1.0, abs(float(z32[k]))), k


@pytest.mark.parametrize("kind", ["dense_glmo", "bin_zlmo", "sparse_metric", "dense_plumb"])
def test_loss_fn_matches_reference_at_training_shapes(oracle_backend, kind):
    """The same host logic + oracle at the reference's own training shapes (configs/glmo.yaml: 64x64 maps, stride 2, N=1024;
    configs/zlmo.yaml: 128x128 maps, stride 3, N=1849, 21 code planes; gsplmo's loss block at B=256 N=64; BASELINE configs[0]'s plumbing
    case: 16 crops, 32x32 maps, stride 2, N=256) against the trajectories the
    unmodified reference class produced (tests/golden/gen_golden_lossfn.py --train-shapes): float64 losses / states to 1e-8, gradients
    (stored rounded to float32) to 1e-6 of their largest entry; the float32 run against the reference's float32 scalars to 1e-4."""
    from lc_amd.losses import Loss_fn
    from tests.golden.gen_golden_lossfn import TRAIN_KINDS, run

    z = np.load(os.path.join(GOLDEN, f"lossfn_{kind}.npz"))
    assert list(z["steps"]) == TRAIN_KINDS[kind][0]
    rec = run(Loss_fn, kind, list(z["steps"]), torch.float64)
    assert set(rec) == {k for k in z.files if not k.startswith("f32_")}
    for k in rec:
        if k != "steps":
            assert rel_err(rec[k], z[k]) <= (1e-6 if "_grad_" in k else 1e-8), k
    rec32 = run(Loss_fn, kind, list(z["steps"]), torch.float32)
    for k in z.files:
        if re.match(r"f32_s\d+_w?loss_", k):
            assert abs(float(rec32[k[4:]]) - float(z[k])) <= 1e-4 * max(1.0, abs(float(z[k]))), k
        elif k.startswith("f32_") and "_state_" in k:
            assert rel_err(rec32[k[4:]], z[k]) <= 1e-3, k


def test_loss_fn_state_dict_keys_match_reference_checkpoints():
    from lc_amd.config import AttrDict
    from lc_amd.losses import Loss_fn

    fn = Loss_fn(AttrDict(pose_loss_cfg=dict(type="cov", clip_weight_grad=True)), AttrDict())
    assert list(fn.state_dict().keys()) == ["weight_grad_clipper.max_norm"]
    fn2 = Loss_fn(AttrDict(pose_loss_cfg=dict(clip_weight_grad=True, clip_scale_grad=True, clip_pts_grad=True)), AttrDict())
    assert set(fn2.state_dict()) == {"weight_grad_clipper.max_norm", "scale_grad_clipper.max_norm", "pts_grad_clipper.max_norm"}
    fn3 = Loss_fn(AttrDict(pose_loss_cfg=dict()), AttrDict(), total_bit_cnt=16)
    assert set(fn3.state_dict()) == {"weight_grad_clipper.max_norm", "xyz_bin_loss_fn.histogram"}  # losses.py:199


def test_dense_matching_shapes_and_phase():
    from lc_amd.losses import dense_pnp_matching_from_xyz

    xyz, w = torch.randn(2, 3, 9, 10), torch.rand(2, 2, 9, 10)
    msk = torch.rand(2, 9, 10) > 0.5
    p2, s, p3, v = dense_pnp_matching_from_xyz(xyz, w, msk, torch.tensor([[1.0, 2, 3], [4, 5, 6]]), sample=3, top_left=(1, 2))
    assert p2.shape == (2, 9, 2) and s.shape == (2, 9, 2) and p3.shape == (2, 9, 3) and v.shape == (2, 9)
    assert p2[0, 0].tolist() == [2.0, 1.0]  # (x, y) of the first sampled pixel
    assert torch.allclose(p3[1, 0], xyz[1, :, 1, 2] * torch.tensor([4.0, 5, 6]))
    assert torch.equal(s[0, 4], w[0, :, 4, 5])


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference checkout only exists in the build container")
def test_dropin_install_rebinds_reference_entry_points():
    import subprocess
    import sys

    code = (
        "import sys, types, warnings; warnings.filterwarnings('ignore'); sys.path.insert(0, '/root/reference'); sys.path.insert(0, %r)\n"
        "from tests.golden.gen_golden import _stub_modules; _stub_modules()\n"
        "import lc_amd.dropin as d; r = d.install(); print(r)\n"
        "import lib.cov_mixed, losses, ptnet, lc_amd.cov_mixed as cm, lc_amd.ptnet as hp\n"
        "from lib.pnp import cer_solver, pnp_ceres\n"
        "assert lib.cov_mixed.Loss_cov_mixed is cm.Loss_cov_mixed and losses.Loss_cov_mixed is cm.Loss_cov_mixed\n"
        "assert cer_solver.__name__ == 'lc_amd.pnp.cer_solver' and pnp_ceres.__name__ == 'lc_amd.pnp.pnp_ceres'\n"
        "from lib.pnp import cv2_solver; assert cv2_solver.__name__ == 'lc_amd.pnp.gpu_solver'  # no OpenCV in this image\n"
        "assert ptnet.softargmax_2d_std is hp.softargmax_2d_std\n"
        "import lc_amd.grad as g, lc_amd.losses as ol, lib.utils.grad as rg\n"
        "assert rg.NormClipper is g.NormClipper and losses.NormClipper is g.NormClipper\n"
        "assert losses.Loss_fn.dense_pose_loss is ol.Loss_fn.dense_pose_loss and losses.Loss_fn.sparse_kpt_loss is ol.Loss_fn.sparse_kpt_loss\n"
        "from lc_amd.config import AttrDict\n"
        "fn = losses.Loss_fn(AttrDict(pose_loss_cfg=dict(type='cov', clip_weight_grad=True, clip_scale_grad=True), seg_loss_type='L1'), AttrDict())\n"
        "assert type(fn.weight_grad_clipper) is g.NormClipper and set(fn.state_dict()) == {'weight_grad_clipper.max_norm', 'scale_grad_clipper.max_norm'}\n"
        "assert all(r.values()), r\n"
        # the reference's OWN Loss_fn, running on the swapped entry points (oracle backend on the CPU), reproduces its golden trajectory
        "import numpy as np, torch\n"
        "from tests import cpu_backend; cpu_backend.apply(setattr)\n"
        "from tests.golden.gen_golden_lossfn import run\n"
        "for kind, steps in (('sparse', [0, 2, 4, 6, 9]), ('dense', [0, 1, 2, 5])):\n"
        "    rec = run(losses.Loss_fn, kind, steps, torch.float64)\n"
        "    z = np.load(%r + '/tests/golden/lossfn_' + kind + '_f64.npz')\n"
        "    for k in z.files:\n"
        "        if k != 'steps':\n"
        "            a, b = np.asarray(rec[k], np.float64), np.asarray(z[k], np.float64)\n"
        "            assert np.abs(a - b).max() <= (1e-6 if '_grad_' in k else 1e-8) * max(1.0, np.abs(b).max()), (kind, k)\n" % (ROOT, ROOT))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]


def test_xyz_to_nn_target_inverts_nn_out_to_xyz_for_the_continuous_head():
    from lc_amd.losses import nn_out_to_xyz, xyz_to_nn_target

    g = torch.Generator().manual_seed(0)
    xyz = torch.randn(2, 5, 6, 3, generator=g) * 30
    scale = torch.tensor([[37.8, 37.9, 45.8], [10.0, 20.0, 30.0]])
    tgt, raw = xyz_to_nn_target(xyz, scale)
    assert raw is None and tgt.shape == (2, 3, 5, 6)
    assert torch.allclose(nn_out_to_xyz(tgt, scale), xyz, atol=1e-5)


def test_rt_to_quaternion_rep_round_trips_all_branches():
    """Every branch of the component selection (w, x, y or z largest, incl. 180-degree turns) round-trips through quaternion_rep_to_RT."""
    from lc_amd.transforms import RT_to_quaternion_rep, quaternion_rep_to_RT

    g = torch.Generator().manual_seed(0)
    q = torch.randn(200, 4, generator=g, dtype=torch.float64)
    q[:4] = torch.eye(4, dtype=torch.float64)          # identity and the three half turns
    q[4:8] = torch.eye(4, dtype=torch.float64) + 1e-9  # next to them
    q = q / q.norm(dim=-1, keepdim=True)
    t = torch.randn(200, 3, generator=g, dtype=torch.float64)
    R, _ = quaternion_rep_to_RT(torch.cat((q, t), -1))
    rep = RT_to_quaternion_rep(R, t)
    assert rep.shape == (200, 7) and torch.allclose(rep[:, :4].norm(dim=-1), torch.ones(200, dtype=torch.float64), atol=1e-12)
    sgn = torch.sign((rep[:, :4] * q).sum(-1, keepdim=True))
    assert (rep[:, :4] * sgn - q).abs().max() < 1e-7 and torch.equal(rep[:, 4:], t)
    k = rep[:, :4].abs().argmax(-1)
    assert (rep[torch.arange(200), k] > 0).all() and set(k.tolist()) == {0, 1, 2, 3}
    R2, _ = quaternion_rep_to_RT(rep)
    assert (R2 - R).abs().max() < 1e-12


def test_label_prep_names_pass_through_to_the_reference(monkeypatch):
    """`lc_amd.losses.annots_on_the_fly / selete_best_pose / xyz_from_homo_z` (label preparation, outside the hot path) delegate to
    the reference's own module when a checkout is reachable, and raise ImportError otherwise."""
    import os
    import sys

    import lc_amd.losses as L

    monkeypatch.delenv("LC_REFERENCE", raising=False)
    monkeypatch.delitem(sys.modules, "losses", raising=False)
    with pytest.raises(ImportError):
        L.xyz_from_homo_z(None, None, None, None)
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "losses.py")):
        pytest.skip("no reference checkout on this machine (GPU box)")
    monkeypatch.setenv("LC_REFERENCE", ref)
    monkeypatch.setattr(sys, "path", list(sys.path))
    g = torch.Generator().manual_seed(0)
    B, H, W = 2, 4, 5
    homo_z = torch.randn(B, H, W, 3, generator=g)
    R = torch.linalg.qr(torch.randn(B, 3, 3, generator=g))[0]
    t, K = torch.randn(B, 3, generator=g), torch.eye(3).expand(B, 3, 3) + 0.1 * torch.randn(B, 3, 3, generator=g)
    got = L.xyz_from_homo_z(homo_z, R, t, K)
    want = homo_z @ (torch.linalg.inv(K).unsqueeze(-3).mT @ R.unsqueeze(-3)) - (t[:, None, None, :] @ R.unsqueeze(-3))
    assert torch.allclose(got, want, atol=1e-5)
    gt = dict(Rt_candi=[torch.cat((R, t[..., None]), -1).unsqueeze(-3)], homo_z_out=homo_z, R_no_aug=R, t_no_aug=t, K_no_aug=K,
              msk_noc=torch.ones(B, H, W))
    Rt_best, pose_best, xyz_gt = L.selete_best_pose(gt, {}, False)
    assert Rt_best.shape == (B, 3, 4) and pose_best.shape == (B, 7) and torch.allclose(xyz_gt, want, atol=1e-5)


def test_chain_and_round3_entry_points_check_their_arguments_before_launching():
    """lc_pnp_lm_chain2_f32 (plain C struct jobs), lc_dense_frontend_select3, lc_dense_aux_fwd2, lc_xyz_bin_loss_fwd2 / _counts / _finish: argument
    errors come back as return code + lc_amd_last_error and empty batches are no-ops, all without touching a GPU."""
    from lc_amd import _lib
    from lc_amd.pnp.pnp_ceres import _Job

    lib = _lib.load()
    buf = ctypes.create_string_buffer(256)
    p = ctypes.addressof(buf)
    empty = _Job(None, None, None, None, None, None, None, None, None, None, None, None, 0, 700, 20, 1e-6, 0, 0)
    assert lib.lc_pnp_lm_chain2_f32(ctypes.byref(empty), ctypes.byref(empty), None, 0, None) == 0  # two empty batches
    assert lib.lc_pnp_lm_chain2_f32(None, ctypes.byref(empty), None, 0, None) != 0 and b"null job" in lib.lc_amd_last_error()
    bad = _Job(p, p, p, None, p, None, None, None, p, p, p, None, 4, 700, 20, 1e-6, 8, 0)  # unknown option bit
    assert lib.lc_pnp_lm_chain2_f32(ctypes.byref(empty), ctypes.byref(bad), None, 0, None) != 0 and b"option" in lib.lc_amd_last_error()
    two = _Job(p, p, p, p, p, None, None, None, p, p, p, None, 4, 700, 20, 1e-6, 0, 0)  # two weight forms at once
    assert lib.lc_pnp_lm_chain2_f32(ctypes.byref(two), ctypes.byref(empty), None, 0, None) != 0 and b"exactly one" in lib.lc_amd_last_error()
    # front end + selection: more sampled pixels than the one launch takes; a mask mode without visibility logits
    rc = lib.lc_dense_frontend_select3(p, p, p, None, p, 0.5, 0, 0, 0, 0, 0, 0, 1, 258, 256, 0, 0, 2, 0, 0.5, 1, 4, 0, 0, p, p, p, None, p, None, 0, None)
    assert rc != 0 and b"16384" in lib.lc_amd_last_error()
    rc = lib.lc_dense_frontend_select3(p, p, p, None, None, 0.5, 0, 0, 0, 0, 0, 0, 1, 64, 64, 0, 0, 2, 2, 0.5, 1, 4, 0, 0, p, p, p, None, p, None, 0, None)
    assert rc != 0 and b"visibility" in lib.lc_amd_last_error()
    assert lib.lc_dense_frontend_select3(p, p, p, None, p, 0.5, 0, 0, 0, 0, 0, 0, 0, 64, 64, 0, 0, 2, 0, 0.5, 1, 4, 0, 0, p, p, p, None, p, None, 0, None) == 0
    # dense auxiliary losses: xyz without exactly one mask form; unknown loss type; empty batch
    rc = lib.lc_dense_aux_fwd2(p, None, None, p, p, p, None, 0, 0, 0, 0, 2, 64, 0, p, p, p, None)
    assert rc != 0 and b"mask" in lib.lc_amd_last_error()
    assert lib.lc_dense_aux_fwd2(None, None, None, None, p, p, None, 0, 0, 0, 0, 2, 64, 7, p, p, p, None) != 0
    assert lib.lc_dense_aux_fwd2(None, None, None, None, p, p, None, 0, 0, 0, 0, 0, 64, 0, p, p, p, None) == 0
    # code loss: more bits than the kernel's per-bit tables hold
    rc = lib.lc_xyz_bin_loss_fwd2(p, p, p, 0, 0, 0, 2, 200, 64, 0.05, p, p, p, p, p, None)
    assert rc != 0 and b"128" in lib.lc_amd_last_error()
    assert lib.lc_xyz_bin_loss_fwd2(p, p, p, 0, 0, 0, 0, 17, 64, 0.05, p, p, p, p, p, None) == 0
    # its sharded form: counts without the place for the BCE means; a finish without counts; an empty batch
    assert lib.lc_xyz_bin_loss_counts(p, p, p, 0, 0, 0, 2, 17, 64, p, None, p, p, None) != 0 and b"null" in lib.lc_amd_last_error()
    assert lib.lc_xyz_bin_loss_counts(p, p, p, 0, 0, 0, 2, 17, 64, None, p, p, p, None) != 0
    assert lib.lc_xyz_bin_loss_counts(p, p, p, 0, 0, 0, 0, 17, 64, p, p, p, p, None) == 0
    assert lib.lc_xyz_bin_loss_finish(None, p, 17, 0.05, p, p, p, None) != 0 and b"null" in lib.lc_amd_last_error()
    assert lib.lc_xyz_bin_loss_finish(p, p, 200, 0.05, p, p, p, None) != 0 and b"128" in lib.lc_amd_last_error()


def test_training_shape_fixtures_span_the_warm_up_ramp_and_the_reference_shapes():
    """What VERDICT r4 #1 asked the new fixtures to be: the reference's own shapes and loss blocks, steps across the warm-up ramp (losses.py:272-276,
    296-302), at least three consecutive calls for the clipper's trajectory -- checked on the committed files and the seeded inputs."""
    from lc_amd import synth
    from lc_amd.config import AttrDict
    from lc_amd.losses import pose_loss_factor
    from tests.golden.gen_golden_lossfn import TRAIN_KINDS

    shapes = {"dense_glmo": ((4, 3, 64, 64), 2, 1024), "bin_zlmo": ((4, 21, 128, 128), 3, 1849), "dense_plumb": ((16, 3, 32, 32), 2, 256)}
    for kind, (steps, spe, bits) in TRAIN_KINDS.items():
        cfg = AttrDict(synth.TRAIN_LOSS_CONFIGS[kind])
        f = [pose_loss_factor(cfg, s, spe) for s in steps]
        assert len(steps) >= 3 and min(f) < 1e-3 and any(0.4 < v < 0.6 for v in f) and max(f) == 1.0, (kind, f)
        gt, out = synth.train_inputs(kind, seed=0)
        gt2, out2 = synth.train_inputs(kind, seed=0)
        assert all(torch.equal(out[k], out2[k]) for k in out)  # seeded
        z = np.load(os.path.join(GOLDEN, f"lossfn_{kind}.npz"))
        if kind in shapes:
            shape, stride, n = shapes[kind]
            head = "xyz_noc_bin" if bits else "xyz_noc"
            assert tuple(out[head].shape) == shape and z[f"s0_grad_{head}"].shape == shape and z[f"s0_grad_{head}"].dtype == np.float32
            H = shape[-1]
            assert (-(-H // stride)) ** 2 == n and cfg["pose_loss_cfg"].get("dense_sample", 2) == stride
            assert sum(gt.get("bit_cnt", [])) == bits
            assert 0.1 < float(gt["msk_vis"].mean()) < 0.5 and float(gt["msk_vis"].sum()) < float(gt["msk_noc"].sum())  # an object with an occluded part
            mn = [float(z[f"s{i}_state_weight_grad_clipper.max_norm"]) for i in range(len(steps))]
            assert len(set(mn)) == len(mn) and min(mn) > 0
        else:
            assert tuple(out["pts2d"].shape) == (256, 64, 2) and z["s0_grad_pts2d"].shape == (256, 64, 2)
