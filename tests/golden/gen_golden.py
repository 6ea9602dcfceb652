#!/usr/bin/env python3
"""Generate golden vectors by running the UNMODIFIED reference (read-only at /root/reference).

Run in the build container only (`python tests/golden/gen_golden.py`); the GPU box never sees
/root/reference.  Output: small .npz files next to this script holding inputs and the reference's
outputs (data only -- no reference source or bytecode is stored).

  lc_loss_<case>.npz   inputs (fp64 masters) + for each dtype in {f32,f64}: loss, grads w.r.t.
                       pts2d / inv_std / pts3d for the stored cotangent, and intermediates
                       w, c (robust_weights_cov), Hinv, A (weighted_pnp_jac_wrt_pts2d), G (jac_update2alter)
  head_<case>.npz      logits -> pts2d, std (ptnet.softargmax_2d_std on the spatial softmax) + input grad
  lossfn_<case>.npz    Loss_fn.forward trajectories (sparse and dense branch)
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("LC_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

from lc_amd import synth  # noqa: E402


def _stub_modules():
    """ptnet imports model.cdpn_resnet -> torchvision/mmcv, absent here; the functions we call are pure torch."""
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvr = types.ModuleType("torchvision.models.resnet")
    tvr.BasicBlock = type("BasicBlock", (torch.nn.Module,), {"expansion": 1})
    tvr.Bottleneck = type("Bottleneck", (torch.nn.Module,), {"expansion": 4})
    tvr.ResNet = type("ResNet", (torch.nn.Module,), {})
    tvr.model_urls = {}
    tvm.resnet = tvr
    tv.models = tvm
    mm = types.ModuleType("mmcv")
    mmc = types.ModuleType("mmcv.cnn")
    mmc.normal_init = lambda *a, **k: None
    mmc.constant_init = lambda *a, **k: None
    mm.cnn = mmc
    for name, mod in (("torchvision", tv), ("torchvision.models", tvm), ("torchvision.models.resnet", tvr),
                      ("mmcv", mm), ("mmcv.cnn", mmc)):
        sys.modules.setdefault(name, mod)


# --------------------------------------------------------------------------------------------
# LC loss cases
# --------------------------------------------------------------------------------------------
def loss_cases():
    cases = {}
    b = synth.make_batch(5, 16, seed=1, dtype=torch.float64)
    cases["base_B5_N16"] = dict(b, valid=None)

    b = synth.make_batch(4, 16, seed=2, dtype=torch.float64)
    cases["valid_ones_B4_N16"] = dict(b, valid=torch.ones(4, 16, dtype=torch.float64))

    b = synth.make_batch(4, 16, seed=3, dtype=torch.float64)
    g = torch.Generator().manual_seed(33)
    v = (torch.rand(4, 16, generator=g) > 0.3).to(torch.float64)
    cases["valid_mask_B4_N16"] = dict(b, valid=v)

    b = synth.make_batch(1, 4, seed=4, dtype=torch.float64)
    cases["tiny_B1_N4"] = dict(b, valid=None)

    b = synth.make_batch(256, 64, seed=0, dtype=torch.float64)
    cases["metric_B256_N64"] = dict(b, valid=None)

    b = synth.make_batch(2, 1024, seed=5, dtype=torch.float64, rotate_K=False)
    cases["dense_B2_N1024"] = dict(b, valid=torch.ones(2, 1024, dtype=torch.float64), want_pts3d=True)

    # zlmo's shape (128x128 maps, dense_sample 3 -> 43 x 43 correspondences; 29 tiles of 64, the last one ragged) with a ragged mask
    b = synth.make_batch(3, 1849, seed=15, dtype=torch.float64, outlier_frac=0.1)
    g = torch.Generator().manual_seed(35)
    cases["dense_B3_N1849_mask"] = dict(b, valid=(torch.rand(3, 1849, generator=g) > 0.25).to(torch.float64), want_pts3d=True)

    # clamp branch: explicit >32 px outliers; and a different max_err_len
    b = synth.make_batch(3, 32, seed=6, dtype=torch.float64)
    b["pts2d"][:, ::5] += torch.tensor([90.0, -70.0], dtype=torch.float64)
    cases["outliers_B3_N32"] = dict(b, valid=None, want_pts3d=True)
    b = synth.make_batch(3, 32, seed=7, dtype=torch.float64)
    cases["maxerr8_B3_N32"] = dict(b, valid=None, kwargs=dict(max_err_len=8))

    # Huber knees of the weights: wide inv_std range
    b = synth.make_batch(4, 24, seed=8, dtype=torch.float64)
    g = torch.Generator().manual_seed(88)
    b["inv_std"] = torch.exp(torch.randn(4, 24, 2, generator=g, dtype=torch.float64) * 1.2)
    cases["knee_B4_N24"] = dict(b, valid=None, want_pts3d=True)

    # SPD fallback: all-zero weights in sample 1 -> H=0 -> cholesky_ex info!=0 -> H:=I
    b = synth.make_batch(3, 16, seed=9, dtype=torch.float64)
    b["inv_std"][1] = 0
    cases["zero_weights_B3_N16"] = dict(b, valid=None)

    # camera z < 0.1 in sample 0: project_apply clamps, residual_with_jac6d does not
    b = synth.make_batch(2, 16, seed=10, dtype=torch.float64)
    b["pose"][0, 4:7] = torch.tensor([1.0, -2.0, 0.05], dtype=torch.float64)
    b["pts3d"][0] *= 0.0005
    cases["zclamp_B2_N16"] = dict(b, valid=None, want_pts3d=True)

    # non-unit quaternion: exercises two_s = 2/||q|| (rotation_conversions.py:52)
    b = synth.make_batch(3, 16, seed=11, dtype=torch.float64)
    b["pose"][:, :4] *= torch.tensor([[1.3], [0.8], [1.0]], dtype=torch.float64)
    cases["nonunit_quat_B3_N16"] = dict(b, valid=None, want_pts3d=True)

    # general last row of K: project_apply uses the full 3x3 matrix, residual_with_jac6d only K[:2] -> residual != 0
    b = synth.make_batch(3, 16, seed=14, dtype=torch.float64)
    b["K"][:, 2] = torch.tensor([1e-4, -2e-4, 1.05], dtype=torch.float64)
    cases["fullK_B3_N16"] = dict(b, valid=None, want_pts3d=True)

    # minimal problem (N=3: six residual rows for six pose parameters).  N=2 is NOT a fixture: H is then rank-deficient and
    # the reference's own fp32 and fp64 paths disagree on whether cholesky_ex flags it (round-off decides: 16.98 vs 21.59)
    b = synth.make_batch(3, 3, seed=15, dtype=torch.float64)
    cases["n3_B3_N3"] = dict(b, valid=None)

    # covariance of the projected bbox corners (cov_2d=True; no call site uses it, cov_mixed.py:125-127)
    b = synth.make_batch(4, 16, seed=13, dtype=torch.float64)
    cases["cov2d_B4_N16"] = dict(b, valid=None, want_pts3d=True, kwargs=dict(cov_2d=True))

    # noise-free (err == 0): linear_err norm at 0, c == 0
    b = synth.make_batch(2, 16, seed=12, dtype=torch.float64, outlier_frac=0.0, noise_px=0.0)
    cases["noisefree_B2_N16"] = dict(b, valid=None)
    return cases


def run_reference_loss(case, dtype):
    from lib import cov_mixed
    from lib.nll import pnp_auto
    import lib.transforms as xforms

    cast = lambda x: None if x is None else x.to(dtype)
    K, pose, X, u, s, bbox = (cast(case[k]) for k in ("K", "pose", "pts3d", "pts2d", "inv_std", "bbox_3d"))
    valid = cast(case.get("valid"))
    kwargs = dict(case.get("kwargs", {}))
    want3 = case.get("want_pts3d", False)
    u = u.clone().requires_grad_(True)
    s = s.clone().requires_grad_(True)
    X = X.clone().requires_grad_(want3)
    loss = cov_mixed.Loss_cov_mixed(K, pose, X, u, s, valid, bbox_3d=bbox, **kwargs)
    go = cast(case["grad_out"])
    grads = torch.autograd.grad(loss, [u, s] + ([X] if want3 else []), go, allow_unused=True)
    out = dict(loss=loss.detach(), g_pts2d=grads[0], g_inv_std=grads[1])
    if want3:
        out["g_pts3d"] = grads[2]
    # intermediates, by calling the reference's own sub-functions the way Loss_cov_mixed does
    with torch.enable_grad():
        R, t = xforms.quaternion_rep_to_RT(pose)
        proj = xforms.project_apply(K, X.detach(), R, t)
        err = u.detach() - proj
        ec = cov_mixed.clamp_error(err, kwargs.get("max_err_len", 32))
        w, c = cov_mixed.robust_weights_cov(s.detach(), ec, valid)
        A, Hinv = pnp_auto.weighted_pnp_jac_wrt_pts2d(proj, pose, K, X.detach(), w, with_cov=True)
        if kwargs.get("cov_2d", False):
            G = cov_mixed.jac_update2alter(pose, lambda st: cov_mixed.xform_2d(st, K, bbox))
        else:
            G = cov_mixed.jac_update2alter(pose, lambda st: cov_mixed.xform_3d(st, bbox))
    out.update(w=w.detach(), c=c.detach(), Hinv=Hinv.detach(), A=A.detach().flatten(-2), G=G.detach(), e=ec.detach())
    return {k: v.numpy() for k, v in out.items()}


def gen_loss(only=None):
    for name, case in loss_cases().items():
        if only and name not in only:
            continue
        B = case["K"].shape[0]
        g = torch.Generator().manual_seed(1234)
        case["grad_out"] = torch.rand(B, generator=g, dtype=torch.float64) + 0.5
        rec = {}
        for k in ("K", "pose", "pts3d", "pts2d", "inv_std", "bbox_3d", "start", "grad_out"):
            rec["in_" + k] = case[k].numpy()
        if case.get("valid") is not None:
            rec["in_valid"] = case["valid"].numpy()
        for k, v in case.get("kwargs", {}).items():
            rec["kw_" + k] = np.asarray(v)
        rec["want_pts3d"] = np.asarray(bool(case.get("want_pts3d", False)))
        for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
            out = run_reference_loss(case, dt)
            big = B * case["pts3d"].shape[1] > 4096
            for k, v in out.items():
                if big and k in ("A", "w", "c", "e", "G"):
                    continue  # keep the large fixtures small: loss + grads + Hinv only
                rec[f"{tag}_{k}"] = v
        path = os.path.join(HERE, f"lc_loss_{name}.npz")
        np.savez_compressed(path, **rec)
        print(f"{name:24s} loss[f64][:3]={rec['f64_loss'][:3]}  -> {os.path.getsize(path) / 1024:.0f} KiB")


# --------------------------------------------------------------------------------------------
# keypoint head
# --------------------------------------------------------------------------------------------
def gen_head():
    _stub_modules()
    import ptnet

    for name, (B, S, H, W, seed) in dict(b2_s4_64x64=(2, 4, 64, 64, 0), b2_s3_32x48=(2, 3, 32, 48, 2)).items():
        logits = synth.make_head_logits(B, S, H, W, seed=seed)
        g = torch.Generator().manual_seed(99 + seed)
        ct_mean = torch.randn(B, S, 2, generator=g)
        ct_std = torch.randn(B, S, 2, generator=g)
        rec = dict(in_logits=logits.numpy(), in_ct_mean=ct_mean.numpy(), in_ct_std=ct_std.numpy())
        for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
            lg = logits.to(dt).requires_grad_(True)
            prob = lg.flatten(start_dim=-2).softmax(dim=-1).reshape_as(lg)  # ptnet.py:61
            prob.retain_grad()
            mean, std = ptnet.softargmax_2d_std(prob)
            (gl,) = torch.autograd.grad([mean, std], [lg], [ct_mean.to(dt), ct_std.to(dt)], retain_graph=True)
            # also the prob-input form (function on its own, ptnet.py:100-115)
            pr = prob.detach().clone().requires_grad_(True)
            m2, s2 = ptnet.softargmax_2d_std(pr)
            (gp,) = torch.autograd.grad([m2, s2], [pr], [ct_mean.to(dt), ct_std.to(dt)])
            rec.update({f"{tag}_mean": mean.detach().numpy(), f"{tag}_std": std.detach().numpy(),
                        f"{tag}_g_logits": gl.numpy()})
            if tag == "f32":
                rec["f32_prob"] = prob.detach().numpy()
                rec["f32_g_prob"] = gp.numpy()
        path = os.path.join(HERE, f"head_{name}.npz")
        np.savez_compressed(path, **rec)
        print(f"head {name}: mean[0,0]={rec['f64_mean'][0, 0]} std[0,0]={rec['f64_std'][0, 0]} -> "
              f"{os.path.getsize(path) / 1024:.0f} KiB")


def head_compact_inputs(B, S, seed, logits_q=None):
    """Inputs of the compact head fixtures.  The logits are stored as int16 multiples of 1/1024 (half the bytes of fp32, and exact:
    `q / 1024` is the same float on every machine -- the synthetic generator itself is not, its exp() differs in the last bit
    between CPUs); the cotangents are stored in the fixture, the two probe tensors come from the integer generator."""
    if logits_q is None:
        logits_q = (synth.make_head_logits(B, S, 64, 64, seed=seed) * 1024).round().clamp(-32768, 32767).to(torch.int16)
    logits = logits_q.float() / 1024
    g = torch.Generator().manual_seed(199 + seed)
    ct_mean = torch.randn(B, S, 2, generator=g)
    ct_std = torch.randn(B, S, 2, generator=g)
    # two random functionals of the input gradient per map, from the INTEGER generator (exact and identical on every machine;
    # randn goes through log/sin/cos, whose last bit is not)
    probe = torch.randint(-(1 << 20), 1 << 20, (2, B, S, 64, 64), generator=torch.Generator().manual_seed(299 + seed)).double() / (1 << 20)
    return logits_q, logits, ct_mean, ct_std, probe


def gen_head_compact():
    """The sizes SURVEY.md 8c names -- (4,16,64,64) and (2,64,64,64) = the S = 64 map count of the metric -- without storing
    12 MB of maps: per-map outputs in full, the input gradient in full for 6 maps and through two seeded random linear
    functionals for EVERY map (reference-computed float64 scalars)."""
    _stub_modules()
    import ptnet

    for name, (B, S, seed) in dict(b4_s16_64x64=(4, 16, 11), b2_s64_64x64=(2, 64, 12)).items():
        logits_q, logits, ct_mean, ct_std, probe = head_compact_inputs(B, S, seed)
        rec = dict(in_shape=np.array([B, S, 64, 64]), in_seed=np.int32(seed), in_logits_q=logits_q.numpy(),
                   in_ct_mean=ct_mean.numpy(), in_ct_std=ct_std.numpy(), in_probe_xor=np.bitwise_xor.reduce(probe.numpy().view(np.uint64).ravel()))
        pick = [(0, 0), (0, S - 1), (B - 1, 0), (B - 1, S - 1), (B // 2, S // 2), (1, 3)]
        rec["g_maps"] = np.array(pick)
        for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
            lg = logits.to(dt).requires_grad_(True)
            prob = lg.flatten(start_dim=-2).softmax(dim=-1).reshape_as(lg)  # ptnet.py:61
            mean, std = ptnet.softargmax_2d_std(prob)
            (gl,) = torch.autograd.grad([mean, std], [lg], [ct_mean.to(dt), ct_std.to(dt)])
            rec.update({f"{tag}_mean": mean.detach().numpy(), f"{tag}_std": std.detach().numpy(),
                        f"{tag}_g_probe": (gl.double()[None] * probe).sum((-1, -2)).numpy(),
                        f"{tag}_g_absmax": gl.abs().amax((-1, -2)).numpy()})
            if tag == "f64":
                rec["f64_g_logits_maps"] = np.stack([gl[b, s].numpy() for b, s in pick]).astype(np.float32)
        path = os.path.join(HERE, f"headc_{name}.npz")
        np.savez_compressed(path, **rec)
        print(f"headc {name}: -> {os.path.getsize(path) / 1024:.0f} KiB")


def gen_pose_errors():
    """lib/utils/error6d.py (numpy/scipy only) on random pose pairs and a random vertex cloud."""
    from lib.utils import error6d
    from scipy.spatial.transform import Rotation

    rng = np.random.default_rng(0)
    B, M = 12, 700
    pts = (rng.random((M, 3)) * 2 - 1).astype(np.float32) * np.array(synth.EXTENT_MM, np.float32)
    R_gt = Rotation.random(B, random_state=1).as_matrix().astype(np.float32)
    dR = Rotation.from_rotvec(rng.normal(size=(B, 3)) * np.array([[0.0], [1e-4], [1e-3]] + [[0.05]] * (B - 3))).as_matrix()
    R_est = (R_gt.astype(np.float64) @ dR).astype(np.float32)
    t_gt = np.stack((rng.uniform(-50, 50, B), rng.uniform(-50, 50, B), rng.uniform(600, 1200, B)), -1).astype(np.float32)
    t_est = (t_gt + rng.normal(size=(B, 3)) * np.array([[0.0], [1e-3], [0.1]] + [[3.0]] * (B - 3))).astype(np.float32)
    out = {k: np.zeros(B) for k in ("adi", "add", "re", "te")}
    for i in range(B):
        args = [a.astype(np.float64) for a in (R_est[i], t_est[i].reshape(3, 1), R_gt[i], t_gt[i].reshape(3, 1))]
        out["adi"][i] = error6d.adi(*args, pts.astype(np.float64))
        out["add"][i] = error6d.add(*args, pts.astype(np.float64))
        out["re"][i] = error6d.re(args[0], args[2])
        out["te"][i] = error6d.te(args[1], args[3])
    path = os.path.join(HERE, "pose_err_b12_m700.npz")
    np.savez_compressed(path, in_R_est=R_est, in_t_est=t_est, in_R_gt=R_gt, in_t_gt=t_gt, in_pts=pts, **{"ref_" + k: v for k, v in out.items()})
    print("pose errors:", {k: v[:4] for k, v in out.items()}, os.path.getsize(path) // 1024, "KiB")


def gen_bits():
    """floatbits.py (pure torch): target generation, training decode (+ logit gradient) and inference decode."""
    import floatbits as fb

    g = torch.Generator().manual_seed(0)
    for name, (B, H, W, bits) in dict(b2_8x10_n655=(2, 8, 10, [6, 5, 5]), b2_16x16_n7=(2, 16, 16, 7)).items():
        noc = torch.rand(B, H, W, 3, generator=g) * 2 - 1
        mod_bits, raw_bits = fb.nn_noc2target(noc, bits)
        # network logits: mostly agreeing with the target code, with errors at random bit positions
        C = mod_bits.shape[1]
        logits = (mod_bits.float() * 2 - 1) * (torch.rand(B, C, H, W, generator=g) * 4 + 0.2)
        flip = torch.rand(B, C, H, W, generator=g) < 0.15
        logits = torch.where(flip, -logits, logits)
        msk = torch.rand(B, H, W, generator=g) > 0.25
        ct = torch.randn(B, H, W, 3, generator=g)
        rec = dict(in_noc=noc.numpy(), in_logits=logits.numpy(), in_raw_bits=raw_bits.numpy(), in_mod_bits=mod_bits.numpy(),
                   in_msk=msk.numpy(), in_ct=ct.numpy(), bits=np.asarray(bits if isinstance(bits, list) else [bits] * 3))
        for tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
            lg = logits.to(dt).requires_grad_(True)
            out = fb.nn_logits2noc_with_gt(lg, raw_bits, bits, msk)
            (gl,) = torch.autograd.grad(out, lg, ct.to(dt))
            inf = fb.nn_logits2noc(logits.to(dt), bits)
            rec.update({f"{tag}_noc_gt": out.detach().numpy(), f"{tag}_g_logits": gl.numpy(), f"{tag}_noc_inf": inf.numpy()})
        path = os.path.join(HERE, f"bits_{name}.npz")
        np.savez_compressed(path, **rec)
        print("bits", name, rec["f64_noc_gt"][0, 0, 0], rec["f64_noc_inf"][0, 0, 0], os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    what = sys.argv[1:] or ["loss", "head"]
    if "bits" in what:
        gen_bits()
    if "errors" in what:
        gen_pose_errors()
    if "loss" in what:
        gen_loss([w[5:] for w in what if w.startswith("only=")])  # e.g. `gen_golden.py loss only=dense_B3_N1849_mask`
    if "head" in what:
        gen_head()
    if "headc" in what:
        gen_head_compact()
    if "lossfn" in what:
        from gen_golden_lossfn import gen_lossfn  # noqa

        gen_lossfn()
