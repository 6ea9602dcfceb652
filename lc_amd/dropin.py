"""Drop-in installer: run the reference's own `train.py` / `test.py` on the HIP hot path without editing them.

    python -m lc_amd.dropin /path/to/lc/train.py --cfg configs/gsplmo.yaml ...

`install()` must run with the reference checkout on `sys.path`.  It swaps exactly the hot-path entry points
 and leaves everything else of the reference untouched:

    lib.cov_mixed.Loss_cov_mixed   -> lc_amd.cov_mixed.Loss_cov_mixed     (also the name `losses` imported, losses.py:15)
    lib.pnp.cer_solver / pnp_ceres -> lc_amd.pnp.cer_solver / pnp_ceres   (registered BEFORE the reference imports them, so the
                                                                          Ceres cffi extension `lib.pnp._ext` is never needed)
    ptnet.softargmax_2d_std        -> lc_amd.ptnet.softargmax_2d_std      (+ ptnet.ptnet.forward's sparse branch fused)
    lib.pnp.cv2_solver             -> lc_amd.pnp.gpu_solver               (only when OpenCV is absent, or on request)
    lib.utils.grad.NormClipper     -> lc_amd.grad.NormClipper             (same constructor and `max_norm` buffer; also the name in `losses`)
    losses.Loss_fn.sparse_kpt_loss / .dense_pose_loss -> the fused-launch methods of lc_amd.losses.Loss_fn (they only use the
                                                          attributes the reference's own Loss_fn instance has)
"""
from __future__ import annotations

import importlib
import runpy
import sys
import types


def install(patch_ptnet: bool = True, gpu_initialiser=None) -> dict:
    """gpu_initialiser: True = also register the RANSAC-P3P kernel as `lib.pnp.cv2_solver` (same `solve` surface,
    `test.py:59,120`); None (default) = only when OpenCV cannot be imported, so that `test.py` runs without it."""
    from . import cov_mixed as cm
    from . import ptnet as head
    from .pnp import cer_solver, gpu_solver, pnp_ceres

    done = {}
    # PnP: register our modules under the reference's names first (lib/ and lib/pnp/ are namespace packages)
    sys.modules["lib.pnp.pnp_ceres"] = pnp_ceres
    sys.modules["lib.pnp.cer_solver"] = cer_solver
    if gpu_initialiser is None:
        try:
            importlib.import_module("cv2")
            gpu_initialiser = False
        except ImportError:
            gpu_initialiser = True
    if gpu_initialiser:
        sys.modules["lib.pnp.cv2_solver"] = gpu_solver
    try:
        pkg = importlib.import_module("lib.pnp")
        pkg.pnp_ceres, pkg.cer_solver = pnp_ceres, cer_solver
        if gpu_initialiser:
            pkg.cv2_solver = gpu_solver
        done["lib.pnp"] = True
    except ImportError:
        done["lib.pnp"] = False
    # loss: patch the defining module and every module that did `from lib.cov_mixed import Loss_cov_mixed`
    try:
        ref_cm = importlib.import_module("lib.cov_mixed")
        ref_cm.Loss_cov_mixed = cm.Loss_cov_mixed
        done["lib.cov_mixed"] = True
    except ImportError:
        done["lib.cov_mixed"] = False
    for name, mod in list(sys.modules.items()):
        if isinstance(mod, types.ModuleType) and name != "lib.cov_mixed" and getattr(mod, "Loss_cov_mixed", None) is not None \
                and mod is not cm:
            mod.Loss_cov_mixed = cm.Loss_cov_mixed
    try:
        ref_losses = importlib.import_module("losses")
        ref_losses.Loss_cov_mixed = cm.Loss_cov_mixed
        done["losses"] = True
    except Exception:  # the reference's losses.py needs scipy/floatbits etc.; absent pieces are the caller's problem
        done["losses"] = False
    # the launch-bound glue of the reference's own Loss_fn: clipper class and the two methods with fused counterparts
    try:
        from . import grad as our_grad
        from . import losses as our_losses

        ref_grad = importlib.import_module("lib.utils.grad")
        ref_grad.NormClipper = our_grad.NormClipper
        if done.get("losses"):
            ref_losses.NormClipper = our_grad.NormClipper
            ref_losses.Loss_fn.sparse_kpt_loss = our_losses.Loss_fn.sparse_kpt_loss
            ref_losses.Loss_fn.dense_pose_loss = our_losses.Loss_fn.dense_pose_loss
        done["loss_glue"] = bool(done.get("losses"))
    except Exception:
        done["loss_glue"] = False
    if patch_ptnet:
        try:
            ref_ptnet = importlib.import_module("ptnet")
            ref_ptnet.softargmax_2d_std = head.softargmax_2d_std
            orig_forward = ref_ptnet.ptnet.forward

            def forward(self, rgb):
                if "kpt_logits" not in self.channel_slices:
                    return orig_forward(self, rgb)
                out_raw, _ = self.net(rgb)  # ptnet.py:55
                return head.sparse_head(out_raw[:, self.channel_slices["kpt_logits"]])  # ptnet.py:59-66 fused

            ref_ptnet.ptnet.forward = forward
            done["ptnet"] = True
        except Exception:
            done["ptnet"] = False
    return done


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit(__doc__)
    script = argv[0]
    import os

    sys.path.insert(0, os.path.dirname(os.path.abspath(script)))
    print("lc_amd.dropin:", install(), file=sys.stderr)
    sys.argv = argv
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
