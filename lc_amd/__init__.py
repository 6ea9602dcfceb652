"""lc_amd -- MI355X-native hot path of fulliu/lc (linear-covariance pose loss, weighted PnP, keypoint soft-argmax).

Call surface (mirrors the reference):
    lc_amd.cov_mixed.Loss_cov_mixed      <- lib/cov_mixed.py:100
    lc_amd.pnp.cer_solver.solve          <- lib/pnp/cer_solver.py:6
    lc_amd.pnp.pnp_ceres.solve           <- lib/pnp/pnp_ceres.py:6
    lc_amd.ptnet.softargmax_2d_std       <- ptnet.py:100
    lc_amd.losses.Loss_fn                <- losses.py:239
    lc_amd.grad.NormClipper              <- lib/utils/grad.py:5
    lc_amd.kpt.kpt_nll_mean              <- losses.py:318 (sparse_kpt_loss)
    lc_amd.dense.dense_front_end / dense_select   <- losses.py:142-161,355-356 / test.py:39-45,94-113
    lc_amd.floatbits                     <- floatbits.py (ZebraPose codes)
    lc_amd.pnp.gpu_solver.solve          <- lib/pnp/cv2_solver.py:8 (RANSAC initialiser)
    lc_amd.metrics.compute_pose_errors   <- lib/utils/evaluate.py:333
    lc_amd.inference.solve_pnp           <- test.py:47-136
    lc_amd.graphs.GraphedLoss / inference.GraphedSolvePnP   hipGraph replay of the launch-bound steps
    lc_amd.dropin                        run the reference's train.py / test.py on all of the above without editing them
Native code: lc_amd/csrc/*.hip -> lc_amd/_C/liblc_amd.so (C ABI in include/lc_amd.h).
"""
__version__ = "0.1.0"
