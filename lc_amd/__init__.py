"""lc_amd -- MI355X-native hot path of fulliu/lc (linear-covariance pose loss, weighted PnP, keypoint soft-argmax).

Call surface (mirrors the reference):
    lc_amd.cov_mixed.Loss_cov_mixed      <- lib/cov_mixed.py:100
    lc_amd.pnp.cer_solver.solve          <- lib/pnp/cer_solver.py:6
    lc_amd.pnp.pnp_ceres.solve           <- lib/pnp/pnp_ceres.py:6
    lc_amd.ptnet.softargmax_2d_std       <- ptnet.py:100
    lc_amd.losses.Loss_fn                <- losses.py:239
Native code: lc_amd/csrc/*.hip -> lc_amd/_C/liblc_amd.so (C ABI in include/lc_amd.h).
"""
__version__ = "0.1.0"
