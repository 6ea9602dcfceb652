"""The HIP PnP initialiser + the inlier refinement (through the C ABI) against vectors of the reference's `cv2_solver.solve`, when
tests/golden/ransac_cv2_*.npz exist (tests/golden/gen_golden_ransac_cv2.py; SKIPS until someone with OpenCV runs it)."""
import os

import numpy as np
import pytest
import torch

from tests.ransac_role import CV2, check_role

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def kernel_ransac(K, X, U, counts, thr, iterations, seed):
    from lc_amd.pnp import gpu_solver

    t = [torch.from_numpy(np.ascontiguousarray(a)).to(DEV) for a in (K, X, U, counts)]
    st, inl, bad = gpu_solver.solve_device(*t, reprojectionError=thr, iterations=iterations, seed=seed, refine=False)
    return st.cpu().numpy().astype(np.float64), inl.cpu().numpy().astype(bool), bad.cpu().numpy().astype(bool)


def kernel_refine(K, X, U, counts, mask, start):
    from lc_amd.pnp import pnp_ceres

    t = [torch.from_numpy(np.ascontiguousarray(a)).to(DEV) for a in (K, X, U)]
    st, _, ret = pnp_ceres.solve_device(*t, None, torch.from_numpy(start).to(DEV), torch.from_numpy(counts.astype(np.int32)).to(DEV), max_iter_count=20,
                                        weight_mask=torch.from_numpy(mask.astype(np.uint8)).to(DEV))
    return st.cpu().numpy(), ret.cpu().numpy()


@pytest.mark.skipif(not CV2, reason="no tests/golden/ransac_cv2_*.npz (OpenCV not in the build image): initialiser parity unpinned")
@pytest.mark.parametrize("path", CV2, ids=[os.path.basename(p) for p in CV2])
def test_kernel_vs_cv2_golden(path):
    print(check_role(path, kernel_ransac, kernel_refine))
