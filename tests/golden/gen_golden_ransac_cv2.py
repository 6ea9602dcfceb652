#!/usr/bin/env python3
"""Pin the PnP initialiser's ROLE to the REAL reference (cv2.solvePnPRansac behind lib/pnp/cv2_solver.py:69-88; OpenCV 4.6.0.66 is
pinned in the reference's scripts/req_0.txt:13).

This cannot run in the build container (no OpenCV: `import cv2` fails), which is why parity at this boundary is "unpinned".  On ANY
machine with the reference checkout and `opencv-python`, one run of

    LC_REFERENCE=/path/to/fulliu-lc python tests/golden/gen_golden_ransac_cv2.py

runs the reference's own `lib.pnp.cv2_solver.solve` on the committed `ransac_*.npz` problem sets (inputs are read from those files, so
they are bit-identical wherever the generator runs) and writes tests/golden/ransac_cv2_<case>.npz -- per pose the reference's validity
flag, state (w,x,y,z,tx,ty,tz) and inlier index set (as a mask) -- after which tests/test_oracle_ransac_cv2_golden.py (CPU: the oracle) and
tests/test_gpu_ransac_cv2_golden.py (GPU: the HIP kernels) stop skipping.  OpenCV's RNG and EPnP cannot be reproduced bit for bit, so
what those tests hold is the ROLE contract of test.py:59,120: our pose lies in the same LM basin (the
refinement from either start ends at the same pose within 1e-4) and names the same consensus set (inlier IoU).  Data only is stored,
never reference source.
"""
import glob
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("LC_REFERENCE", "/root/reference")
sys.path.insert(0, REF)


def problem_sets():
    return sorted(p for p in glob.glob(os.path.join(HERE, "ransac_*.npz")) if not os.path.basename(p).startswith("ransac_cv2_"))


def main(out_dir=HERE):
    try:
        import cv2
        from lib.pnp import cv2_solver  # the reference's own wrapper around cv2.solvePnPRansac
    except Exception as e:  # noqa: BLE001
        sys.exit(f"cannot import OpenCV / the reference's lib.pnp.cv2_solver from {REF}: {e!r} -- nothing was written")
    for path in problem_sets():
        z = np.load(path)
        name = os.path.basename(path)[len("ransac_"):-4]
        K, X, U, counts = z["in_K"], z["in_pts3d"], z["in_pts2d"], z["in_counts"]
        B, N = X.shape[:2]
        # the reference's list form (cv2_solver.py:31-38): one array per pose, ragged
        take = lambda a: [torch.from_numpy(a[i, :int(counts[i])].copy()) for i in range(B)]  # noqa: E731
        invalids, states, inliers = cv2_solver.solve([torch.from_numpy(k.copy()) for k in K], take(X), take(U),
                                                     reprojectionError=float(z["in_reproj_err"]), num_workers=1)
        mask = np.zeros((B, N), np.uint8)
        for i, idx in enumerate(inliers):
            mask[i, np.asarray(idx, np.int64)] = 1
        out = os.path.join(out_dir, f"ransac_cv2_{name}.npz")
        np.savez_compressed(out, invalid=np.asarray([bool(v) for v in invalids]), states=np.stack([np.asarray(s, np.float64) for s in states]),
                            inlier_mask=mask, cv2_version=np.asarray(getattr(cv2, "__version__", "unknown")),
                            problem_set=np.asarray(os.path.basename(path)))
        print(f"{out}: {B} poses, {int(sum(bool(v) for v in invalids))} invalid, inliers {mask.sum(1).tolist()}")


if __name__ == "__main__":
    main()
