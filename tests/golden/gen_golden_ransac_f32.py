#!/usr/bin/env python3
"""Records of the HIP PnP initialiser itself (run ON THE MI355X: `gpurun -- 'python tests/golden/gen_golden_ransac_f32.py'`, the files land in
gpurun_out/ and are copied into tests/golden/): for three of the committed `ransac_*` problem sets the kernel's float32 hypotheses (read back
from the split form's workspace), its per-hypothesis inlier counts and the BITS of its per-hypothesis inlier error (the chunk partials summed
in chunk order, as the selection kernel sums them), and its outputs (winner, inlier count, inlier mask, validity).

What they are for: `tests/test_oracle_ransac.py` (CPU, no GPU needed) runs the float32-faithful oracle (`oracle/p3p_ransac_oracle.py:
ransac_f32`, a numpy restatement of the kernel's division-free scoring) on the stored hypotheses and must reproduce every stored integer and
every stored error bit -- so the claim "oracle and kernel agree exactly" is re-checked wherever the CPU suite runs, and a change to either
side that breaks it shows up without a GPU.  These are KERNEL records, not reference vectors (the reference's OpenCV RANSAC cannot be matched
bit for bit; see gen_golden_ransac_cv2.py for that boundary)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
CASES = ("clean_B16_N32", "outliers_B24_N64", "ragged_B8_N40")


def main(out_dir=os.path.join(ROOT, "gpurun_out")):
    from lc_amd.pnp import gpu_solver

    dev = torch.device("cuda:0")
    os.makedirs(out_dir, exist_ok=True)
    for name in CASES:
        z = np.load(os.path.join(HERE, f"ransac_{name}.npz"))
        K, X, U, counts = (torch.from_numpy(z["in_" + k]).to(dev) for k in ("K", "pts3d", "pts2d", "counts"))
        thr, iters, seed = float(z["in_reproj_err"]), int(z["in_iterations"]), int(z["in_seed"])
        ws = []
        outs = [gpu_solver.solve_device(K, X, U, counts, reprojectionError=thr, iterations=iters, seed=seed, refine=False, return_hypothesis=True, split=s,
                                        workspace_out=ws if s else None) for s in (True, False)]
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(*outs)), "the two launch forms disagree"
        st, inl, bad, hyp, n_in = (t.cpu().numpy() for t in outs[0])
        B, N = X.shape[:2]
        hyp64, hyp32, pc, pe = (v.cpu().numpy() for v in gpu_solver.workspace_views(ws[0], B, N, iters))
        H = hyp32.shape[1]
        cnt, err = np.zeros((B, H), np.int32), np.zeros((B, H), np.float32)
        for b in range(B):
            C = (max(int(counts[b]), 0) + 63) // 64 if int(counts[b]) >= 4 else 0
            for c in range(C):
                cnt[b] += pc[b, c]
                err[b] = (err[b] + pe[b, c]).astype(np.float32)
        path = os.path.join(out_dir, f"kernel_ransac_f32_{name}.npz")
        np.savez_compressed(path, problem_set=np.asarray(f"ransac_{name}.npz"), hyp32=hyp32, hyp64_winner=np.stack([hyp64[b, max(int(hyp[b]), 0)] for b in range(B)]),
                            per_hyp_count=cnt, per_hyp_err_bits=err.view(np.uint32), best_hyp=hyp.astype(np.int32), n_inliers=n_in.astype(np.int32),
                            inlier_mask=inl.astype(np.uint8), invalid=bad.astype(np.uint8), states=st)
        print(path, os.path.getsize(path) // 1024, "KiB; invalid", int(bad.sum()), "winners", hyp.tolist())


if __name__ == "__main__":
    main()
