#!/usr/bin/env python3
"""Pin the weighted-PnP solve to the REAL reference (Ceres 2.1.0 behind lib/pnp/cxx/ceres.cpp).

This cannot run in the build container (no Ceres/Eigen/glog/cffi: `import lib.pnp.pnp_ceres` fails with
"No module named 'lib.pnp._ext'"), which is why PnP parity is "unpinned" today.  On ANY machine where the reference's
extension has been built (scripts/build-ceres.sh + lib/pnp/setup_ceres.py of fulliu/lc), one run of

    LC_REFERENCE=/path/to/fulliu-lc python tests/golden/gen_golden_pnp_ceres.py

writes tests/golden/pnp_ceres_<case>.npz (inputs + the reference's `states, result_tr, rets`), and
tests/test_oracle_pnp_ceres_golden.py (CPU: the oracle) and tests/test_gpu_pnp_ceres_golden.py (GPU: the HIP kernel)
stop skipping and assert parity against them.  Inputs come from tests/pnp_cases.py (seeded, torch CPU), so they are
bit-identical wherever the generator runs; data only is stored, never reference source.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = os.environ.get("LC_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

from tests.pnp_cases import PNP_CASES, pnp_case  # noqa: E402


def main(out_dir=HERE):
    try:
        from lib.pnp import pnp_ceres  # the reference's cffi binding of ceres.cpp (needs lib/pnp/_ext built against Ceres 2.1.0)
    except Exception as e:  # noqa: BLE001
        sys.exit(f"cannot import the reference's Ceres extension from {REF}: {e!r}\n"
                 f"build it first (scripts/build-ceres.sh, python lib/pnp/setup_ceres.py) -- nothing was written")
    for name in PNP_CASES:
        c = pnp_case(name)
        B = len(c["start"])
        n = c["counts"]
        # the reference's list form (pnp_ceres.py:43-52): one contiguous array per job, ragged
        lists = lambda a: [torch.from_numpy(a[i, :max(int(n[i]), 1)].copy()) for i in range(B)]  # noqa: E731
        states, result_tr, rets = pnp_ceres.solve([torch.from_numpy(k) for k in c["K"]], lists(c["pts3d"]), lists(c["pts2d"]),
                                                  lists(c["sqrtL"]), [torch.from_numpy(s.copy()) for s in c["start"]],
                                                  [int(v) for v in n], max_iter_count=c["max_iter"], num_workers=1,
                                                  function_tolerance=c["ftol"])
        out = os.path.join(out_dir, f"pnp_ceres_{name}.npz")
        np.savez_compressed(out, **{f"in_{k}": v for k, v in c.items()}, states=np.asarray(states, np.float32),
                            result_tr=np.asarray(result_tr, np.float32), rets=np.asarray(rets, np.int32))
        print(f"{out}: {B} jobs, {int(np.asarray(rets).sum())} invalid")


if __name__ == "__main__":
    main()
