import re, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tests.test_gpu_dense import _train_shape_record
from tests.util import rel_err
from lc_amd import synth
for kind in ("dense_glmo","bin_zlmo"):
    for hd, ls in ((torch.float16,4096.0),(torch.bfloat16,1.0),(None,1.0)):
        z, rec = _train_shape_record(kind, hd, ls)
        for k in rec:
            if k=="steps": continue
            if re.match(r"s\d+_w?loss_", k):
                e=abs(float(rec[k]) - float(z[k]))/max(1.0, abs(float(z[k])))
            elif "_grad_" in k:
                got, want = np.asarray(rec[k], np.float64), np.asarray(z[k], np.float64)
                nf=0
                if k.endswith("_grad_xyz_noc") and hd is not None:
                    gt, out = synth.train_inputs(kind, seed=int(k[1:k.index("_")]))
                    m, x, t = gt["msk_noc"][:, None].float(), out["xyz_noc"], gt["xyz_noc_tgt"]
                    flips = (torch.sign(x * m - t) != torch.sign(x.to(hd).float() * m - t)).numpy()
                    nf=int(flips.sum())
                    got, want = np.where(flips, 0.0, got), np.where(flips, 0.0, want)
                e=np.abs(got - want).max() / np.abs(want).max()
                k=f"{k} flips={nf} max={np.abs(want).max():.3e}"
            else:
                e=rel_err(rec[k], z[k])
            print(kind, hd, k, f"{e:.3e}")
