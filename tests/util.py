import glob
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def case_name(path, prefix):
    return os.path.basename(path)[len(prefix):-4]


def load_loss_case(path, dtype):
    """Inputs of a lc_loss_*.npz fixture cast to `dtype`, plus kwargs."""
    z = np.load(path)
    ins = {k[3:]: torch.from_numpy(z[k]).to(dtype) for k in z.files if k.startswith("in_")}
    kwargs = {k[3:]: z[k].item() for k in z.files if k.startswith("kw_")}
    return z, ins, kwargs, bool(z["want_pts3d"])


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()
