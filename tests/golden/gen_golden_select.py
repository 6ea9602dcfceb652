#!/usr/bin/env python3
"""Golden vectors for the test-time point selection: runs the REFERENCE's `quantile_msk`
(/root/reference/test.py:39-45).  test.py cannot be imported in this container (it imports cv2 through lib.pnp.cv2_solver),
so the generator compiles that one function from the reference file's syntax tree at generation time -- nothing of the
reference's text is stored here; only inputs and outputs are committed (select_*.npz).

    python tests/golden/gen_golden_select.py        # needs /root/reference
"""
import ast
import os
from typing import Union

import numpy as np
import torch
from torch import Tensor

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/test.py"


def reference_function(name):
    tree = ast.parse(open(REF).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    ns = {"torch": torch, "Tensor": Tensor, "Union": Union}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF, "exec"), ns)
    return ns[name]


def main():
    quantile_msk = reference_function("quantile_msk")
    g = torch.Generator().manual_seed(7)
    for name, (B, N, q) in {"q50_B4_N64": (4, 64, 0.5), "q80_B3_N1024": (3, 1024, 0.8), "q05_B2_N37": (2, 37, 0.05),
                            "q100_B2_N16": (2, 16, 1.0), "q0_B2_N16": (2, 16, 0.0),
                            "q20_B2_N16384": (2, 16384, 0.2)}.items():  # configs/zlmo.yaml:30-37: 128x128 candidates per object, quantile 0.2
        inv_std = torch.rand(B, N, 2, generator=g) * 2 + 0.01
        seg = torch.rand(B, N, generator=g) > 0.4
        out = {"in_inv_std": inv_std.numpy(), "in_seg": seg.numpy(), "q": np.float64(q),
               "msk_quantile": quantile_msk(inv_std, q).numpy()}
        # quantile_in_mask (test.py:101-104): the reference's own expressions around its quantile_msk
        vis_ratio = seg.float().mean(dim=-1)
        quantile = 1 - (1 - q) * vis_ratio
        out["msk_quantile_in_mask"] = (quantile_msk(inv_std * seg[..., None], quantile) * seg).numpy()
        np.savez_compressed(os.path.join(HERE, f"select_{name}.npz"), **out)
        print(name, out["msk_quantile"].sum(-1), out["msk_quantile_in_mask"].sum(-1))


if __name__ == "__main__":
    main()
