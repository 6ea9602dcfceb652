"""The clipping oracle (oracle/grad_oracle.py) and the host logic of lc_amd.grad.NormClipper (on the oracle backend)
against golden trajectories of the reference's NormClipper."""
import numpy as np
import pytest
import torch

from tests.cpu_backend import oracle_backend  # noqa: F401
from tests.util import golden_files, case_name

FILES = golden_files("clip_")


def _kwargs(z):
    return {k[3:]: z[k].item() for k in z.files if k.startswith("kw_")}


@pytest.mark.parametrize("path", FILES, ids=[case_name(p, "clip_") for p in FILES])
@pytest.mark.parametrize("tag,dtype,tol", [("f64", torch.float64, 1e-12), ("f32", torch.float32, 2e-6)])
def test_clipper_on_oracle_backend_vs_reference(path, tag, dtype, tol, oracle_backend):  # noqa: F811
    from lc_amd.grad import NormClipper

    z = np.load(path)
    clip = NormClipper(**_kwargs(z))
    for i in range(int(z["steps"])):
        g = torch.from_numpy(z[f"in_{i}"]).to(dtype)
        out = clip.clip(g)
        ref = torch.from_numpy(z[f"{tag}_out_{i}"])
        assert (out - ref).abs().max() <= tol * max(1.0, ref.abs().max().item())
        assert abs(clip.max_norm.item() - z[f"{tag}_max_norm_{i}"]) <= tol * max(1.0, abs(z[f"{tag}_max_norm_{i}"]))
        assert abs(float(clip.last_norm) - z[f"{tag}_last_norm_{i}"]) <= tol * max(1.0, abs(z[f"{tag}_last_norm_{i}"]))
    assert set(clip.state_dict()) == {"max_norm"}
