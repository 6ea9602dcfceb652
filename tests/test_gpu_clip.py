"""GPU parity of the fused NormClipper launches (lc_clip.hip) against golden trajectories of the reference's class."""
import numpy as np
import pytest
import torch

from tests.util import golden_files, case_name

pytestmark = pytest.mark.gpu
FILES = golden_files("clip_")
DEV = "cuda:0"


@pytest.mark.parametrize("path", FILES, ids=[case_name(p, "clip_") for p in FILES])
def test_clipper_vs_reference_trajectory(path):
    from lc_amd.grad import NormClipper

    z = np.load(path)
    kw = {k[3:]: z[k].item() for k in z.files if k.startswith("kw_")}
    clip = NormClipper(**kw).to(DEV)
    for i in range(int(z["steps"])):
        out = clip.clip(torch.from_numpy(z[f"in_{i}"]).to(DEV))
        ref = torch.from_numpy(z[f"f64_out_{i}"])
        assert (out.cpu().double() - ref).abs().max() <= 2e-6 * max(1.0, ref.abs().max().item())
        assert abs(clip.max_norm.item() - z[f"f64_max_norm_{i}"]) <= 2e-6 * max(1.0, abs(z[f"f64_max_norm_{i}"]))
        assert abs(float(clip.last_norm) - z[f"f64_last_norm_{i}"]) <= 2e-6 * max(1.0, abs(z[f"f64_last_norm_{i}"]))
    assert clip.max_norm.is_cuda and set(clip.state_dict()) == {"max_norm"}


def test_clipper_large_tensor_lists_and_determinism():
    """8 M elements (multi-block reduction with the last-block pattern), a list of tensors, unaligned views; two runs agree bitwise."""
    from lc_amd.grad import NormClipper
    from oracle import grad_oracle as orc

    g = torch.Generator().manual_seed(0)
    big = torch.randn(8_000_003, generator=g)
    parts = [big[:5_000_001].clone(), big[5_000_001:].clone(), torch.randn(7, generator=g)]
    outs = []
    for _ in range(2):
        clip = NormClipper(initial_max_norm=50.0).to(DEV)
        res = clip.clip([p.to(DEV) for p in parts])
        res2 = clip.clip([p.to(DEV)[1:] * 3 for p in parts])  # odd offsets: scalar path
        outs.append([r.cpu() for r in res + res2] + [clip.max_norm.cpu()])
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    d = [p.double() for p in parts]
    ref, st, norm = orc.apply(d, orc.sum_of_squares(d), torch.tensor(-1.0, dtype=torch.float64), 50.0, 1.7, 0.1)
    for a, b in zip(outs[0][:3], ref):
        assert (a.double() - b).abs().max() <= 2e-6 * b.abs().max()
    d2 = [p.double()[1:] * 3 for p in parts]
    ref2, st2, _ = orc.apply(d2, orc.sum_of_squares(d2), st, 50.0, 1.7, 0.1)
    assert abs(outs[0][-1].item() - st2.item()) <= 2e-6 * st2.item()


def test_clipper_rejects_other_norms_and_cpu_tensors():
    from lc_amd.grad import NormClipper

    clip = NormClipper().to(DEV)
    with pytest.raises(NotImplementedError):
        clip.clip(torch.ones(4, device=DEV), norm_type=float("inf"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        clip.clip(torch.ones(4))


def test_clipper_state_recurrence_survives_hipgraph_replay():
    """max_norm is updated in place (fixed addresses): replaying a captured hook call advances the EMA exactly like eager calls."""
    from lc_amd.grad import NormClipper

    g = torch.Generator().manual_seed(1)
    grads = [(torch.randn(4, 2, 16, 16, generator=g) * s).to(DEV) for s in (1.0, 5.0, 0.3, 20.0, 1.0)]
    eager = NormClipper().to(DEV)
    ref_out, ref_state = [], []
    for x in grads:
        ref_out.append(eager.clip(x).clone())
        ref_state.append(eager.max_norm.item())

    graphed = NormClipper().to(DEV)
    static_in = grads[0].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graphed._ws(static_in.device)  # workspace allocated outside the capture
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    from lc_amd.inference import quiet_capture

    graph = torch.cuda.CUDAGraph()
    with quiet_capture(), torch.cuda.graph(graph):
        static_out = graphed.clip(static_in)
    graphed.max_norm.fill_(-1.0)  # the capture itself does not execute: start the recurrence from scratch
    for x, o, st in zip(grads, ref_out, ref_state):
        static_in.copy_(x)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(static_out, o) and graphed.max_norm.item() == st
