"""Two- and EIGHT-rank DistributedDataParallel steps on the REAL kernels (all ranks on the one GPU of the test box, collectives over gloo; eight is
the rank count of BASELINE configs[3] / [4], whose 8-GPU form cannot be run from here):
examples/train_sparse_ddp.py and examples/train_dense_ddp.py against the single-process run on the concatenated crops.

What a sharded job must preserve (reference: train.py:57-67 is one process; lib/utils/grad.py:19-30,66-68 takes the clipping norm
over the whole batch and keeps it in a checkpointed buffer):
  (i)   finite losses;
  (ii)  identical parameters on both ranks after the steps (DDP's all-reduced gradients);
  (iii) NormClipper.max_norm equal on both ranks AND equal to the single-process run on the concatenated batch -- the squared norm
        is all-reduced and scaled to the job's mean loss (`shard_loss_scale`), so the buffer does not depend on the GPU count;
  (iii') the ZebraPose code histogram (`Loss_xyz_bin.histogram`, also checkpointed) likewise: error and pixel counts all-reduced;
  (iv)  the job's loss (mean over ranks) and parameters follow the single-process run (batch-norm statistics frozen: per-rank batch
        statistics are the one thing a sharded step cannot share without SyncBN)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS = 6


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(script, extra, dump, ranks, emulate=2):
    common = [os.path.join(ROOT, "examples", script), "--steps", str(STEPS), "--batch", "4", "--width", "16", "--bn-eval", "--dump", dump, *extra]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    if ranks > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
               "--master-port", str(_port()), *common, "--backend", "gloo", "--share-gpu"]
    else:
        cmd = [sys.executable, *common, "--emulate-ranks", str(emulate)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    return [torch.load(f"{dump}.rank{r}.pt") for r in range(ranks)]


@pytest.mark.parametrize("script,extra,world", [("train_sparse_ddp.py", ["--sparse-cnt", "16", "--fp32"], 2),
                                                ("train_dense_ddp.py", ["--dtype", "fp32", "--np-seed", "3"], 2),
                                                ("train_dense_ddp.py", ["--dtype", "fp32", "--np-seed", "3"], 8),
                                                ("train_dense_ddp.py", ["--dtype", "fp32", "--np-seed", "3", "--bin"], 8),
                                                ("train_dense_ddp.py", ["--dtype", "fp32", "--np-seed", "3", "--zlmo", "--batch", "2"], 8)],
                         ids=["sparse-2", "dense-2", "dense-8", "binary-code-8", "zlmo-shape-8"])
def test_ranks_match_the_single_process_run(tmp_path, script, extra, world):
    ranks = _run(script, extra, str(tmp_path / "many"), world)
    (one,) = _run(script, extra, str(tmp_path / "one"), 1, emulate=world)
    r0, r1 = ranks[0], ranks[-1]
    # (i)
    for d in (*ranks, one):
        assert len(d["losses"]) == STEPS and all(map(lambda v: v == v and abs(v) < 1e9, d["losses"]))
    # (ii) bit for bit, every rank
    assert all(torch.equal(r0["params"], r["params"]) for r in ranks[1:])
    # (iii)
    assert all(r0["clip_states"] == r["clip_states"] for r in ranks[1:])
    if "dense" in script:
        assert any(v > 0 for v in r0["final_clip"].values())  # the hooks ran
    for a, b in zip(r0["clip_states"], one["clip_states"]):
        assert a.keys() == b.keys()
        for k in a:
            assert abs(a[k] - b[k]) <= 2e-3 * max(abs(b[k]), 1e-6), (k, a[k], b[k])
    if "--bin" in extra or "--zlmo" in extra:  # the code histogram (losses.py:203-208, a checkpointed buffer): the whole batch's on every rank, i.e. the single process'
        h = [r["loss_state"]["xyz_bin_loss_fn.histogram"] for r in ranks]
        assert all(torch.equal(h[0], x) for x in h[1:]) and float((h[0] - 0.5).abs().max()) > 1e-3
        assert (h[0] - one["loss_state"]["xyz_bin_loss_fn.histogram"]).abs().max() <= 1e-6
    # (iv)
    job = [sum(r["losses"][i] for r in ranks) / world for i in range(STEPS)]
    for x, y in zip(job, one["losses"]):
        assert abs(x - y) <= 2e-3 * max(abs(y), 1.0), (job, one["losses"])
    # Adam normalises every gradient entry, so round-off in a near-zero entry moves its weight by up to lr per step in either run:
    # compare the UPDATES as vectors (same start: torch.manual_seed(0) on every rank and in the single process)
    assert torch.equal(r0["params_at_start"], one["params_at_start"])
    u2, u1 = r0["params"] - r0["params_at_start"], one["params"] - one["params_at_start"]
    assert u1.norm() > 0 and ((u2 - u1).norm() / u1.norm()).item() <= 0.05, ((u2 - u1).norm() / u1.norm()).item()


def test_comm_report_of_an_eight_rank_zlmo_shaped_job(tmp_path):
    """`--report-comm` (examples/ddp_common.py: CommReport) on the 8-rank shared-GPU rehearsal of the zlmo-shaped step: the ARITHMETIC of the
    report, not its timings (gloo stages every collective through the host and the ranks share one GPU) -- per step and rank the gradient
    payload is the model's parameter bytes, the buckets add up to it, the loss side issues exactly two small all-reduces (the NormClipper's
    squared norm: 4 bytes; the code histogram's C + 1 int64 counts: 8 x 22 bytes), and the xGMI estimates printed beside them are SURVEY.md
    section 5's formulas (ring: 2 (w-1)/w of the payload over one 153 GB/s link; direct reduce-scatter + all-gather: 2/w of it per link)."""
    import json

    dump = str(tmp_path / "comm")
    ranks = _run("train_dense_ddp.py", ["--dtype", "fp32", "--np-seed", "3", "--zlmo", "--batch", "2", "--report-comm"], dump, 8)
    rep = json.load(open(dump + ".comm.json"))
    s, per_rank = rep["summary"], rep["per_rank_steps"]
    n_params = ranks[0]["params"].numel()
    assert s["world"] == 8 and s["backend"] == "gloo" and len(per_rank) == 8 and all(len(r) == STEPS for r in per_rank)
    assert s["grad_payload_bytes"] == 4 * n_params
    for r in per_rank:
        for d in r:
            assert d["grad_bytes"] == 4 * n_params and d["buckets"] >= 1 and len(d["bucket_ms"]) == d["buckets"]
            assert d["small_calls"] == 2 and d["small_bytes"] == 4 + 8 * (21 + 1)
            assert 0 < d["grad_window_ms"] <= d["step_ms"] and all(v > 0 for v in d["bucket_ms"])
    gb = 4 * n_params / 1e9
    assert abs(s["xgmi_estimate"]["ring_ms"] - 2 * 7 / 8 * gb / 153.0 * 1e3) < 1e-9 and abs(s["xgmi_estimate"]["direct_ms"] - 2 / 8 * gb / 153.0 * 1e3) < 1e-9
    assert abs(s["survey_estimate_104MB_8gpu"]["ring_ms"] - 1.19) < 0.01 and abs(s["survey_estimate_104MB_8gpu"]["direct_ms"] - 0.17) < 0.005  # SURVEY.md section 5's figures
