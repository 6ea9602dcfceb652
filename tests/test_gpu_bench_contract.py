"""bench.py's output contract on the GPU box: the single-GPU JSON line (metric / roofline / steady_state / cpu_baseline keys) and
the N = 2 control flow (barriers, per-rank region times, MAX over ranks, ranks_seen) with both ranks sharing the one GPU of
the test box (LC_BENCH_SHARE_GPU=1: the timing collectives then run over gloo; RCCL refuses two ranks per device)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_single_gpu_line_has_the_contract_keys():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "3", "--regions", "5", "--cpu-budget", "2",
                          "--steady-batch", "4096"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["metric"] == base["metric"] and d["unit"] == "poses/s" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "configs[1]" in d["config"]["workload"] and "model" not in d["config"]
    assert abs(d["value"] - 256 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("valu_issue", "hbm") and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    if r["bound"] == "valu_issue":
        assert r["hbm"]["bound"] == "hbm" and r["counters_from"]["file"].startswith("profiles/") and r["traffic"] > 0
    assert d["steady_state"]["B"] == 4096 and d["steady_state"]["poses_per_s"] > d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert d["ranks_seen"] == 1 and len(d["per_rank_ms_per_step"]) == 1
    assert d["head"]["roofline"]["bound"] == "hbm"


def test_two_rank_control_flow_on_one_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, LC_BENCH_SHARE_GPU="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3",
                          "--regions", "5"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["collective_backend"] == "gloo" and len(d["per_rank_ms_per_step"]) == 2
    assert d["config"]["global_batch"] == 512 and "cpu_baseline" not in d and "steady_state" not in d
    assert abs(d["value"] - 512 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["ms_per_step"] >= max(d["per_rank_ms_per_step"]) * 0.5  # the reported step is a MAX-over-ranks region, not a mean
