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
    # top level = SURVEY 8(d)'s figure: algorithmic flops with the batch's measured LM iterations against the fp64 vector peak, formed from the
    # protocol's step; the HBM roofline and the kernel's own VALU-issue ratio (self-derived: labelled, not the headline) ride along
    r = d["roofline"]
    assert r["bound"] == "fp64_vector" and r["unit"] == "TFLOP/s" and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 1.0 <= r["lm_iterations"]["mean"] <= r["lm_iterations"]["max"] <= 50
    per_pose = 115e3 + 23e3 * (r["lm_iterations"]["mean"] + 1)
    assert abs(r["achieved"] - per_pose * 256 / (d["ms_per_step"] * 1e-3) / 1e12) <= 1e-6 * r["achieved"]
    assert r["counters_from"]["file"].startswith("profiles/") and r["traffic"] > 0
    hb = r["hbm"]
    assert hb["bound"] == "hbm" and abs(hb["achieved"] - 5640 * 256 / (d["ms_per_step"] * 1e-3) / 1e9) <= 1e-6 * hb["achieved"]
    vi = r["valu_issue"]
    assert vi["bound"] == "valu_issue" and 0 < vi["frac"] < 1 and abs(vi["achieved"] - d["value"]) <= 1e-6 * d["value"] and "Self-derived" in vi["note"]
    assert "strong" not in d  # one rank: the strong split is the weak one
    # both launch forms at the top level
    assert d["config"]["launch"] in ("graph_region", "fused")
    assert d["value_stream_order"] > 0 and abs(d["value_stream_order"] - 256 / (d["ms_per_step_stream_order"] * 1e-3)) <= 1e-6 * d["value"]
    # the dense configs' hot-path shapes
    for name, n in (("glmo_dense", 1024), ("zlmo_dense", 1849)):
        blk = d["dense"][name]
        assert f"N={n}" in blk["workload"] and blk["value"] > 0 and blk["unit"] == "poses/s"
        for k in ("lc_cov_loss_kernel", "lc_pnp_lm_wide_kernel"):
            assert blk[k]["kernel_us"] > 0 and 0 < blk[k]["hbm_frac"] < 1 and 0 < blk[k]["fp64_vector_frac"] < 1
        two = blk["two_launches"]
        assert two["ms_per_step"] * 1e3 >= 0.5 * (blk["lc_cov_loss_kernel"]["kernel_us"] + blk["lc_pnp_lm_wide_kernel"]["kernel_us"])
        # the step itself is ONE launch: not slower than the two, not faster than the longer of its halves
        assert blk["step"].startswith("one launch") and blk["value"] > two["value"]
        assert blk["ms_per_step"] * 1e3 >= 0.8 * max(blk["lc_cov_loss_kernel"]["kernel_us"], blk["lc_pnp_lm_wide_kernel"]["kernel_us"])
        assert abs(blk["value"] - 32 / (blk["ms_per_step"] * 1e-3)) <= 1e-6 * blk["value"]
        for k in ("lc_cov_loss_kernel", "lc_pnp_lm_wide_kernel"):  # the committed counter pass of this very shape rides along
            assert blk[k]["traffic"] > 0 and blk[k]["counters_from"]["file"].startswith("profiles/") and 0 < blk[k]["valu_active_share_of_wave_cycles"] < 1
    # the test-time chain (8f rows f1 + f2 + a24), replayed as one graph
    # ... at the reference's own knobs: configs/zlmo.yaml:30-37 (16 384 candidates per object, weighted_filtered) and configs/glmo.yaml:28-32
    assert set(d["test_time"]) == {"zlmo", "zlmo_bf16", "glmo", "gsplmo", "hybrid_r03"}
    assert "bf16 maps" in d["test_time"]["zlmo_bf16"]["workload"] and d["test_time"]["zlmo_bf16"]["us_per_call_replayed_200"] <= d["test_time"]["zlmo"]["us_per_call_replayed_200"] * 1.05
    assert "16 keypoints" in d["test_time"]["gsplmo"]["workload"] and d["test_time"]["gsplmo"]["solver"] == "weighted"
    assert "16384 candidates" in d["test_time"]["zlmo"]["workload"] and "quantile_in_mask 0.2" in d["test_time"]["zlmo"]["workload"]
    assert "1024 candidates" in d["test_time"]["glmo"]["workload"] and "quantile 0.3" in d["test_time"]["glmo"]["workload"]
    assert d["test_time"]["zlmo"]["solver"] == "weighted-filtered" and d["test_time"]["glmo"]["solver"] == "weighted"
    for tt in d["test_time"].values():
        assert tt["replay_equals_eager"] and 0 < tt["us_per_call_replayed_200"] <= tt["us_per_call_replayed"] * 1.05 and 0 < tt["us_per_call_replayed"] <= tt["us_per_call_eager"] * 1.05 < 4000  # zlmo is GPU-bound: eager == replayed
        sparse = "keypoints" in tt["workload"]  # 16 points at 0.3 px of noise: depth is known to a few mm at best
        assert tt["median_translation_error_mm"] < (12.0 if sparse else 5.0) and tt["max_translation_error_mm"] < (100.0 if sparse else 25.0) and tt["max_rotation_error"] < 0.1
    assert d["steady_state"]["B"] == 4096 and d["steady_state"]["poses_per_s"] > d["value"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    # one protocol for `value` and the sweep (a sustained loop per thread count): the headline cannot contradict its own sweep
    assert c["value"] >= 0.8 * max(c["poses_per_s_by_threads"].values()) and c["value"] == c["poses_per_s_by_threads"][str(c["cores"])]
    assert all(v["calls"] >= 3 for v in c["sustained"].values()) and "1" in c["poses_per_s_by_threads"]
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
    assert d["config"]["launch_agreed_by_all_ranks"] is True and d["value_stream_order"] > 0
    # the strong split beside the weak headline ("32/GPU at B=256, 8 GPUs"): ONE global batch of 256 over the ranks, same protocol
    st = d["strong"]
    assert st["scaling"] == "strong" and st["ranks_seen"] == 2 and st["global_batch"] == 256 and st["per_rank_batch"] == 128 and len(st["per_rank_ms_per_step"]) == 2
    assert abs(st["value"] - 256 / (st["ms_per_step"] * 1e-3)) <= 1e-6 * st["value"] and d["scaling"] == "weak"


def test_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the script starts the two ranks itself (the driver's command form)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["LC_BENCH_SHARE_GPU"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "3", "--regions", "5"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # ONE line, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and len(d["per_rank_ms_per_step"]) == 2 and d["config"]["global_batch"] == 512
    assert d["roofline"]["kernel_us"]["lc_pose_unit_kernel"] > 0  # the N-rank line keeps the per-rank kernel time


def test_gpus_8_on_one_shared_gpu_and_a_rank_that_dies():
    """The whole of the 8-rank control flow that can be rehearsed without an 8-GPU node: `python bench.py --gpus 8` (the driver's command
    form, no launcher) with every rank on the one GPU -- eight region vectors through the aggregation, the 8-way MIN agreement on the
    launch form, one JSON line, rc 0; and the failure leg: a rank that exits non-zero before the first barrier makes the parent return
    non-zero well inside the 300 s gloo timeout instead of hanging on the barrier."""
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["LC_BENCH_SHARE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "3", "--regions", "5"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["ranks_seen"] == 8 and len(d["per_rank_ms_per_step"]) == 8 and d["config"]["global_batch"] == 2048
    assert d["config"]["launch_agreed_by_all_ranks"] is True and d["collective_backend"] == "gloo" and d["scaling"] == "weak"
    assert abs(d["value"] - 2048 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"] and d["ms_per_step"] >= 0.5 * max(d["per_rank_ms_per_step"])
    assert len(d["timing"]["region_ms_per_step"]) == 5 and "cpu_baseline" not in d
    st = d["strong"]  # both conventions carry all eight ranks
    assert st["ranks_seen"] == 8 and st["per_rank_batch"] == 32 and st["global_batch"] == 256 and abs(st["value"] - 256 / (st["ms_per_step"] * 1e-3)) <= 1e-6 * st["value"]
    t0 = time.time()
    bad = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(env, LC_BENCH_FAIL_RANK="5"))
    assert bad.returncode != 0 and time.time() - t0 < 280, (bad.returncode, time.time() - t0)
    assert not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]  # no result line from a run that lost a rank
    assert "LC_BENCH_FAIL_RANK" in bad.stderr


def test_gpus_1_through_the_launcher_equals_the_plain_run():
    """N = 1 under torchrun (WORLD_SIZE=1) takes the same in-process path as the plain command."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    args = ["--gpus", "1", "--steps", "20", "--warmup", "3", "--regions", "5", "--no-cpu-baseline", "--no-head", "--workload", "metric",
            "--steady-batch", "0"]
    a = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900)
    b = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900)
    assert a.returncode == 0 and b.returncode == 0, (a.stderr[-1500:], b.stderr[-1500:])
    da, db = _last_json(a.stdout), _last_json(b.stdout)
    assert da["n_gpus"] == db["n_gpus"] == 1 and da["config"] == db["config"] and set(da) == set(db)
    assert 0.8 < da["value"] / db["value"] < 1.25


def test_rccl_code_path_with_one_rank():
    """One rank under a launcher with the timing collectives forced on: group creation bound to the device, the probe all-reduce and
    the barrier run over RCCL (backend "nccl") exactly as every rank of an 8-GPU run executes them."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, LC_BENCH_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "3", "--regions", "5",
                          "--no-cpu-baseline", "--no-head", "--workload", "metric", "--steady-batch", "0"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = _last_json(out.stdout)
    assert d["n_gpus"] == 1 and d["collective_backend"] == "nccl" and d["rccl_version"] and "rccl_error" not in d, (d.get("rccl_error"), out.stderr[-1500:])
    assert d["config"]["launch"] == "graph_region"  # the capture coexists with the RCCL group's watchdog thread
