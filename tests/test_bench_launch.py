"""bench.py's `--gpus N` handling on CPU: the pure decision (`launch_plan`), the refusal of a WORLD_SIZE that contradicts
--gpus, and the self-launch itself -- without a GPU the N child ranks exit non-zero ("needs an MI355X"), which must come back as
the parent's return code (the parent relays; it never runs a rank itself when N > 1)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launch_plan():
    import bench

    lp = bench.launch_plan
    assert lp(1, None, 0, False) == ("rank", 1)
    assert lp(1, None, 8, False) == ("rank", 1)
    assert lp(8, None, 8, False) == ("self_launch", 8)
    assert lp(2, None, 1, True) == ("self_launch", 2)            # one-GPU test mode
    assert lp(2, None, 1, False)[0] == "error"                   # two ranks need two GPUs
    assert lp(8, "8", 8, False) == ("rank", 8)                   # launched by torchrun: one of the ranks
    assert lp(1, "1", 1, False) == ("rank", 1)
    assert lp(8, "1", 8, False)[0] == "error" and lp(1, "8", 8, False)[0] == "error"
    assert lp(0, None, 8, False)[0] == "error"


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 2 and "WORLD_SIZE=4" in out.stderr and not out.stdout.strip()


def test_self_launch_relays_the_ranks_return_code():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["LC_BENCH_SHARE_GPU"] = "1"  # lets a box without two GPUs reach the launch; the ranks then stop at "needs an MI355X"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--regions", "1"],
                         capture_output=True, text=True, timeout=600, env=env)
    import torch

    if torch.cuda.is_available():  # on a GPU box this is the real two-rank run
        assert out.returncode == 0, out.stderr[-2000:]
    else:
        assert out.returncode != 0 and "needs an MI355X" in out.stderr, out.stderr[-2000:]
        assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]


def test_committed_counter_passes_cover_the_bench_shapes():
    """`roofline.traffic` and the dense blocks' `traffic` come from the newest committed profiles/<round>/pmc_traffic.json: every kernel
    the line looks up must be found there under the key bench.py forms (a renamed template instantiation once dropped the dense blocks'
    counters from the line without any CPU test noticing)."""
    import bench

    c, src = bench.static_counters("lc_pose_unit_kernel", 256, 64)
    assert c and c["bytes_per_launch"] > 0 and c["sq"]["SQ_ACTIVE_INST_VALU"] > 0 and src["file"].startswith("profiles/")
    for B, N in ((32, 1024), (32, 1849)):
        for key in bench.dense_counter_keys(B, N, True):
            c, src = bench.static_counters(key, B, N)
            assert c and c["bytes_per_launch"] > 0 and c["sq"]["SQ_WAVE_CYCLES"] > 0 and src["git_sha"], key
