"""The example training step (backbone -> fused head -> Loss_fn -> backward -> optimiser) runs and learns on one GPU."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_example_train_step_runs():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_sparse_ddp.py"), "--steps", "6", "--batch", "4",
                          "--sparse-cnt", "16", "--width", "16"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("step")]
    assert len(lines) == 6 and "median step" in out.stdout
    assert all("nan" not in l.lower() for l in lines)
