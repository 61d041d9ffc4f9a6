"""examples/train_synthetic.py end to end on the GPU: renderer (fused plane glue, geo after warm-up) + FusedAdam drive the L1 loss
down on a synthetic multi-view scene."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_example_training_reduces_the_loss():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_synthetic.py"), "--iters", "60", "--points", "6000", "--width", "192",
                        "--height", "128", "--geo-from", "30", "--quiet"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    m = re.search(r"loss ([0-9.]+) -> ([0-9.]+)", r.stdout)
    assert m, r.stdout
    first, last = float(m.group(1)), float(m.group(2))
    assert last < 0.75 * first, r.stdout
