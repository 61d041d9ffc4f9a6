"""The workgroup -> tile layouts of the blend kernels (csrc/common.h: round-robin, runs per XCD, tile blocks per XCD) hand out every
(tile, wave of the tile) exactly once on every grid: tests/csrc/test_tile_map.hip, built with hipcc and run on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_every_layout_visits_every_tile_once(tmp_path):
    exe = str(tmp_path / "ttm")
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-I", os.path.join(ROOT, "ibgs_amd", "csrc"), "-o", exe,
                        os.path.join(ROOT, "tests", "csrc", "test_tile_map.hip")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "tile map ok" in r.stdout, r.stdout + r.stderr
