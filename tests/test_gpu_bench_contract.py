"""bench.py prints exactly ONE JSON line on stdout with the fields the driver reads (metric / value / unit / n_gpus / steps /
warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config) plus the roofline object; run here on
the small C1 workload so that it takes seconds."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--geo"], ["--forward-only"]])
def test_bench_emits_one_contract_line(extra):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--config", "C1",
                        "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "fps" and d["value"] > 0 and abs(d["value"] - 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["scaling"] == "weak" and "workload" in d["config"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    # the blend kernels are VALU-issue bound: `bound` says so, achieved / peak / frac stay the HBM figures of SURVEY 8(d)
    assert rf["bound"] == "valu" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and 0 < rf["frac"] < 1
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    # profile-derived fields are either tagged with where they came from or absent -- never silently stale
    assert (rf["traffic"] is None) == (rf["traffic_source"] is None)
    assert rf["valu"] is None or "source" in rf["valu"]
    assert d["median_ms_hipevent"] > 0
    if not extra:
        g = d["geo"]              # second line of SURVEY 8(d) under the same clock
        assert g["ms_per_step"] > 0 and "render_geo" in g["workload"] and g["roofline"]["kernel"].startswith("render_bwd_geo")
    else:
        assert "geo" not in d


@pytest.mark.gpu
def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no launcher starts two ranks itself.  On this 1-GPU box: (a) with RCCL it must refuse
    loudly (exit code 2, nothing on stdout); (b) with the ranks allowed to share the device over gloo it must print n_gpus = 2
    and the exchange object."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--config", "C1", "--no-cpu-baseline"]
    if torch.cuda.device_count() < 2:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
        assert r.returncode == 2 and not r.stdout.strip() and "refusing" in r.stderr, (r.returncode, r.stdout, r.stderr[-500:])
    env.update(IBGS_BENCH_SHARE_GPU="1", IBGS_DIST_BACKEND="gloo")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "view-parallel x2" and d["scaling"] == "weak"
    x = d["rccl"]
    assert x["world"] == 2 and x["backend"] == "gloo" and x["exchange"] == "factored" and x["exchange_ms"] > 0 and x["bytes_per_rank"] > 0
    assert abs(d["value"] - 2 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]
    # the backward wrote the four dense gradients straight into the all-reduce bucket (no pack copy), and the host agreed every step
    assert x["grads_copied_into_bucket"] == 0 and x["agreements"] >= d["steps"] and x["agree_host_ms"] >= 0


@pytest.mark.gpu
def test_bench_eight_view_step_on_one_gpu():
    """BASELINE config C4's step shape -- 8 views per step, one per rank, gradients exchanged -- with the 8 ranks sharing this box's one GPU
    over gloo (small C1-sized scene): the factored exchange gathers 8 views' factors, every rank ends the step, one line with n_gpus = 8.
    (An 8-GPU RCCL run is the driver's; this checks the code path it will take.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(IBGS_BENCH_SHARE_GPU="1", IBGS_DIST_BACKEND="gloo")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--config", "C1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["parallelism"] == "view-parallel x8" and d["scaling"] == "weak"
    x = d["rccl"]
    assert x["world"] == 8 and x["exchange"] == "factored" and x["grads_copied_into_bucket"] == 0
    assert abs(d["value"] - 8 * 1000.0 / d["ms_per_step"]) < 1e-6 * d["value"]


def test_bench_refuses_missing_gpus_without_touching_them():
    """CPU container: zero devices -> `--gpus 2` exits with code 2 before any rank is started."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has the devices")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "IBGS_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=env)
    assert r.returncode == 2 and not r.stdout.strip() and "refusing" in r.stderr
