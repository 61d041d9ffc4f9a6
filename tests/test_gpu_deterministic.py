"""IBGS_FLAG_DETERMINISTIC (rasterizer.DETERMINISTIC): the backward without float atomics -- per-(Gaussian, tile) sums go to a
slab and are added per Gaussian in list order.  Two runs must agree BIT FOR BIT in every gradient,
for the colour path and the geo path, with both wave shapes; against the default (atomic) mode and the oracle only the summation
order differs."""
import numpy as np
import pytest
import torch

import oracle
from ibgs_amd import rasterizer
from tests import hipref
from tests.metrics import rel_l2
from tests.scenes import add_sources, scene

pytestmark = pytest.mark.gpu
NAMES = ("means3D", "means2D", "means2D_abs", "shs", "opacities", "scales", "rotations", "all_map")


def _grads(inp, gr, det):
    old = rasterizer.DETERMINISTIC
    rasterizer.DETERMINISTIC = det
    try:
        outs, lv, _ = hipref.run_forward(inp)
        loss = 0
        for k, g in gr.items():
            loss = loss + (outs[k] * torch.as_tensor(g, device="cuda")).sum()
        loss.backward()
        torch.cuda.synchronize()
    finally:
        rasterizer.DETERMINISTIC = old
    return {k: lv[k].grad.clone() for k in NAMES if lv.get(k) is not None and lv[k].grad is not None}


@pytest.mark.parametrize("shape", ["tile", "quadrant"])
@pytest.mark.parametrize("geo", [False, True])
def test_two_runs_are_bit_identical_and_match_the_atomic_mode(geo, shape):
    old = rasterizer.WAVE_SHAPE
    rasterizer.WAVE_SHAPE = shape
    try:
        inp = scene(P=6000, W=256, H=176, deg=2, seed=33, opacity="trained", planes=geo, scale_mul=1.4)
        rnd = lambda s, k: np.random.default_rng(k).normal(size=s).astype(np.float32)
        H, W = inp["H"], inp["W"]
        gr = {"color": rnd((3, H, W), 1)}
        if geo:
            inp = add_sources(inp, n_src=3, L=4)
            gr.update(normal_map=rnd((3, H, W), 2), median_depth=rnd((1, H, W), 3), warped_image=rnd((15, H, W), 4))
        a, b = _grads(inp, gr, True), _grads(inp, gr, True)
        c = _grads(inp, gr, False)
        assert set(a) == set(b) == set(c) and len(a) >= 7
        for k in a:
            assert torch.equal(a[k], b[k]), "%s differs between two deterministic runs" % k
            assert float(a[k].abs().max()) > 0
            assert rel_l2(a[k].cpu().numpy(), c[k].cpu().numpy()) < 1e-5, k          # same sums, other order
        ref = oracle.forward(inp, cull=True)
        rb = oracle.backward(inp, ref, gr["color"], gr.get("normal_map"), gr.get("median_depth"), gr.get("warped_image"))
        for k, rk in (("means3D", "dL_dmeans3D"), ("opacities", "dL_dopacity"), ("shs", "dL_dsh"), ("scales", "dL_dscales")):
            assert rel_l2(a[k].cpu().numpy().reshape(rb[rk].shape), rb[rk]) < 1e-3, k
    finally:
        rasterizer.WAVE_SHAPE = old


def test_required_size_and_missing_scratch_are_reported():
    from ibgs_amd import _lib
    lib = _lib.load()
    assert lib.ibgs_required_deterministic(10**6, 10**5) > 10**6 * (64 + 16)
    assert lib.ibgs_required_deterministic(0, 0) > 0
