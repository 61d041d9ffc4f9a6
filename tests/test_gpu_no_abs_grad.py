"""IBGS_FLAG_NO_ABS_GRAD: when `means2D_abs` needs no gradient (after densify_until_iter, at test time: train.py:400-410 is the only reader of that
statistic) the colour blend skips the two |.| moments.  Every other gradient must be what it is with the statistic, and the oracle's."""
import numpy as np
import pytest
import torch

import oracle
from ibgs_amd import rasterizer
from tests import hipref
from tests.metrics import rel_l2
from tests.scenes import scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", ["tile", "quadrant"])
def test_gradients_without_the_abs_statistic(shape):
    inp = scene(P=4000, W=208, H=144, deg=3, seed=23, opacity="trained")
    g = np.random.default_rng(4).normal(size=(3, 144, 208)).astype(np.float32)
    ref = oracle.forward(inp, cull=True)
    gb = oracle.backward(inp, ref, g)
    old = rasterizer.WAVE_SHAPE
    res = {}
    try:
        rasterizer.WAVE_SHAPE = shape
        for want in (True, False):
            outs, lv, _ = hipref.run_forward(inp)
            lv["means2D_abs"].requires_grad_(want) if want else None
            if not want:          # a leaf that needs no gradient: rebuild the call with it detached
                st = hipref.settings_from(inp, "cuda")
                lv = hipref.leaf_inputs(inp, "cuda")
                lv["means2D_abs"] = torch.zeros_like(lv["means2D_abs"])          # requires_grad = False
                outs = dict(zip(["color"], rasterizer.GaussianRasterizer(st)(means3D=lv["means3D"], means2D=lv["means2D"], means2D_abs=lv["means2D_abs"],
                                                                               opacities=lv["opacities"], shs=lv["shs"], scales=lv["scales"], rotations=lv["rotations"])[:1]))
            (outs["color"] * torch.as_tensor(g, device="cuda")).sum().backward()
            res[want] = {k: (v.grad.cpu().numpy() if (v is not None and v.grad is not None) else None) for k, v in lv.items()}
    finally:
        rasterizer.WAVE_SHAPE = old
    assert res[True]["means2D_abs"] is not None and np.abs(res[True]["means2D_abs"]).sum() > 0
    assert res[False]["means2D_abs"] is None
    for k, rk in (("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("shs", "dL_dsh"), ("opacities", "dL_dopacity"), ("scales", "dL_dscales"), ("rotations", "dL_drotations")):
        assert rel_l2(res[False][k], res[True][k]) < 2e-6, k                       # the same sums (float atomics: order noise only)
        assert rel_l2(res[False][k], gb[rk].reshape(res[False][k].shape)) < 1e-3, k


def test_geo_backward_without_the_abs_statistic():
    """The same flag on the geo path (one wave per tile): every gradient but the statistic unchanged."""
    from tests.scenes import add_sources
    inp = add_sources(scene(P=2500, W=176, H=112, deg=2, seed=33, opacity="trained", planes=True, scale_mul=1.5), n_src=3, L=4)
    H, W = inp["H"], inp["W"]
    r = np.random.default_rng(6)
    g = {"color": r.normal(size=(3, H, W)).astype(np.float32), "normal_map": r.normal(size=(3, H, W)).astype(np.float32),
         "median_depth": r.normal(size=(1, H, W)).astype(np.float32), "warped_image": r.normal(size=(15, H, W)).astype(np.float32)}
    old = rasterizer.WAVE_SHAPE
    res = {}
    try:
        rasterizer.WAVE_SHAPE = "tile"
        for want in (True, False):
            st = hipref.settings_from(inp, "cuda")
            lv = hipref.leaf_inputs(inp, "cuda")
            if not want:
                lv["means2D_abs"] = torch.zeros_like(lv["means2D_abs"])
            outs = rasterizer.GaussianRasterizer(st)(means3D=lv["means3D"], means2D=lv["means2D"], means2D_abs=lv["means2D_abs"], opacities=lv["opacities"],
                                                     shs=lv["shs"], scales=lv["scales"], rotations=lv["rotations"], all_map=lv["all_map"])
            loss = (outs[0] * torch.as_tensor(g["color"], device="cuda")).sum() + (outs[2] * torch.as_tensor(g["normal_map"], device="cuda")).sum() \
                + (outs[3] * torch.as_tensor(g["median_depth"], device="cuda")).sum() + (outs[5] * torch.as_tensor(g["warped_image"], device="cuda")).sum()
            loss.backward()
            res[want] = {k: (v.grad.cpu().numpy() if (v is not None and v.grad is not None) else None) for k, v in lv.items()}
    finally:
        rasterizer.WAVE_SHAPE = old
    assert res[False]["means2D_abs"] is None and np.abs(res[True]["means2D_abs"]).sum() > 0
    for k in ("means3D", "means2D", "shs", "opacities", "scales", "rotations", "all_map"):
        assert rel_l2(res[False][k], res[True][k]) < 2e-6, k
