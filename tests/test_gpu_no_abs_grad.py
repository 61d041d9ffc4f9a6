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
