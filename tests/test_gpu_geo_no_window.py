"""The geo backward without depth / warp gradients (round 6): when neither dL/dmedian_depth nor dL/dwarped_image comes in -- a loss on `render` and `rendered_normal`
alone -- the library builds no window table (csrc/render_bwd.hip: geo_window_kernel is not launched) and the blend loop looks nothing up.  The gradients must be
those of the same call with zero-filled upstream gradients (the table then holds zeros), bit for bit under the deterministic backward."""
import numpy as np
import pytest
import torch

from ibgs_amd import _lib, rasterizer
from tests import hipref
from tests.scenes import add_sources, scene

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("W,H,P", [(160, 112, 2500), (1296, 848, 30000)])          # quadrant waves / one wave per tile (>= 4096 tiles)
@pytest.mark.parametrize("normal_too", [False, True])
def test_no_window_pass_equals_zero_filled_gradients(W, H, P, normal_too):
    inp = add_sources(scene(P=P, W=W, H=H, deg=2, seed=5, opacity="trained", planes=True), n_src=3, L=4)
    rng = np.random.default_rng(1)
    g = torch.as_tensor(rng.standard_normal((3, H, W)).astype(np.float32), device="cuda")
    gn = torch.as_tensor(rng.standard_normal((3, H, W)).astype(np.float32), device="cuda")
    old = rasterizer.DETERMINISTIC
    rasterizer.DETERMINISTIC = True
    try:
        res, stages = {}, {}
        for zeros in (False, True):
            outs, lv, _ = hipref.run_forward(inp)
            loss = (outs["color"] * g).sum()
            if normal_too:
                loss = loss + (outs["normal_map"] * gn).sum()
            if zeros:          # the same loss with the depth / warp outputs taking part at weight zero: zero-filled gradients reach the library
                loss = loss + (outs["median_depth"] * 0.0).sum() + (outs["warped_image"] * 0.0).sum()
            _lib.timing_enable(_lib.STAGES)
            loss.backward()
            torch.cuda.synchronize()
            stages[zeros] = {k: v[0] for k, v in _lib.timing_collect().items()}
            _lib.timing_enable([])
            res[zeros] = {k: lv[k].grad.detach().clone() for k in ("means3D", "shs", "opacities", "scales", "rotations", "all_map", "means2D", "means2D_abs")}
        assert stages[True].get("geo_window", 0.0) > 0.0 and stages[False].get("geo_window", 0.0) == 0.0, (stages[False], stages[True])
        for k in res[True]:
            assert res[True][k].abs().sum() > 0 or k == "means2D_abs" or (k == "all_map" and not normal_too), k          # (dL/dall_map is all zero when only the colour has a gradient)
            assert torch.equal(res[True][k], res[False][k]), "%s: the backward without the window pass differs from zero-filled gradients" % k
    finally:
        rasterizer.DETERMINISTIC = old
        _lib.timing_enable([])
