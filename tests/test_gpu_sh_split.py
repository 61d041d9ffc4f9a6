"""SH coefficients as the model's two arrays (`_features_dc` (P, 1, 3), `_features_rest` (P, M - 1, 3): `ibgs_forward_args.shs_rest`, round 5) against their
torch.cat, the reference's `get_features` (scene/gaussian_model.py:140-143).  Pure data layout: the image, the records and -- under the deterministic
backward -- every gradient must be equal BIT FOR BIT (dL/dmeans3D, which alone reads the coefficient values in a separately compiled kernel: to 2e-6), at every degree, for M = 16 (the wave-cooperative paths; P not a multiple of 64: a partial last
block) and M = 9 (the reference's default sh_degree = 2: the plain paths), with Gaussians off screen and untouched ones in the mix."""
import numpy as np
import pytest
import torch

from ibgs_amd import rasterizer, renderer, simple_scene
from tests import hipref
from tests.scenes import scene

pytestmark = pytest.mark.gpu


def _run(inp, split, grad_seed=3):
    st = hipref.settings_from(inp, "cuda")
    lv = hipref.leaf_inputs(inp, "cuda")
    rast = rasterizer.GaussianRasterizer(st)
    kw = dict(means3D=lv["means3D"], means2D=lv["means2D"], means2D_abs=lv["means2D_abs"], opacities=lv["opacities"], scales=lv["scales"], rotations=lv["rotations"])
    if split:
        dc = lv["shs"].detach()[:, :1].contiguous().requires_grad_(True); rest = lv["shs"].detach()[:, 1:].contiguous().requires_grad_(True)
        outs = rast(shs=dc, shs_rest=rest, **kw)
    else:
        outs = rast(shs=lv["shs"], **kw)
    g = torch.randn(outs[0].shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(grad_seed))
    (outs[0] * g).sum().backward()
    grads = {k: v.grad.detach().clone() for k, v in lv.items() if v is not None and v.grad is not None}
    if split:
        grads["shs"] = torch.cat((dc.grad, rest.grad), dim=1)
    return [o.detach().clone() for o in outs], grads


@pytest.mark.parametrize("P,deg,Mc", [(5003, 3, 16), (4096, 2, 16), (3001, 1, 16), (2000, 0, 16), (3000, 2, 9), (777, 1, 4)])
def test_split_sh_is_bit_identical_to_the_concatenated_coefficients(P, deg, Mc):
    inp = scene(P=P, W=208, H=144, deg=deg, seed=70 + deg, opacity="trained")
    if Mc != 16:
        inp["shs"] = np.ascontiguousarray(inp["shs"][:, :Mc])
    assert inp["shs"].shape[1] == Mc
    old = rasterizer.DETERMINISTIC
    try:
        rasterizer.DETERMINISTIC = True          # no float atomics: gradients comparable bit for bit
        a_out, a_g = _run(inp, False)
        b_out, b_g = _run(inp, True)
    finally:
        rasterizer.DETERMINISTIC = old
    assert torch.equal(a_out[0], b_out[0]) and torch.equal(a_out[1], b_out[1])
    vis = a_out[1] > 0
    assert 0 < int(vis.sum()) < P, "the scene should hold Gaussians on and off screen"
    assert set(a_g) == set(b_g)
    for k in a_g:
        if k == "means3D":          # the one gradient that reads the coefficient VALUES (d colour / d direction); the split layout is its own kernel instantiation, whose
            assert torch.allclose(a_g[k], b_g[k], rtol=2e-6, atol=1e-9), k          # multiply-adds the compiler may contract differently: last-bit differences
        else:
            assert torch.equal(a_g[k], b_g[k]), k
    assert float(a_g["shs"].abs().sum()) > 0 and float(a_g["shs"][~vis].abs().sum()) == 0


def test_render_hands_the_model_s_two_arrays_over():
    from tests.test_gpu_renderer import _setup
    dev, g, pc, cams, scn = _setup()
    pipe, args = simple_scene.default_pipe(), simple_scene.default_args()
    bg = torch.tensor([0.1, 0.1, 0.2], device=dev)
    res = {}
    old = (renderer.SPLIT_SH, rasterizer.DETERMINISTIC)
    try:
        rasterizer.DETERMINISTIC = True
        for split in (False, True):
            renderer.SPLIT_SH = split
            for p in pc.parameters():
                p.grad = None
            o = renderer.render(cams[0], pc, scn, pipe, args, bg, True, 3, 4, render_geo=False, return_depth_normal=False)
            o["render"].sum().backward()
            res[split] = (o["render"].detach().clone(), pc._features_dc.grad.clone(), pc._features_rest.grad.clone())
    finally:
        renderer.SPLIT_SH, rasterizer.DETERMINISTIC = old
    for x, y in zip(res[False], res[True]):
        assert torch.equal(x, y)
    assert float(res[True][2].abs().sum()) > 0
