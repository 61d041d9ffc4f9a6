"""SURVEY 8(a) row G(vii) as one HIP kernel each way (`ibgs_depth_normal_forward / _backward`, ibgs_amd/depthnormal.py) against
  * the REFERENCE's own `normal_from_depth_image` (tests/golden/depth_normal.npz, produced by importing utils/graphics_utils.py in the build container),
  * the torch formulation of the same glue (`renderer.render_normal` + render()'s normalisation) evaluated in float64 -- values and gradients: the fused
    fp32 kernel may be at most twice as far from that as torch's own fp32 evaluation is (the normals are differences of neighbouring back-projected
    points: both fp32 evaluations carry ~1e-7 x focal length of cancellation noise),
  * `render()` end to end with the switch on and off."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from ibgs_amd import renderer
from ibgs_amd.depthnormal import depth_normal
from tests.metrics import rel_l2

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "depth_normal.npz")


def _cam(fx, fy, cx, cy):
    c = SimpleNamespace(Fx=fx, Fy=fy, Cx=cx, Cy=cy)
    c.get_calib_matrix_nerf = lambda scale=1.0: (torch.tensor([[fx / scale, 0, cx / scale], [0, fy / scale, cy / scale], [0, 0, 1]]).float(), torch.eye(4))
    return c


def _torch_glue(cam, depth):
    """render_normal + the normalisation of render() (gaussian_renderer/__init__.py:338-342), in the dtype of `depth`."""
    K, _ = cam.get_calib_matrix_nerf()
    n = renderer.normal_from_depth_image(depth, K.to(depth.dtype)).permute(2, 0, 1)
    return n / (torch.norm(n, dim=0, keepdim=True) + 1e-8)


def test_fused_depth_normal_equals_the_reference_function():
    g = np.load(GOLD)
    K = g["K"]
    cam = _cam(float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]))
    d = torch.as_tensor(g["depth"], device="cuda")
    got = depth_normal(cam, d).cpu().numpy()
    want = torch.as_tensor(g["normal"]).permute(2, 0, 1)          # the reference's (H, W, 3), already unit length (F.normalize)
    want = (want / (torch.norm(want, dim=0, keepdim=True) + 1e-8)).numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() < 2e-5 and (got[:, 0, :] == 0).all() and (got[:, :, -1] == 0).all()


@pytest.mark.parametrize("W,H,fx", [(97, 61, 120.0), (640, 360, 900.0), (1920, 1080, 2668.0)])
def test_values_and_gradient_against_the_float64_formulation(W, H, fx):
    gen = torch.Generator().manual_seed(W)
    # a smooth surface plus noise (real median-depth maps are piecewise smooth; pure noise would make every normal a coin toss)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    depth = 3.0 + 0.8 * torch.sin(3.0 * xx) * torch.cos(2.0 * yy) + 0.02 * torch.randn(H, W, generator=gen)
    cot = torch.randn(3, H, W, generator=gen)
    cam = _cam(fx, fx * 1.01, 0.5 * W, 0.5 * H)
    res = {}
    for name, fn, dt, dev in (("f64", _torch_glue, torch.float64, "cpu"), ("torch32", _torch_glue, torch.float32, "cuda"), ("fused", depth_normal, torch.float32, "cuda")):
        d = depth.to(dt).to(dev).requires_grad_(True)
        out = fn(cam, d)
        (out * cot.to(dt).to(dev)).sum().backward()
        res[name] = (out.detach().cpu().double().numpy(), d.grad.cpu().double().numpy())
    for i, what in ((0, "normal map"), (1, "dL/ddepth")):
        e_fused, e_torch = rel_l2(res["fused"][i], res["f64"][i]), rel_l2(res["torch32"][i], res["f64"][i])
        print("[depth normal] %dx%d %s: rel L2 vs float64 -- fused %.2e, torch fp32 %.2e" % (W, H, what, e_fused, e_torch))
        assert e_fused <= max(2.0 * e_torch, 1e-6), what
    assert (res["fused"][0][:, 0, :] == 0).all() and (res["fused"][0][:, -1, :] == 0).all()          # the padded border


def test_render_with_the_fused_depth_normal():
    from tests.test_gpu_renderer import _setup
    from ibgs_amd import simple_scene
    dev, g, pc, cams, scene = _setup()
    pipe, args = simple_scene.default_pipe(), simple_scene.default_args()
    bg = torch.tensor([0.1, 0.1, 0.2], device=dev)
    with torch.no_grad():
        for j in cams[0].nearest_id:
            scene.rendered_depth_list[j] = renderer.render_depth(cams[j], pc, scene, pipe, args, bg, True, 3, 4)
    H, W = cams[0].image_height, cams[0].image_width
    cot = torch.randn(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    out, grads = {}, {}
    old = renderer.FUSED_DEPTH_NORMAL
    try:
        for fused in (False, True):
            renderer.FUSED_DEPTH_NORMAL = fused
            for p in pc.parameters():
                p.grad = None
            o = renderer.render(cams[0], pc, scene, pipe, args, bg, True, 3, 4, render_geo=True, return_depth_normal=True)
            (o["median_intersected_depth_normal"] * cot).sum().backward()
            out[fused] = o["median_intersected_depth_normal"].detach().cpu().numpy()
            grads[fused] = {n: p.grad.detach().cpu().numpy().copy() for n, p in pc.named_parameters() if p.grad is not None}
    finally:
        renderer.FUSED_DEPTH_NORMAL = old
    assert out[True].shape == (3, H, W) and np.abs(out[True] - out[False]).mean() < 1e-5
    assert set(grads[True]) == set(grads[False])
    for n in grads[True]:
        if np.abs(grads[False][n]).max() > 0:
            assert rel_l2(grads[True][n], grads[False][n]) < 2e-3, (n, rel_l2(grads[True][n], grads[False][n]))
