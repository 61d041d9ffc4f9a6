"""The model's activations in one kernel each way (ibgs_amd/activations.py, csrc/activate.hip) against torch's own: exp, F.normalize, sigmoid of the
reference's GaussianModel (scene/gaussian_model.py:44-52) and autograd's backward through them -- including the rows F.normalize clamps -- and the
renderer with and without the fusion."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from ibgs_amd import activations, renderer, simple_scene, synthetic as syn

pytestmark = pytest.mark.gpu


def raw(P, seed, dev="cuda"):
    g = torch.Generator(device=dev).manual_seed(seed)
    s = torch.randn(P, 3, device=dev, generator=g) * 1.5 - 3.0
    r = torch.randn(P, 4, device=dev, generator=g)
    o = torch.randn(P, 1, device=dev, generator=g) * 3.0
    if P >= 8:          # rows the clamp of F.normalize decides, huge and tiny logits
        r[0] = 0.0; r[1] = 1e-20; r[2] = torch.tensor([1e-13, 0, 0, 0], device=dev); r[3] = 1e18
        o[0] = 40.0; o[1] = -40.0; o[2] = 100.0; o[3] = -100.0
        s[0] = -30.0; s[1] = 10.0
    return s, r, o


@pytest.mark.parametrize("P", [1, 63, 64, 65, 1000, 262145])
def test_forward_and_backward_match_torch(P):
    s, r, o = raw(P, P)
    a = [t.clone().requires_grad_(True) for t in (s, r, o)]
    b = [t.clone().requires_grad_(True) for t in (s, r, o)]
    fs, fr, fo = activations.fused_activations(*a)
    ts, tr, to = torch.exp(b[0]), F.normalize(b[1]), torch.sigmoid(b[2])
    # forward: the same operations in the same order; torch's own kernels may round an exp or a 4-term sum differently by an ulp
    for x, y, what in ((fs, ts, "exp"), (fr, tr, "normalize"), (fo, to, "sigmoid")):
        assert x.shape == y.shape
        assert torch.allclose(x, y, rtol=3e-7, atol=1e-37), (what, float((x - y).abs().max()))
    g = torch.Generator(device="cuda").manual_seed(987654)
    ws, wr, wo = (torch.randn(t.shape, device="cuda", generator=g) for t in (fs, fr, fo))
    ((fs * ws).sum() + (fr * wr).sum() + (fo * wo).sum()).backward()
    ((ts * ws).sum() + (tr * wr).sum() + (to * wo).sum()).backward()
    for x, y, what in zip(a, b, ("exp", "normalize", "sigmoid")):
        gx, gy = x.grad, y.grad
        assert torch.isfinite(gx).all() == torch.isfinite(gy).all(), what
        m = torch.isfinite(gy)
        # (the normalisation's gradient is a difference of two nearly equal terms for a gradient along the quaternion: absolute bar scaled by the incoming gradient / |x|)
        scale = float(gy[m].abs().max()) if m.any() else 1.0
        assert torch.allclose(gx[m], gy[m], rtol=2e-5, atol=2e-6 * scale), (what, float((gx[m] - gy[m]).abs().max()), scale)


def test_only_the_wanted_gradients_are_computed():
    s, r, o = raw(500, 3)
    s.requires_grad_(True); o.requires_grad_(True)          # the rotations stay frozen
    fs, fr, fo = activations.fused_activations(s, r, o)
    (fs.sum() + fr.sum() + fo.sum()).backward()
    assert s.grad is not None and o.grad is not None and r.grad is None
    assert torch.allclose(s.grad, torch.exp(s.detach()), rtol=3e-7)


def test_the_renderer_with_and_without_the_fusion():
    """renderer.render (geo pass, fused plane map) on a SimpleGaussians model: FUSED_ACTIVATIONS on against off -- same images, same parameter gradients to rounding."""
    W, H, P = 256, 160, 6000
    g = syn.make_gaussians(P, 5, sh_degree=2, max_coeffs=9, opacity="trained")
    rng = np.random.default_rng(0)
    g["normal"] = rng.normal(size=(P, 3)).astype(np.float32); g["offset"] = (0.01 * rng.normal(size=(P, 1))).astype(np.float32)
    cams = simple_scene.orbit_cameras(W, H, n_views=4, device="cuda", nearest=2)
    scene = simple_scene.SimpleScene(cams, images=torch.rand(4, 3, H, W, device="cuda"), device="cuda")
    pipe, args = simple_scene.default_pipe(), simple_scene.default_args()
    bg = torch.zeros(3, device="cuda")
    res = {}
    for on in (False, True):
        renderer.FUSED_ACTIVATIONS = on
        try:
            pc = simple_scene.SimpleGaussians(g, sh_degree=2, device="cuda")
            with torch.no_grad():
                scene.rendered_depth_list = renderer.render_depth_batch(cams, pc, scene, pipe, args, bg, True, 2, 4)
            out = renderer.render(cams[0], pc, scene, pipe, args, bg, True, 2, 4, render_geo=True)
            (out["render"].square().sum() + out["median_intersected_depth"].sum() * 0.01).backward()
            res[on] = (out["render"].detach().clone(), scene.rendered_depth_list.clone(), {n: getattr(pc, n).grad.clone() for n in ("_scaling", "_rotation", "_opacity", "_xyz")})
        finally:
            renderer.FUSED_ACTIVATIONS = True
    # an ulp in an activated value moves a pixel by ~1e-6; the lists themselves are the same (no Gaussian sits on a cull threshold in this scene)
    assert torch.allclose(res[True][0], res[False][0], atol=2e-5), float((res[True][0] - res[False][0]).abs().max())
    assert torch.allclose(res[True][1], res[False][1], atol=1e-4)
    for n in res[True][2]:
        a, b = res[True][2][n], res[False][2][n]
        assert float((a - b).norm() / (b.norm() + 1e-30)) < 1e-4, n
