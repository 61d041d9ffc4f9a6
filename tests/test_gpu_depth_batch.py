"""batched depth-only passes (ibgs_forward_args.n_views).  Every view of the batch equals its own
single-view render_depth pass bit for bit -- also with a ragged image height (last tile row partly outside), with
per-view fields of view, with both normal modes and with a 1-slot buffer (tile culling off, position dependent)."""
import math

import numpy as np
import pytest
import torch

from ibgs_amd import renderer, simple_scene, synthetic as syn

pytestmark = pytest.mark.gpu


def _setup(P, W, H, n_views, seed):
    dev = torch.device("cuda")
    g = syn.make_gaussians(P, seed, sh_degree=1, max_coeffs=4, opacity="trained")
    g["scales"] = (g["scales"] * 1.5).astype(np.float32)
    rng = np.random.default_rng(seed)
    g["normal"] = rng.normal(size=(P, 3)).astype(np.float32); g["offset"] = (0.03 * rng.normal(size=(P, 1))).astype(np.float32)
    pc = simple_scene.SimpleGaussians(g, sh_degree=1, device=dev)
    pc._gnp = g                                       # the numpy originals, for the oracle comparison
    cams = simple_scene.orbit_cameras(W, H, n_views=n_views, device=dev, nearest=3)
    scene = simple_scene.SimpleScene(cams, device=dev)
    return dev, pc, cams, scene, simple_scene.default_pipe(), simple_scene.default_args(), torch.zeros(3, device=dev)


@pytest.mark.parametrize("W,H,n,learnt,L", [(160, 112, 4, True, 4), (200, 100, 3, False, 4), (96, 70, 5, True, 1), (64, 64, 8, False, 5), (128, 96, 1, True, 4)])
def test_batch_equals_single_passes(W, H, n, learnt, L):
    dev, pc, cams, scene, pipe, args, bg = _setup(3000, W, H, max(n, 3), seed=W + n)
    views = cams[:n]
    if n >= 3:                                        # per-view intrinsics: give one camera a different field of view
        views[1].FoVx *= 0.8; views[1].FoVy *= 0.8     # (tanfov feeds the focal lengths; both paths read the same camera object)
    with torch.no_grad():
        singles = torch.stack([renderer.render_depth(c, pc, scene, pipe, args, bg, learnt, 3, L) for c in views])
        batch = renderer.render_depth_batch(views, pc, scene, pipe, args, bg, learnt, 3, L)
    assert batch.shape == (n, 1, H, W) and batch.dtype == torch.float32
    assert float(singles.abs().max()) > 0
    assert torch.equal(batch, singles)
    # ... and every view against the ORACLE's depth-only pass of that camera (numpy plane map, reference AABB lists): the
    # batch is not only self-consistent.  L = 1 runs unculled lists on both sides (the per-256-round break is position dependent).
    import oracle
    g = pc._gnp
    for v, cam in enumerate(views):
        camd = {"viewmatrix": cam.world_view_transform.cpu().numpy(), "campos": cam.camera_center.cpu().numpy()}
        am = syn.plane_all_map(g["means3D"], g["scales"], g["rotations"], camd, normal=g["normal"] if learnt else None,
                               offset=g["offset"] if learnt else None)
        inp = {"means3D": g["means3D"], "shs": g["shs"], "opacities": g["opacities"], "scales": g["scales"], "rotations": g["rotations"],
               "all_map": am, "W": W, "H": H, "tanfovx": math.tan(cam.FoVx * 0.5), "tanfovy": math.tan(cam.FoVy * 0.5),
               "viewmatrix": camd["viewmatrix"], "projmatrix": cam.full_proj_transform.cpu().numpy(), "campos": camd["campos"],
               "bg": np.zeros(3, np.float32), "sh_degree": 1, "render_depth_only": True, "buffer_length": L}
        ref = oracle.forward(inp)["median_depth"]
        d = np.abs(batch[v].cpu().numpy() - ref)
        assert d.mean() / (np.abs(ref).mean() + 1e-9) < 1e-4, (v, d.mean())
        assert (d > 1e-3 * (1 + np.abs(ref))).mean() < 2e-3, v


def test_render_uses_the_batch_for_its_sources_and_matches_the_loop():
    dev, pc, cams, scene, pipe, args, bg = _setup(2500, 160, 112, 6, seed=5)
    scene.original_image_list = torch.rand(6, 3, 112, 160, device=dev)
    args.multi_view_max_angle = 90; args.multi_view_max_dis = 10.0
    with torch.no_grad():
        a = renderer.render(cams[1], pc, scene, pipe, args, bg, True, 3, 4, render_geo=True, do_find_closest_frame=True, do_render_src_depth=True)
        old = renderer.FUSED_PLANE_MAP
        renderer.FUSED_PLANE_MAP = False              # forces the per-view loop (and the torch glue)
        try:
            b = renderer.render(cams[1], pc, scene, pipe, args, bg, True, 3, 4, render_geo=True, do_find_closest_frame=True, do_render_src_depth=True)
        finally:
            renderer.FUSED_PLANE_MAP = old
    assert torch.allclose(a["render"], b["render"], atol=1e-6)
    d = (a["median_intersected_depth"] - b["median_intersected_depth"]).abs().mean() / b["median_intersected_depth"].abs().mean()
    assert float(d) < 1e-3


def test_batch_edge_cases_empty_and_everything_culled():
    dev, pc, cams, scene, pipe, args, bg = _setup(500, 80, 48, 3, seed=1)
    from ibgs_amd.rasterizer import rasterize_depth_batch
    vms = torch.stack([c.world_view_transform for c in cams]); pms = torch.stack([c.full_proj_transform for c in cams])
    cps = torch.stack([c.camera_center for c in cams])
    tx = [math.tan(c.FoVx * 0.5) for c in cams]; ty = [math.tan(c.FoVy * 0.5) for c in cams]
    # no Gaussians at all
    e = torch.zeros(0, 3, device=dev)
    d, r = rasterize_depth_batch(e, torch.zeros(0, 1, device=dev), e, torch.zeros(0, 4, device=dev), None, 1.0, vms, pms, cps, tx, ty, 48, 80, 4,
                                 plane_mode=2)
    assert d.shape == (3, 1, 48, 80) and not d.any() and r.shape == (3, 0)
    # every Gaussian behind every camera: R = 0, twice (the second call runs with a rendered_hint from ... nothing)
    far = (pc.get_xyz.detach() * 0.01 + torch.tensor([0.0, 0.0, 500.0], device=dev))
    for _ in range(2):
        d, r = rasterize_depth_batch(far, pc.get_opacity.detach(), pc.get_scaling.detach(), pc.get_rotation.detach(), None, 1.0, vms, pms, cps,
                                     tx, ty, 48, 80, 4, plane_mode=2)
        assert not r.any() and not d.any()
