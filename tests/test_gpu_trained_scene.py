"""Parity on the workload the reference's trainer spends most of its iterations on (train.py:289-292: render_geo on TRAINED Gaussians):
plane-like shapes (scene/gaussian_model.py:156-173), a heavy tail of sizes (densification, :580-604), an uneven image, trained opacities --
`synthetic.make_gaussians(anisotropy="plane", scale_sigma=1, cluster=0.3)`, the generator of bench.py's `trained_geo` line -- and on what that
scene brought with it: exact tile culling for rectangles of ANY size (row runs recomputed by the binning for rectangles beyond the 256-tile
mask) and the binning's wave-cooperative walk over the cells of a large rectangle.  Same checks and bars as tests/test_gpu_parity.py."""
import numpy as np
import pytest
import torch

import oracle
from ibgs_amd import rasterizer, synthetic as syn
from tests import hipref
from tests.metrics import l1
from tests.scenes import add_sources
from tests.test_gpu_parity import (GEO_GRAD_TOL, canon_valid, check_color, check_grads, check_stages, rnd, run,  # noqa: F401
                                   wave_shape)

pytestmark = pytest.mark.gpu


def trained_scene(P=4000, W=208, H=144, deg=3, seed=71, scale_mul=1.0, planes=False, cluster=0.3, sigma=1.0, anisotropy="plane"):
    inp = syn.make_scene(P, W, H, sh_degree=deg, seed=seed, opacity="trained", with_planes=planes, anisotropy=anisotropy, scale_sigma=sigma, cluster=cluster)
    if scale_mul != 1.0:
        inp["scales"] = (inp["scales"] * scale_mul).astype(np.float32)
        if planes:
            inp["all_map"] = syn.plane_all_map(inp["means3D"], inp["scales"], inp["rotations"], inp["_cam"])
    return inp


def rect_area(ref):
    r = ref["rect4"].astype(np.int64)
    return (r[:, 2] - r[:, 0]) * (r[:, 3] - r[:, 1])


def test_colour_path_on_the_trained_generator():
    inp = trained_scene(scale_mul=3.0)          # (at 208 x 144 the median radius of the C3-sized scene, ~14 px, needs the scales tripled)
    ref, o, ist, leaves, gb = run(inp, {"color": rnd((3, 144, 208), 1)})
    rad = ref["radii"][ref["radii"] > 0]
    assert np.percentile(rad, 99) > 8 * np.median(rad), "the scene has no heavy tail"
    check_stages(ist, o, ref); check_color(o, ist, ref); check_grads(leaves, gb)


def test_geo_path_on_the_trained_generator():
    inp = add_sources(trained_scene(P=2500, W=176, H=112, deg=2, seed=73, scale_mul=4.0, planes=True), n_src=3, L=4)
    H, W = inp["H"], inp["W"]
    grads = {"color": rnd((3, H, W), 7), "normal_map": rnd((3, H, W), 8), "median_depth": rnd((1, H, W), 9), "warped_image": rnd((15, H, W), 10)}
    ref, o, ist, leaves, gb = run(inp, grads)
    check_stages(ist, o, ref); check_color(o, ist, ref)
    assert (ref["valid_src_idx"][0] >= 0).mean() > 0.2, "scene does not exercise the warp path"
    assert np.array_equal(ist["low_high"][:, 0], ref["cache_low"]) and np.array_equal(ist["low_high"][:, 1], ref["cache_high"])
    same = np.all(canon_valid(ist["valid_idx"]) == canon_valid(ref["valid_src_idx"]), axis=0)
    assert (~same).sum() <= 2
    assert l1(o["normal_map"], ref["normal_map"]) < 1e-6
    ok = same.reshape(H, W)
    for k, tol in (("median_depth", 1e-4), ("cam_feat", 1e-5), ("warped_image", 1e-5), ("min_depth_diff", 1e-5), ("camera_ray", 1e-5)):
        d = np.abs(o[k] - ref[k])[:, ok]
        assert d.mean() / (np.abs(ref[k][:, ok]).mean() + 1e-9) < tol, k
    check_grads(leaves, gb, tol=GEO_GRAD_TOL)


@pytest.mark.parametrize("W,H,P,mul", [(640, 400, 3000, 6.0), (1280, 720, 1500, 10.0)])
def test_rectangles_of_any_size_are_culled(W, H, P, mul):
    """Rectangles of more than 256 tiles: preprocess counts their row runs (the whole wave walks a tall rectangle), the binning recomputes
    the runs per cell (the whole wave walks a large rectangle's cells).  Lists bit-identical to the oracle's; cull = False still gives
    the reference's AABB lists; no public output differs between the two."""
    inp = trained_scene(P=P, W=W, H=H, deg=1, seed=77, scale_mul=mul)
    g = {"color": rnd((3, H, W), 3)}
    full = oracle.forward(inp, cull=False)
    gfull = oracle.backward(inp, full, g["color"])
    ref, o, ist, leaves, gb = run(inp, g, cull=True)
    area = rect_area(ref)
    big = area > 256
    rows_mode = big & (ref["tmask"][:, 0] == 0)
    assert rows_mode.sum() > 100 and (ref["tiles_touched"][rows_mode] < area[rows_mode]).mean() > 0.7, (big.sum(), rows_mode.sum())
    assert ((area > 64) & (area <= 256)).sum() > 50
    r = ref["rect4"].astype(np.int64)
    assert ((r[:, 3] - r[:, 1])[(area > 0) & ~big] > 16).sum() > 3, "no tall masked rectangle (the wave-cooperative mask assembly)"
    assert ref["num_rendered"] < 0.75 * full["num_rendered"]
    check_stages(ist, o, ref); check_color(o, ist, ref); check_grads(leaves, gb)
    # ... and against the reference's lists
    assert np.array_equal(o["radii"], full["radii"]) and l1(o["color"], full["color"]) < 1e-6
    check_grads(leaves, gfull)
    ref2, o2, ist2, leaves2, gb2 = run(inp, g, cull=False)
    assert ist2["R"] == full["num_rendered"]
    check_stages(ist2, o2, ref2); check_color(o2, ist2, ref2); check_grads(leaves2, gb2)


def test_large_rectangles_in_the_geo_path():
    """Geo pass over a frame with rectangles beyond the mask (planes, needles and near-isotropic Gaussians mixed)."""
    inp = trained_scene(P=1200, W=640, H=400, deg=1, seed=79, scale_mul=7.0, planes=True, anisotropy="mixed")
    inp = add_sources(inp, n_src=2, L=4)
    H, W = inp["H"], inp["W"]
    grads = {"color": rnd((3, H, W), 7), "normal_map": rnd((3, H, W), 8), "median_depth": rnd((1, H, W), 9), "warped_image": rnd((15, H, W), 10)}
    ref = oracle.forward(inp, cull=True)
    area = rect_area(ref)
    assert ((area > 256) & (ref["tmask"][:, 0] == 0)).sum() > 50
    ref, o, ist, leaves, gb = run(inp, grads)
    check_stages(ist, o, ref)
    assert l1(o["color"], ref["color"]) <= 1e-5 and (ist["n_contrib"] != ref["n_contrib"]).mean() <= 2e-4
    assert l1(o["normal_map"], ref["normal_map"]) < 1e-5


def test_batched_depth_views_with_large_rectangles():
    """The row runs of a large rectangle are recomputed by the binning from the record's pixel position, which is the VIEW's own while the
    rectangle's rows are the stacked grid's (ibgs_forward_args.n_views): every view must still equal its single pass."""
    from ibgs_amd.rasterizer import rasterize_depth_batch
    W, H, P = 640, 400, 1500
    g = syn.make_gaussians(P, 81, sh_degree=0, opacity="trained", anisotropy="plane", scale_sigma=1.0, cluster=0.3)
    g["scales"] = (g["scales"] * 7.0).astype(np.float32)
    cams = [syn.make_camera(W, H, azimuth_deg=a) for a in (0.0, 20.0, -35.0)]
    dev = "cuda"
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    vms = torch.stack([t(c["viewmatrix"]) for c in cams]); pms = torch.stack([t(c["projmatrix"]) for c in cams])
    cps = torch.stack([t(c["campos"]) for c in cams])
    depths, radii = rasterize_depth_batch(t(g["means3D"]), t(g["opacities"]), t(g["scales"]), t(g["rotations"]), None, 1.0, vms, pms, cps,
                                          [c["tanfovx"] for c in cams], [c["tanfovy"] for c in cams], H, W, 4, plane_mode=2)
    nbig = 0
    for v, c in enumerate(cams):
        inp = dict(g); inp.update(W=W, H=H, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"], viewmatrix=c["viewmatrix"], projmatrix=c["projmatrix"], campos=c["campos"],
                                  bg=np.zeros(3, np.float32), sh_degree=0, scale_modifier=1.0, render_geo=False, render_depth_only=True, n_src=1, buffer_length=4,
                                  all_map=syn.plane_all_map(g["means3D"], g["scales"], g["rotations"], c))
        ref = oracle.forward(inp, cull=True)
        nbig += int(((rect_area(ref) > 256) & (ref["tmask"][:, 0] == 0)).sum())
        assert np.array_equal(radii[v].cpu().numpy(), ref["radii"])
        d = np.abs(depths[v].cpu().numpy() - ref["median_depth"])
        assert d.mean() / (np.abs(ref["median_depth"]).mean() + 1e-9) < 1e-5, v
    assert nbig > 100


@pytest.mark.parametrize("P,W,H,deg,seed", [(9000, 668, 604, 0, 582813), (9000, 1143, 524, 2, 166922)])
def test_row_runs_of_the_binning_equal_the_count(P, W, H, deg, seed):
    """Two scenes a random sweep found (tools/fuzz_parity.py 30 11 - big, cases 4 and 21): a row-culled rectangle of 1 386 tiles whose
    determinant A C - B B, formed with a fused multiply-add at the binning's call site, put one run one tile short of what preprocess had
    counted -- R one below the oracle's, 244 tile ranges shifted.  The determinant is formed in one uncontracted place now (common.h)."""
    from tests.scenes import scene
    inp = scene(P=P, W=W, H=H, deg=deg, seed=seed, opacity="init", scale_mul=7.5)
    ref = oracle.forward(inp, cull=True)
    big = (rect_area(ref) > 256) & (ref["tmask"][:, 0] == 0)
    assert big.sum() > 50
    outs, lv, _ = hipref.run_forward(inp, debug=True)
    ist = hipref.internal_state(outs, inp)
    o = hipref.to_np(outs)
    assert np.array_equal(ist["tiles"], ref["tiles_touched"]) and ist["R"] == ref["num_rendered"]
    assert np.array_equal(ist["ranges"], ref["ranges"]) and np.array_equal(ist["point_list"], ref["point_list"])
    assert l1(o["color"], ref["color"]) < 1e-6
