"""Parity where trained IBGS / PGSR scenes live: plane-like Gaussians (one axis 10^-2 .. 10^-3 of the others), needles, and needles so
long and thin that the reference's `power > 0` test decides what is blended (forward.cu:420, backward.cu:645).  Same checks and bars
as tests/test_gpu_parity.py (integer stages exact, colour mean L1 <= 1e-6, gradients relative L2 <= 1e-3), both wave shapes."""
import numpy as np
import pytest
import torch

import oracle
from ibgs_amd import rasterizer
from tests import hipref
from tests.metrics import l1, rel_l2
from tests.scenes import add_sources, giant_needles, scene
from tests.test_gpu_parity import (GEO_GRAD_TOL, canon_valid, check_color, check_grads, check_stages, rnd, run,  # noqa: F401
                                   wave_shape)

pytestmark = pytest.mark.gpu


def report(tag, o, ist, ref):
    d = np.abs(o["color"] - ref["color"])
    print("\n[aniso] %s: R %d, colour mean L1 %.2e max %.2e, n_contrib equal on %.4f %% of the pixels, oracle power>0 skips %d"
          % (tag, ref["num_rendered"], d.mean(), d.max(), 100.0 * (ist["n_contrib"] == ref["n_contrib"]).mean(), oracle.power_skips()[0]))


@pytest.mark.parametrize("anisotropy,opacity", [("plane", "trained"), ("plane", "init"), ("needle", "trained"), ("mixed", "trained")])
def test_colour_path_on_anisotropic_gaussians(anisotropy, opacity):
    inp = scene(P=4000, W=208, H=144, deg=3, seed=31, opacity=opacity, anisotropy=anisotropy)
    ref, o, ist, leaves, gb = run(inp, {"color": rnd((3, 144, 208), 1)})
    report("%s/%s" % (anisotropy, opacity), o, ist, ref)
    co = ref["conic_opacity"][ref["radii"] > 0]
    aspect2 = (co[:, 0] * co[:, 2]) / np.maximum(co[:, 0] * co[:, 2] - co[:, 1] ** 2, 1e-30)          # a c / det: grows with the 2D aspect ratio
    assert (aspect2 > 25.0).mean() > (0.002 if anisotropy == "plane" else 0.2), "the scene holds no strongly anisotropic footprints"
    check_stages(ist, o, ref); check_color(o, ist, ref); check_grads(leaves, gb)


def test_geo_path_on_plane_like_gaussians():
    inp = add_sources(scene(P=2500, W=176, H=112, deg=2, seed=33, opacity="trained", planes=True, scale_mul=1.5, anisotropy="plane"), n_src=3, L=4)
    H, W = inp["H"], inp["W"]
    grads = {"color": rnd((3, H, W), 7), "normal_map": rnd((3, H, W), 8), "median_depth": rnd((1, H, W), 9), "warped_image": rnd((15, H, W), 10)}
    ref, o, ist, leaves, gb = run(inp, grads)
    report("plane/geo", o, ist, ref)
    check_stages(ist, o, ref); check_color(o, ist, ref)
    assert (ref["valid_src_idx"][0] >= 0).mean() > 0.2, "scene does not exercise the warp path"
    assert np.array_equal(ist["low_high"][:, 0], ref["cache_low"]) and np.array_equal(ist["low_high"][:, 1], ref["cache_high"])
    same = np.all(canon_valid(ist["valid_idx"]) == canon_valid(ref["valid_src_idx"]), axis=0)
    assert same.mean() > 0.999
    assert l1(o["normal_map"], ref["normal_map"]) < 1e-6
    ok = same.reshape(H, W)
    for k, tol in (("median_depth", 1e-4), ("cam_feat", 1e-5), ("warped_image", 1e-5), ("min_depth_diff", 1e-5), ("camera_ray", 1e-5)):
        d = np.abs(o[k] - ref[k])[:, ok]
        assert d.mean() / (np.abs(ref[k][:, ok]).mean() + 1e-9) < tol, k
    check_grads(leaves, gb, tol=GEO_GRAD_TOL)


@pytest.mark.parametrize("anisotropy", ["plane", "needle"])
def test_tile_culling_changes_no_result_on_anisotropic_gaussians(anisotropy):
    """HIP with culling against the oracle WITHOUT culling (the reference's AABB lists): thin ellipses are where the cull removes most."""
    inp = scene(P=3000, W=192, H=128, deg=1, seed=41, opacity="trained", anisotropy=anisotropy)
    g = {"color": rnd((3, 128, 192), 7)}
    full = oracle.forward(inp, cull=False)
    gfull = oracle.backward(inp, full, g["color"])
    culled = oracle.forward(inp, cull=True)
    assert culled["num_rendered"] < 0.7 * full["num_rendered"]
    for k in ("color", "radii", "final_T"):
        assert np.array_equal(culled[k], full[k]), k                    # oracle vs oracle: bit-identical
    ref, o, ist, leaves, _ = run(inp, g, cull=True)
    assert ist["R"] == culled["num_rendered"]
    assert l1(o["color"], full["color"]) < 1e-6 and np.array_equal(o["radii"], full["radii"])
    check_grads(leaves, gfull)


@pytest.mark.parametrize("stretch,thin,fires", [(4.0, 0.05, True), (2.5, 0.03, True), (2.0, 0.05, False)])
def test_reference_power_skip_on_giant_needles(stretch, thin, fires):
    """Conics within rounding of singular: the oracle drops many pairs through `power > 0`.  The default HIP path (reference expression for
    those Gaussians) must agree with it like on any other scene; with IBGS_FLAG_NO_REF_POWER_SKIP the image is visibly different."""
    inp = giant_needles(stretch=stretch, thin=thin)
    H, W = inp["H"], inp["W"]
    ref, o, ist, leaves, gb = run(inp, {"color": rnd((3, H, W), 3)})
    nskip = oracle.power_skips()
    report("giant needles %g/%g" % (stretch, thin), o, ist, ref)
    co = ref["conic_opacity"][ref["radii"] > 0]
    assert (co[:, 1] ** 2 > np.float32(0.99999) * co[:, 0] * co[:, 2]).mean() > 0.4          # most Gaussians take the reference-expression branch
    check_stages(ist, o, ref); check_color(o, ist, ref); check_grads(leaves, gb)
    if not fires:          # near-singular conics, but no pair is dropped: the branch alone must not change anything
        assert nskip == (0, 0)
        return
    assert nskip[0] > 100 and nskip[1] > 100, nskip
    old = rasterizer.REF_POWER_SKIP
    try:
        rasterizer.REF_POWER_SKIP = False
        outs, _, _ = hipref.run_forward(inp, requires_grad=False)
    finally:
        rasterizer.REF_POWER_SKIP = old
    off = outs["color"].cpu().numpy()
    print("[aniso]    without the skip: colour mean L1 %.2e max %.2e" % (l1(off, ref["color"]), np.abs(off - ref["color"]).max()))
    assert l1(off, ref["color"]) > 10.0 * l1(o["color"], ref["color"])
    assert np.isfinite(off).all()


def test_geo_and_depth_only_passes_on_giant_needles():
    base = giant_needles(P=250, W=176, H=112, seed=9)
    from ibgs_amd import synthetic as syn
    base["all_map"] = syn.plane_all_map(base["means3D"], base["scales"], base["rotations"], base["_cam"])
    inp = add_sources(base, n_src=2, L=4)
    H, W = inp["H"], inp["W"]
    grads = {"color": rnd((3, H, W), 7), "normal_map": rnd((3, H, W), 8), "median_depth": rnd((1, H, W), 9), "warped_image": rnd((15, H, W), 10)}
    ref, o, ist, leaves, gb = run(inp, grads)
    assert oracle.power_skips()[0] > 100
    report("giant needles/geo", o, ist, ref)
    check_stages(ist, o, ref); check_color(o, ist, ref)
    assert l1(o["normal_map"], ref["normal_map"]) < 1e-6
    check_grads(leaves, gb, tol=GEO_GRAD_TOL)
    d = dict(base); d.update(render_depth_only=True, buffer_length=4)
    rd = oracle.forward(d)
    outs, _, _ = hipref.run_forward(d, requires_grad=False)
    dd = np.abs(outs["median_depth"].cpu().numpy() - rd["median_depth"])
    assert dd.mean() / (np.abs(rd["median_depth"]).mean() + 1e-9) < 1e-5
