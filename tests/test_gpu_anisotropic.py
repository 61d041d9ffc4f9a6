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

# THE ARBITER (round 4): `oracle.variant("f64")`, the same C source with every float a double.  A float evaluation -- the oracle proper, its
# fma twin, the HIP path -- is as good as its distance from that one: on anisotropic scenes every float bar is
#     |HIP - f64| <= max(the usual bar, F64_K x |oracle fp32 - f64|)
# per quantity, i.e. the HIP path may be at most F64_K times as far from the exact evaluation of the reference's algorithm as the reference's
# own precision is.  (Until round 3 the floor was the difference between the oracle's two fp32 builds, which says how far two roundings are
# from each other and nothing about which is right.)  The fma twin stays for the DECISIONS (n_contrib: which pairs are blended).
F64_K = 2.0

# What the reference's own arithmetic leaves undetermined.  nvcc contracts a*b+c into fma by default (-fmad=true) in a pattern that cannot
# be known here; the oracle is built without any contraction.  `oracle.variant("fma")` is the SAME C source with gcc free to contract: on
# near-isotropic scenes the two builds agree to 1e-6, on needle scenes they differ from each other by 1e-5 in the image and by 2e-3 ..
# 8e-3 in dL/dscales, dL/drotations, dL/dmeans3D (those pass through the inversion of a near-singular cov2D, backward.cu:241-371, which
# amplifies 1e-7 rounding by 1 / (1 - b^2 / (a c))), and on the giant needles they are different images.  A bar tighter than that
# self-difference would test the oracle's build flags, not the HIP path: on anisotropic scenes every float bar is
# max(the usual bar, 3 x the oracle's own fma / no-fma difference), and the north-star bars (image L1 1e-4) still hold absolutely.


class fixed_summation_order:
    """On needle scenes the per-Gaussian gradients are sums of tile contributions a thousand times larger than their total: with float atomics the ORDER of
    the additions moves dL/dmeans3D by 3e-3 of its norm from run to run (25 runs: 3.7e-3 .. 6.5e-3 from the float64 build, the fp32 oracle 3.6e-3) -- the
    reference's own atomicAdd backward has the same freedom.  A bar that is to mean something cannot sit inside that scatter: those cases run the
    deterministic backward (IBGS_FLAG_DETERMINISTIC: one fixed order), the default mode keeps its coverage on every other scene."""

    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        self.old = rasterizer.DETERMINISTIC
        rasterizer.DETERMINISTIC = self.on or self.old

    def __exit__(self, *exc):
        rasterizer.DETERMINISTIC = self.old
        return False


def fma_twin(inp, grads):
    with oracle.variant("fma"):
        r = oracle.forward(inp, tex_quant=rasterizer.TEX_QUANT, cull=True)
        g = oracle.backward(inp, r, grads["color"], grads.get("normal_map"), grads.get("median_depth"), grads.get("warped_image"),
                            tex_quant=rasterizer.TEX_QUANT)
    return r, g


def f64_truth(inp, grads):
    with oracle.variant("f64"):
        r = oracle.forward(inp, tex_quant=rasterizer.TEX_QUANT, cull=True)
        g = oracle.backward(inp, r, grads["color"], grads.get("normal_map"), grads.get("median_depth"), grads.get("warped_image"),
                            tex_quant=rasterizer.TEX_QUANT)
    return r, g


def check_color_aniso(o, ist, ref, twin, r64=None):
    floor = l1(twin["color"], ref["color"])
    d = l1(o["color"], ref["color"])
    assert d <= 1e-4 and d <= max(1e-6, 3.0 * floor), "colour mean L1 %.3e (oracle fma / no-fma: %.3e)" % (d, floor)
    if r64 is not None:
        d64, f64 = l1(o["color"], r64["color"]), l1(ref["color"], r64["color"])
        print("[aniso]    colour mean L1 vs the float64 build: HIP %.2e, oracle fp32 %.2e" % (d64, f64))
        assert d64 <= max(1e-6, F64_K * f64), "colour mean L1 vs float64 %.3e (oracle fp32 vs float64: %.3e)" % (d64, f64)
    bad = (ist["n_contrib"] != ref["n_contrib"]).mean()
    assert bad <= max(2e-4, 3.0 * (twin["n_contrib"] != ref["n_contrib"]).mean()), "n_contrib differs on %.4f %% of the pixels" % (100 * bad)
    tgt = np.random.default_rng(0).uniform(0, 1, ref["color"].shape)
    from tests.metrics import psnr
    assert abs(psnr(o["color"], tgt)[0] - psnr(ref["color"], tgt)[0]) <= 0.05


def check_grads_aniso(leaves, gb, g64, base_tol=1e-3, only=None):
    """Every gradient against the float64 build: the HIP path at most F64_K times as far from it as the fp32 oracle (`gb`) is."""
    from tests.test_gpu_parity import GRAD_PAIRS
    worst = {}
    for lk, rk in GRAD_PAIRS:
        if leaves.get(lk) is None or (only is not None and lk not in only):
            continue
        a = leaves[lk].grad.cpu().numpy(); t = np.asarray(g64[rk]).reshape(a.shape); b = gb[rk].reshape(a.shape)
        if np.abs(t).max() == 0:
            assert np.abs(a).max() == 0, lk
            continue
        floor = rel_l2(b, t)
        e = rel_l2(a, t)
        worst[lk] = (e, floor)
        assert e <= max(base_tol, F64_K * floor), "%s relL2 vs float64 %.3e (oracle fp32 vs float64: %.3e)" % (lk, e, floor)
    print("[aniso]    grads relL2 vs the float64 build (HIP | oracle fp32): " + ", ".join("%s %.1e|%.1e" % (k, v[0], v[1]) for k, v in worst.items()))


def report(tag, o, ist, ref):
    d = np.abs(o["color"] - ref["color"])
    print("\n[aniso] %s: R %d, colour mean L1 %.2e max %.2e, n_contrib equal on %.4f %% of the pixels, oracle power>0 skips %d"
          % (tag, ref["num_rendered"], d.mean(), d.max(), 100.0 * (ist["n_contrib"] == ref["n_contrib"]).mean(), oracle.power_skips()[0]))


@pytest.mark.parametrize("anisotropy,opacity", [("plane", "trained"), ("plane", "init"), ("needle", "trained"), ("mixed", "trained")])
def test_colour_path_on_anisotropic_gaussians(anisotropy, opacity):
    inp = scene(P=4000, W=208, H=144, deg=3, seed=31, opacity=opacity, anisotropy=anisotropy)
    g = {"color": rnd((3, 144, 208), 1)}
    with fixed_summation_order(anisotropy != "plane"):
        ref, o, ist, leaves, gb = run(inp, g)
    twin, _ = fma_twin(inp, g)
    r64, g64 = f64_truth(inp, g)
    report("%s/%s" % (anisotropy, opacity), o, ist, ref)
    co = ref["conic_opacity"][ref["radii"] > 0]
    aspect2 = (co[:, 0] * co[:, 2]) / np.maximum(co[:, 0] * co[:, 2] - co[:, 1] ** 2, 1e-30)          # a c / det: grows with the 2D aspect ratio
    assert (aspect2 > 25.0).mean() > (0.002 if anisotropy == "plane" else 0.2), "the scene holds no strongly anisotropic footprints"
    check_stages(ist, o, ref); check_color_aniso(o, ist, ref, twin, r64); check_grads_aniso(leaves, gb, g64)
    if anisotropy == "plane":          # the shape real scenes have: the ordinary bars hold as they are
        check_color(o, ist, ref); check_grads(leaves, gb)


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
    assert (~same).sum() <= 2
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
    with fixed_summation_order(anisotropy != "plane"):
        ref, o, ist, leaves, _ = run(inp, g, cull=True)
    _, g64 = f64_truth(inp, g)
    assert ist["R"] == culled["num_rendered"]
    assert l1(o["color"], full["color"]) < (1e-6 if anisotropy == "plane" else 1e-5) and np.array_equal(o["radii"], full["radii"])
    check_grads_aniso(leaves, gfull, g64)


@pytest.mark.parametrize("stretch,thin,fires", [(4.0, 0.05, True), (2.5, 0.03, True), (2.0, 0.05, False)])
def test_reference_power_skip_on_giant_needles(stretch, thin, fires):
    """Conics within rounding of singular: the oracle drops many pairs through `power > 0`.  The default HIP path (reference expression for
    those Gaussians) must agree with it like on any other scene; with IBGS_FLAG_NO_REF_POWER_SKIP the image is visibly different."""
    inp = giant_needles(stretch=stretch, thin=thin)
    H, W = inp["H"], inp["W"]
    g = {"color": rnd((3, H, W), 3)}
    ref, o, ist, leaves, gb = run(inp, g)
    nskip = oracle.power_skips()
    report("giant needles %g/%g" % (stretch, thin), o, ist, ref)
    co = ref["conic_opacity"][ref["radii"] > 0]
    assert (co[:, 1] ** 2 > np.float32(0.99999) * co[:, 0] * co[:, 2]).mean() > 0.4          # most Gaussians take the reference-expression branch
    # the image and the sums over pixels agree like on any scene (the oracle's fma twin is a different image here); the gradients
    # behind the inversion of a singular cov2D are noise in the reference itself (twin: relative differences of order one)
    check_stages(ist, o, ref)
    assert l1(o["color"], ref["color"]) <= 1e-5 and (ist["n_contrib"] != ref["n_contrib"]).mean() <= 2e-4
    assert l1(ist["final_T"], ref["final_T"]) < 1e-5
    # (no float64 arbiter here: with conics within rounding of singular the float64 build blends other pairs -- it is a different image, like the
    # fma twin; the fp32 oracle IS the target, and the sums over pixels must meet the ordinary bar)
    check_grads(leaves, gb, skip=("means3D", "scales", "rotations", "cov3D_precomp"))
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
    check_stages(ist, o, ref)
    assert l1(o["color"], ref["color"]) <= 1e-5 and (ist["n_contrib"] != ref["n_contrib"]).mean() <= 2e-4
    assert l1(o["normal_map"], ref["normal_map"]) < 1e-5
    check_grads(leaves, gb, tol=GEO_GRAD_TOL, skip=("means3D", "scales", "rotations", "cov3D_precomp"))
    d = dict(base); d.update(render_depth_only=True, buffer_length=4)
    rd = oracle.forward(d)
    outs, _, _ = hipref.run_forward(d, requires_grad=False)
    dd = np.abs(outs["median_depth"].cpu().numpy() - rd["median_depth"])
    assert dd.mean() / (np.abs(rd["median_depth"]).mean() + 1e-9) < 1e-5
