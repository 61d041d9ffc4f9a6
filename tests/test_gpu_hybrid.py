"""The hybrid colour kernels (frames of 768 ... 4 095 tiles, no wave shape forced): per tile either one wave or four quadrant waves,
chosen on the device -- by the backward from how far the forward walked the tile's list (bit 31 of the tile's word in the launch order), by the forward
from that flag when the caller hands it the order the same camera's last backward left (rasterizer.ORDER_HINT); without one it splits every tile.  Any mixture must give the oracle's image and gradients; the deterministic mode's slab holds four rows per list entry whichever
shape walked it."""
import numpy as np
import pytest
import torch

import oracle
from ibgs_amd import _lib, rasterizer, synthetic as syn
from tests import hipref
from tests.metrics import rel_l2
from tests.scenes import add_sources
from tests.test_gpu_parity import GEO_GRAD_TOL, GRAD_PAIRS, check_color, check_grads, rnd

pytestmark = pytest.mark.gpu

SPLIT = 0x80000000


def uneven_scene(W=960, H=544, P=30000, seed=91, opacity="init"):
    """2 040 tiles (a SIMD's fair share is two tiles' worth of an even frame), half of the Gaussians in one blob: a few tiles hold most of the
    work, most tiles little."""
    inp = syn.make_scene(P, W, H, sh_degree=1, seed=seed, opacity=opacity, cluster=0.5)
    inp["scales"] = (inp["scales"] * 3.0).astype(np.float32)
    return inp


def arena_words(outs, inp, name, count):
    lib = _lib.load()
    img = outs["color"].grad_fn.saved_tensors[-1].cpu().numpy()
    off = lib.ibgs_img_offset(int(inp["W"]), int(inp["H"]), name.encode())
    return np.frombuffer(img.tobytes()[off:off + 4 * count], dtype=np.uint32).copy()


NAMES = ["color", "radii", "normal_map", "median_depth", "cam_feat", "warped_image", "min_depth_diff", "camera_ray", "use_first_src_frame_mask"]


def step(inp, g, st):
    """forward + backward with ONE settings object per scene: the launch order hint is kept per camera, and a camera is its view matrix tensor"""
    lv = hipref.leaf_inputs(inp, "cuda", True)
    outs = dict(zip(NAMES, rasterizer.GaussianRasterizer(st)(means3D=lv["means3D"], means2D=lv["means2D"], means2D_abs=lv["means2D_abs"], opacities=lv["opacities"],
                                                            shs=lv["shs"], colors_precomp=lv["colors_precomp"], scales=lv["scales"], rotations=lv["rotations"],
                                                            cov3D_precomp=lv["cov3D_precomp"], all_map=lv["all_map"])))
    leaves = lv
    ist = hipref.internal_state(outs, inp)
    nt = ((inp["W"] + 15) // 16) * ((inp["H"] + 15) // 16)
    meta = arena_words(outs, inp, "meta", 32)
    walked = arena_words(outs, inp, "tile_walked", nt * 4).reshape(nt, 4)
    (outs["color"] * torch.as_tensor(g, device="cuda")).sum().backward(retain_graph=True)
    torch.cuda.synchronize()
    nslots = int(_lib.load().ibgs_tile_order_slots(int(inp["W"]), int(inp["H"])))
    order = arena_words(outs, inp, "tile_order", nslots)
    return hipref.to_np(outs), ist, leaves, meta, walked, order


@pytest.mark.parametrize("opacity", ["init", "trained"])
def test_mixture_of_tile_and_quadrant_waves(opacity):
    assert rasterizer.WAVE_SHAPE is None and rasterizer.ORDER_HINT
    inp = uneven_scene(opacity=opacity)
    nt = ((inp["W"] + 15) // 16) * ((inp["H"] + 15) // 16)
    g = rnd((3, inp["H"], inp["W"]), 5)
    ref = oracle.forward(inp, cull=True)
    gb = oracle.backward(inp, ref, g)
    rasterizer._order_hints.clear()
    st = hipref.settings_from(inp, "cuda", False)
    # first call of this camera: no measurement to go by -- every tile is walked by four quadrant waves (what the library did before the hybrid kernels)
    o, ist, leaves, meta, walked, order = step(inp, g, st)
    assert meta[10] == 4 and meta[11] == 0
    n = ist["ranges"][:, 1].astype(np.int64) - ist["ranges"][:, 0]
    split = (walked[:, 1:] > 0).any(axis=1)          # (a tile wave leaves words 1..3 at zero; a split tile's quadrants all see something in this scene)
    assert split[n > 64].all(), "a tile of the first call was walked by one wave"
    check_color(o, ist, ref); check_grads(leaves, gb)
    # the backward left its order: every tile once, heavy ones flagged
    tiles = order[order != 0xFFFFFFFF]
    assert np.array_equal(np.sort(tiles & ~np.uint32(SPLIT)), np.arange(nt, dtype=np.uint32))
    flagged = np.zeros(nt, bool); flagged[(tiles[(tiles & SPLIT) != 0] & ~np.uint32(SPLIT)).astype(np.int64)] = True
    assert 4 < flagged.sum() < nt - nt // 4
    wmax = walked.max(axis=1).astype(np.int64)
    assert wmax[flagged].min() >= wmax[~flagged].max(), "the flag follows the walked length"
    # second call of the same camera: the forward follows the backward's flags
    o2, ist2, leaves2, meta2, walked2, order2 = step(inp, g, st)
    assert meta2[11] == 1, "the hint was not accepted"
    split2 = (walked2[:, 1:] > 0).any(axis=1)
    assert not split2[~flagged].any() and split2[flagged].all(), "the forward's shapes are not the backward's flags"
    assert (~split2 & (n > 64)).sum() > nt // 4, "the second call walked (nearly) every tile with four waves: no mixture"
    check_color(o2, ist2, ref); check_grads(leaves2, gb)
    assert np.array_equal(np.sort(order2 & ~np.uint32(SPLIT)), np.sort(order & ~np.uint32(SPLIT)))


def test_deterministic_backward_under_a_mixture():
    inp = uneven_scene(seed=93)
    g = rnd((3, inp["H"], inp["W"]), 6)
    rasterizer._order_hints.clear()
    st = hipref.settings_from(inp, "cuda", False)
    _, _, atomic, *_ = step(inp, g, st)
    old = rasterizer.DETERMINISTIC
    try:
        rasterizer.DETERMINISTIC = True
        runs = [step(inp, g, st)[2] for _ in range(3)]
    finally:
        rasterizer.DETERMINISTIC = old
    for lk, _ in GRAD_PAIRS:
        if atomic.get(lk) is None or atomic[lk].grad is None:
            continue
        a = runs[1][lk].grad
        assert torch.equal(a, runs[2][lk].grad), lk          # same mixture of shapes (the previous backward's flags), no atomics: the same bits
        b = atomic[lk].grad
        if float(b.abs().max()) > 0:
            assert rel_l2(a.cpu().numpy(), b.cpu().numpy()) < 1e-4, lk          # (atomic order, and another mixture of shapes: the first call split every tile)
            assert rel_l2(runs[0][lk].grad.cpu().numpy(), b.cpu().numpy()) < 1e-4, lk


def test_image_bits_do_not_depend_on_the_mixture():
    """first call (every tile split), second call (the backward's flags: most tiles walked by one wave): the image and the per-pixel state are bit-identical."""
    inp = uneven_scene(seed=97, opacity="trained")
    g = rnd((3, inp["H"], inp["W"]), 7)
    rasterizer._order_hints.clear()
    st = hipref.settings_from(inp, "cuda", False)
    o1, ist1, _, _, walked1, _ = step(inp, g, st)
    o2, ist2, _, meta2, walked2, _ = step(inp, g, st)
    assert meta2[11] == 1 and not np.array_equal((walked1[:, 1:] > 0).any(axis=1), (walked2[:, 1:] > 0).any(axis=1))
    assert np.array_equal(o1["color"], o2["color"])
    assert np.array_equal(ist1["final_T"], ist2["final_T"]) and np.array_equal(ist1["n_contrib"], ist2["n_contrib"])
    assert np.array_equal(walked1.max(axis=1), walked2.max(axis=1))


def test_geo_backward_mixture():
    """render_geo: the forward keeps quadrant waves on such frames, the backward picks per tile (render_bwd_geo_hybrid_kernel)."""
    inp = syn.make_scene(9000, 640, 400, sh_degree=1, seed=99, opacity="trained", with_planes=True, anisotropy="plane", cluster=0.5)
    inp["scales"] = (inp["scales"] * 5.0).astype(np.float32)
    inp["all_map"] = syn.plane_all_map(inp["means3D"], inp["scales"], inp["rotations"], inp["_cam"])
    inp = add_sources(inp, n_src=3, L=4)
    H, W = inp["H"], inp["W"]
    nt = ((W + 15) // 16) * ((H + 15) // 16)
    assert 768 <= nt < 4096
    grads = {"color": rnd((3, H, W), 7), "normal_map": rnd((3, H, W), 8), "median_depth": rnd((1, H, W), 9), "warped_image": rnd((15, H, W), 10)}
    ref = oracle.forward(inp, cull=True)
    gb = oracle.backward(inp, ref, grads["color"], grads["normal_map"], grads["median_depth"], grads["warped_image"])
    st = hipref.settings_from(inp, "cuda", False)
    lv = hipref.leaf_inputs(inp, "cuda", True)
    outs = dict(zip(NAMES, rasterizer.GaussianRasterizer(st)(means3D=lv["means3D"], means2D=lv["means2D"], means2D_abs=lv["means2D_abs"], opacities=lv["opacities"],
                                                            shs=lv["shs"], colors_precomp=lv["colors_precomp"], scales=lv["scales"], rotations=lv["rotations"],
                                                            cov3D_precomp=lv["cov3D_precomp"], all_map=lv["all_map"])))
    loss = 0
    for k, g in grads.items():
        loss = loss + (outs[k] * torch.as_tensor(g, device="cuda")).sum()
    loss.backward(retain_graph=True)
    torch.cuda.synchronize()
    order = arena_words(outs, inp, "tile_order", int(_lib.load().ibgs_tile_order_slots(W, H)))
    tiles = order[order != 0xFFFFFFFF]
    assert np.array_equal(np.sort(tiles & ~np.uint32(SPLIT)), np.arange(nt, dtype=np.uint32))
    nsplit = int(((tiles & SPLIT) != 0).sum())
    assert 8 < nsplit < nt - 50, ("no mixture", nsplit, nt)
    check_grads(lv, gb, tol=GEO_GRAD_TOL)
