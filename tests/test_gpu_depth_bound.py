"""ibgs_forward_args.depth_bound_hint / depth_bound_out (include/ibgs_rast.h): a per-camera, per-tile depth behind which the camera's previous forward reached
nothing lets the preprocess stage drop most visible Gaussians of a saturated scene before the sort, the SH pass and the binning.  The hint is a guess; the
result must not be one.  Checked here: frames rendered with the hint are BIT-identical to frames without it -- images, radii, per-pixel arena state and,
in the deterministic backward mode, every gradient -- while the lists really are shorter; a bound the scene has moved past (or a buffer of garbage) is
caught on the device and repaired by the guarded second pass; a rendered_hint that is too small on top of it still gives the unbounded frame."""
import numpy as np
import pytest
import torch

from ibgs_amd import _lib, rasterizer
from tests import hipref
from tests.scenes import add_sources, scene

pytestmark = pytest.mark.gpu

IMG_KEYS = ("color", "radii", "normal_map", "median_depth", "cam_feat", "warped_image", "min_depth_diff", "camera_ray", "use_first_src_frame_mask")
PIX_KEYS = ("final_T", "n_contrib")          # (the per-pixel words every mode writes in full; what the geo pass keeps beside them shows in its outputs and gradients)
GRAD_KEYS = ("means3D", "means2D", "means2D_abs", "opacities", "shs", "scales", "rotations", "all_map")


def meta_words(outs, inp):
    lib = _lib.load()
    W, H = int(inp["W"]), int(inp["H"])
    img = outs["color"].grad_fn.saved_tensors[-1].cpu().numpy()
    off = lib.ibgs_img_offset(W, H, b"meta")
    return np.frombuffer(img.tobytes()[off:off + 128], dtype=np.uint32).copy()


def one_frame(inp, bound, weights):
    outs, lv, _ = hipref.run_forward(inp, depth_bound=bound)
    st = hipref.internal_state(outs, inp)
    meta = meta_words(outs, inp)
    full_R = int(outs["color"].grad_fn.num_rendered)
    loss = sum((outs[k] * w).sum() for k, w in weights.items())
    loss.backward()
    grads = {k: lv[k].grad.clone() for k in GRAD_KEYS if lv.get(k) is not None and lv[k].grad is not None}
    return {"outs": {k: outs[k].detach().clone() for k in IMG_KEYS if outs.get(k) is not None}, "st": st, "meta": meta, "R": full_R, "grads": grads, "grads_of": tuple(weights)}


def frames(inp, n, bound, mutate=None):
    """n forwards + backwards of one camera; `mutate(i, inp)` may return changed inputs for frame i."""
    rasterizer._bound_hints.clear(); rasterizer._last_rendered.clear()
    geo = bool(inp.get("render_geo", False))
    H, W = int(inp["H"]), int(inp["W"])
    gen = torch.Generator(device="cuda").manual_seed(3)
    weights = {"color": torch.randn(3, H, W, device="cuda", generator=gen)}
    if geo:
        weights["normal_map"] = torch.randn(3, H, W, device="cuda", generator=gen)
        weights["median_depth"] = torch.randn(1, H, W, device="cuda", generator=gen)
        weights["warped_image"] = torch.randn(15, H, W, device="cuda", generator=gen)
    old = rasterizer.DETERMINISTIC
    rasterizer.DETERMINISTIC = True
    try:
        out = []
        for i in range(n):
            cur = mutate(i, inp) if mutate else inp
            out.append(one_frame(cur, bound, weights))
        return out
    finally:
        rasterizer.DETERMINISTIC = old


@pytest.fixture(autouse=True)
def one_camera(monkeypatch):
    """A trainer's cameras keep their view matrix on the device and are recognised by its address; tests/hipref.py uploads a fresh tensor per call, which
    the allocator may or may not put where the last one was.  Here every call is "the" camera of its frame size and mode."""
    monkeypatch.setattr(rasterizer, "_camera_key", lambda viewmatrix, device, W, H, geo, stream: ("test camera", W, H, bool(geo)))


def same_frame(a, b, what):
    for k in a["outs"]:
        assert torch.equal(a["outs"][k], b["outs"][k]), (what, k)
    for k in PIX_KEYS:
        assert np.array_equal(a["st"][k], b["st"][k]), (what, k)
    assert a["R"] == b["R"], what
    assert a["grads"].keys() == b["grads"].keys()
    for k in a["grads"]:
        assert torch.equal(a["grads"][k], b["grads"][k]), (what, "grad", k)


def dense(geo, W=416, H=288, P=60000):
    inp = scene(P=P, W=W, H=H, deg=3, seed=31, opacity="trained", planes=geo, scale_mul=2.0)
    return add_sources(inp, n_src=3, L=4, depth=np.full((3, H, W), 4.0, np.float32)) if geo else inp


@pytest.mark.parametrize("geo", [False, True], ids=["colour", "geo"])
@pytest.mark.parametrize("size", [(416, 288), (1280, 720)], ids=["hybrid", "large"])
def test_bounded_frames_are_the_unbounded_frames(geo, size):
    W, H = size
    inp = dense(geo, W, H, P=60000 if W < 1000 else 150000)
    ref = frames(inp, 2, False)
    got = frames(inp, 4, True)
    for i, f in enumerate(got):
        same_frame(f, ref[min(i, 1)], "frame %d" % i)
        assert i == 0 or f["meta"][12] == 0, "a bound this camera left itself, on an unchanged scene, holds"          # (frame 0 has no rendered_hint: the word is not written)
    # frame 0 sizes R synchronously (no rendered_hint yet: no bound either way), frame 1 runs under the +inf buffer and leaves the first real bound,
    # frames 2 and 3 are bounded: shorter lists, fewer Gaussians in the depth sort
    assert got[1]["st"]["R"] == got[1]["R"]
    for f in got[2:]:
        assert f["st"]["R"] < f["R"], (f["st"]["R"], f["R"])          # (how much goes depends on how saturated the scene is: printed below)
        assert len(f["st"]["order"]) < len(ref[1]["st"]["order"])
    assert got[3]["st"]["R"] == got[2]["st"]["R"], "the bound is stable on a static scene"
    print("\n[depth bound %s %dx%d] pairs %d -> %d, sorted Gaussians %d -> %d" % ("geo" if geo else "colour", W, H, got[2]["R"], got[2]["st"]["R"],
                                                                                   len(ref[1]["st"]["order"]), len(got[2]["st"]["order"])))


def test_a_bound_the_scene_has_moved_past_is_repaired():
    """Frames 0-2 on the scene, then its front half turns transparent: pixels no longer terminate where they did, the stale bound is violated, the repair
    pass redoes the frame -- and the frame after that runs bounded again, under the new bound."""
    inp = dense(False)
    d = inp["means3D"] @ inp["viewmatrix"].reshape(4, 4)[:3, 2] + inp["viewmatrix"].reshape(4, 4)[3, 2]          # view-space depth (row-vector convention)
    thin = dict(inp); thin["opacities"] = np.where(d[:, None] < np.median(d), 0.02, inp["opacities"]).astype(np.float32)
    mutate = lambda i, base: base if i < 3 else thin
    ref = frames(inp, 5, False, mutate)
    got = frames(inp, 5, True, mutate)
    for i, f in enumerate(got):
        same_frame(f, ref[min(i, 4)] if i >= 3 else ref[min(i, 1)], "frame %d" % i)
    assert got[2]["meta"][12] == 0 and got[2]["st"]["R"] < got[2]["R"]
    assert got[3]["meta"][12] == 1 and got[3]["meta"][13] > 0, "the stale bound must have been caught"
    assert got[3]["st"]["R"] == got[3]["R"], "the repaired frame holds the unbounded lists"
    assert got[4]["meta"][12] == 0 and got[4]["st"]["R"] < got[4]["R"]


@pytest.mark.parametrize("fill", ["zeros", "tiny", "random", "nan", "negative"])
def test_garbage_in_the_hint_buffer_costs_time_only(fill):
    inp = dense(False)
    ref = frames(inp, 2, False)
    rasterizer._bound_hints.clear(); rasterizer._last_rendered.clear()
    warm = frames(inp, 2, True)          # (leaves the camera's buffer and the rendered_hint history behind)
    buf = next(iter(rasterizer._bound_hints.values()))
    gen = torch.Generator(device="cuda").manual_seed(5)
    if fill == "zeros": buf.zero_()
    elif fill == "tiny": buf.fill_(0.3)
    elif fill == "random": buf.copy_(torch.rand(buf.shape, device="cuda", generator=gen) * 12.0)
    elif fill == "nan": buf.fill_(float("nan"))
    else: buf.fill_(-1.0)
    old = rasterizer.DETERMINISTIC
    rasterizer.DETERMINISTIC = True
    try:
        H, W = int(inp["H"]), int(inp["W"])
        weights = {"color": torch.randn(3, H, W, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))}
        f = one_frame(inp, True, weights)
        g = one_frame(inp, True, weights)
    finally:
        rasterizer.DETERMINISTIC = old
    same_frame(f, ref[1], fill); same_frame(g, ref[1], fill + " (next frame)")
    same_frame(warm[1], ref[1], "warm-up")
    assert f["meta"][12] == 1, "the garbage must have been caught"
    assert g["meta"][12] == 0 and g["st"]["R"] < g["R"], "the repaired frame left a sound bound"


def test_rendered_hint_too_small_under_a_bound():
    """The repair of a too small rendered_hint (binning + blend once more at the exact size) on a frame whose geometry state is the bounded one."""
    inp = dense(False)
    ref = frames(inp, 2, False)
    got = frames(inp, 3, True)
    key = next(iter(rasterizer._last_rendered))
    rasterizer._last_rendered[key] = [1000]          # pretend the last frames were nearly empty
    old = rasterizer.DETERMINISTIC
    rasterizer.DETERMINISTIC = True
    try:
        H, W = int(inp["H"]), int(inp["W"])
        weights = {"color": torch.randn(3, H, W, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))}
        misses = rasterizer.HINT_MISSES
        f = one_frame(inp, True, weights)
        assert rasterizer.HINT_MISSES == misses + 1
        g = one_frame(inp, True, weights)
    finally:
        rasterizer.DETERMINISTIC = old
    same_frame(f, ref[1], "hint miss"); same_frame(g, ref[1], "frame after the miss")
    same_frame(got[2], ref[1], "bounded")
    assert f["st"]["R"] == f["R"] and g["st"]["R"] < g["R"]


def test_split_sh_arrays_under_a_bound():
    """`shs_rest` (the model's two SH arrays): the SH pass of a bounded frame skips the dropped Gaussians, that of the repair pass must not."""
    from ibgs_amd.rasterizer import GaussianRasterizer
    inp = dense(False)
    st = hipref.settings_from(inp, "cuda")

    def frame(bound, opac):
        lv = hipref.leaf_inputs(inp, "cuda")
        dc = lv["shs"].detach()[:, :1].contiguous().requires_grad_(True); rest = lv["shs"].detach()[:, 1:].contiguous().requires_grad_(True)
        o = (lv["opacities"].detach() * opac).requires_grad_(True)
        old = rasterizer.DEPTH_BOUND
        rasterizer.DEPTH_BOUND = bound
        try:
            outs = GaussianRasterizer(st)(means3D=lv["means3D"], means2D=lv["means2D"], means2D_abs=lv["means2D_abs"], opacities=o, shs=dc, shs_rest=rest,
                                          scales=lv["scales"], rotations=lv["rotations"])
        finally:
            rasterizer.DEPTH_BOUND = old
        meta = meta_words({"color": outs[0]}, inp)
        outs[0].square().sum().backward()
        return outs[0].detach().clone(), dc.grad.clone(), rest.grad.clone(), o.grad.clone(), meta

    old = rasterizer.DETERMINISTIC
    rasterizer.DETERMINISTIC = True
    try:
        plan = (1.0, 1.0, 1.0, 1.0, 0.05, 0.05)          # frames 0-3 on the scene, then every opacity drops: the bound of frame 3 cannot hold in frame 4
        rasterizer._bound_hints.clear(); rasterizer._last_rendered.clear()
        ref = [frame(False, k) for k in plan]
        rasterizer._bound_hints.clear(); rasterizer._last_rendered.clear()
        got = [frame(True, k) for k in plan]
    finally:
        rasterizer.DETERMINISTIC = old
    for i, (a, b) in enumerate(zip(got, ref)):
        for x, y, what in zip(a[:4], b[:4], ("color", "dL/dsh dc", "dL/dsh rest", "dL/dopacity")):
            assert torch.equal(x, y), (i, what)
    assert got[3][4][12] == 0 and got[4][4][12] == 1


@pytest.mark.parametrize("shape", ["tile", "quadrant"])
@pytest.mark.parametrize("geo", [False, True], ids=["colour", "geo"])
def test_forced_wave_shapes_under_a_bound(shape, geo):
    """The `done` words every blend kernel variant leaves (one wave per tile, one per quadrant, two per tile) feed the same check."""
    old = rasterizer.WAVE_SHAPE
    rasterizer.WAVE_SHAPE = shape
    try:
        inp = dense(geo, 208, 144, P=30000)
        ref = frames(inp, 2, False)
        got = frames(inp, 4, True)
    finally:
        rasterizer.WAVE_SHAPE = old
    for i, f in enumerate(got):
        same_frame(f, ref[min(i, 1)], "%s frame %d" % (shape, i))
    assert got[3]["meta"][12] == 0 and got[3]["st"]["R"] < got[3]["R"]


def test_nothing_visible_and_tiny_frames_under_a_bound():
    """R = 0 (every Gaussian behind the camera) and a frame of a single tile row: the bound buffers exist, nothing is dropped, nothing breaks."""
    inp = scene(P=500, W=96, H=16, deg=1, seed=3, opacity="trained", scale_mul=3.0)
    ref = frames(inp, 2, False)
    got = frames(inp, 4, True)
    for i, f in enumerate(got):
        same_frame(f, ref[min(i, 1)], "one tile row, frame %d" % i)
    far = dict(inp); far["means3D"] = (inp["means3D"] - np.array([0, 0, 500.0], np.float32)).astype(np.float32)
    ref = frames(far, 2, False)
    got = frames(far, 4, True)
    for i, f in enumerate(got):
        assert f["R"] == 0 or f["R"] == ref[min(i, 1)]["R"]
        assert torch.equal(f["outs"]["color"], ref[min(i, 1)]["outs"]["color"])
