"""ibgs_forward_args.rendered_hint (deferred R read-back, include/ibgs_rast.h): whatever the hint -- absent,
generous, far too small -- the forward returns the same R, the same sorted lists and bit-identical images, and
the backward works from an arena carved for a capacity other than R."""
import numpy as np
import pytest
import torch

from ibgs_amd import rasterizer, synthetic as syn
from tests import hipref

pytestmark = pytest.mark.gpu


def _run(inp, hint_state, geo=False):
    """hint_state: None = synchronous sizing, int = pretend the previous call returned that R."""
    key_geo = bool(inp.get("render_geo", False))
    rasterizer._last_rendered.clear()
    old = rasterizer.RENDERED_HINT
    rasterizer.RENDERED_HINT = hint_state is not None
    if hint_state is not None:
        P, W, H = inp["means3D"].shape[0], int(inp["W"]), int(inp["H"])
        rasterizer._last_rendered[(torch.cuda.current_device(), P, W, H, key_geo, bool(inp.get("render_depth_only", False)))] = hint_state
    try:
        outs, lv, _ = hipref.run_forward(inp)
        ist = hipref.internal_state(outs, inp)
    finally:
        rasterizer.RENDERED_HINT = old
    return outs, lv, ist


@pytest.mark.parametrize("cfg", ["colour", "geo"])
def test_results_do_not_depend_on_the_hint(cfg):
    if cfg == "colour":
        inp = syn.make_scene(20000, 320, 240, sh_degree=3, seed=4, opacity="trained")
    else:
        from tests.test_gpu_parity import add_sources, scene
        inp = add_sources(scene(P=4000, W=192, H=128, deg=2, seed=26, opacity="trained", planes=True, scale_mul=1.5), n_src=3, L=4)
    base_o, base_l, base = _run(inp, None)
    R = base["R"]
    # (the capacity is recovered from the arena's size, whose carve-up is 128-byte aligned: it is known to a few dozen entries)
    assert abs(base["binning_capacity"] - R) < 64          # synchronous sizing: carved for R itself
    g = torch.randn_like(base_o["color"])
    (base_o["color"] * g).sum().backward()
    for prev in (R, 3 * R, max(R // 10, 1), 1):
        o, l, st = _run(inp, prev)
        hint = prev + prev // 4 + 4096
        assert st["R"] == R and int(o["color"].grad_fn.num_rendered) == R
        assert abs(st["binning_capacity"] - (max(hint, R) if hint >= R else R)) < 64
        assert np.array_equal(st["point_list"], base["point_list"]) and np.array_equal(st["ranges"], base["ranges"])
        for k in ("color", "median_depth", "normal_map", "warped_image", "cam_feat", "use_first_src_frame_mask"):
            assert torch.equal(o[k], base_o[k]), (k, prev)
        (o["color"] * g).sum().backward()
        for k in ("means3D", "opacities", "scales"):
            a, b = l[k].grad, base_l[k].grad          # same lists, same images; only the float-atomic summation order differs
            assert float((a - b).norm() / (b.norm() + 1e-30)) < 1e-5, (k, prev)


def test_hint_follows_the_previous_call():
    inp = syn.make_scene(5000, 160, 120, sh_degree=1, seed=2)
    rasterizer._last_rendered.clear()
    o1, _, _ = hipref.run_forward(inp, requires_grad=False)
    assert [rasterizer.LAST_BINNING_CAPACITY] == rasterizer._last_rendered[next(iter(rasterizer._last_rendered))]
    R = rasterizer.LAST_BINNING_CAPACITY
    o2, _, _ = hipref.run_forward(inp, requires_grad=False)
    assert rasterizer.LAST_BINNING_CAPACITY == R + R // 4 + 4096 and torch.equal(o1["color"], o2["color"])
    # debug mode keeps the reference's stage-by-stage synchronous behaviour
    o3, _, _ = hipref.run_forward(inp, debug=True, requires_grad=False)
    assert rasterizer.LAST_BINNING_CAPACITY == R and torch.equal(o1["color"], o3["color"])


def test_hint_larger_than_an_empty_result():
    """A generous hint while nothing is visible (R = 0): kernels sized for the hint find a zero count on the device."""
    inp = syn.make_scene(800, 96, 64, sh_degree=0, seed=3)
    inp["means3D"] = (inp["means3D"] + np.array([0, 0, 300.0], np.float32)).astype(np.float32)
    inp["bg"] = np.array([0.2, 0.4, 0.6], np.float32)
    o, l, st = _run(inp, 50000)
    assert st["R"] == 0 and int(o["color"].grad_fn.num_rendered) == 0
    assert torch.allclose(o["color"], torch.tensor([0.2, 0.4, 0.6], device="cuda")[:, None, None].expand_as(o["color"]))
    o["color"].sum().backward()
    assert all(v.grad is None or not v.grad.any() for v in l.values() if v is not None)
