"""T1 (rasterizer_impl.cu:34-148): the reference packs the source images into textures in the forward AND again in the
backward of every geo call.  Here the backward reuses the forward's pack when nothing else used the per-stream scratch in
between, and packs again when something did.  Both cases must give the gradients of an isolated forward + backward."""
import numpy as np
import pytest
import torch

from ibgs_amd import rasterizer, renderer, simple_scene
from tests.metrics import rel_l2
from tests.test_gpu_fused_planes import _scene

pytestmark = pytest.mark.gpu

NAMES = ("_xyz", "_rotation", "_scaling", "_opacity", "_features_dc")


def _forward(view, g, dev, cams, scene, pipe, args, bg):
    pc = simple_scene.SimpleGaussians(g, sh_degree=2, device=dev)
    out = renderer.render(cams[view], pc, scene, pipe, args, bg, learnt_normal=False, nb_src_frames=3, buffer_length=4,
                          render_geo=True, return_depth_normal=False)
    gen = torch.Generator(device=dev).manual_seed(11 + view)
    loss = (out["warped_image"] * torch.randn(out["warped_image"].shape, device=dev, generator=gen)).sum() + out["render"].sum()
    return pc, loss


def _grads(pc):
    return {n: getattr(pc, n).grad.detach().cpu().numpy() for n in NAMES}


def test_backward_reuses_or_repacks_the_source_rgba():
    dev, g, cams, scene, pipe, args, bg = _scene()
    with torch.no_grad():
        pc0 = simple_scene.SimpleGaussians(g, sh_degree=2, device=dev)
        for j in sorted(set(cams[0].nearest_id) | set(cams[1].nearest_id)):
            scene.rendered_depth_list[j] = renderer.render_depth(cams[j], pc0, scene, pipe, args, bg, True, 3, 4)
    assert list(cams[0].nearest_id) != list(cams[1].nearest_id)

    alone = {}
    for v in (0, 1):
        pc, loss = _forward(v, g, dev, cams, scene, pipe, args, bg)
        before = rasterizer._tex_writes[0]
        loss.backward()
        assert rasterizer._tex_writes[0] == before, "an isolated backward must not pack again"
        alone[v] = _grads(pc)

    # two forwards, then the two backwards.  Round 5: every source stack keeps its own pack (up to TEX_CACHE_BYTES), so neither backward packs again ...
    pc_a, loss_a = _forward(0, g, dev, cams, scene, pipe, args, bg)
    pc_b, loss_b = _forward(1, g, dev, cams, scene, pipe, args, bg)
    before = rasterizer._tex_writes[0]
    loss_a.backward(); loss_b.backward()
    assert rasterizer._tex_writes[0] == before, "each stack's pack is still held: no backward packs again"
    for v, pc in ((0, pc_a), (1, pc_b)):
        got = _grads(pc)
        for n in NAMES:
            assert rel_l2(got[n], alone[v][n]) < 1e-5, (v, n, rel_l2(got[n], alone[v][n]))
    # ... and with room for ONE pack only (the behaviour until round 4): view 0's pack is evicted by view 1's forward
    old_cap = rasterizer.TEX_CACHE_BYTES
    rasterizer.TEX_CACHE_BYTES = 1
    rasterizer._tex_pool.clear()          # (packs held from above would simply be found again: eviction happens when a pack is ADDED)
    try:
        pc_a, loss_a = _forward(0, g, dev, cams, scene, pipe, args, bg)
        pc_b, loss_b = _forward(1, g, dev, cams, scene, pipe, args, bg)
        before = rasterizer._tex_writes[0]
        loss_a.backward()
        assert rasterizer._tex_writes[0] == before + 1, "view 0's backward has to pack its own sources again"
        loss_b.backward()                                  # ... which evicted view 1's pack in turn
        assert rasterizer._tex_writes[0] == before + 2
    finally:
        rasterizer.TEX_CACHE_BYTES = old_cap
    for v, pc in ((0, pc_a), (1, pc_b)):
        got = _grads(pc)
        for n in NAMES:
            assert np.abs(alone[v][n]).sum() > 0, n
            assert rel_l2(got[n], alone[v][n]) < 1e-5, (v, n, rel_l2(got[n], alone[v][n]))


def test_forward_packs_once_per_source_stack():
    """A forward that is handed the very image stack whose pack still sits in the stream's scratch skips the pack kernel (TEX_CACHE); an
    in-place write to the stack, another stack object or another call in between pack again.  Outputs identical either way."""
    from tests import hipref
    from tests.scenes import add_sources, scene as mk
    inp = add_sources(mk(P=1500, W=112, H=80, deg=1, seed=5, opacity="trained", planes=True, scale_mul=1.6), n_src=3, L=4)
    st = hipref.settings_from(inp, "cuda")
    lv = hipref.leaf_inputs(inp, "cuda", requires_grad=False)

    def fwd(settings):
        r = rasterizer.GaussianRasterizer(settings)
        with torch.no_grad():
            return r(means3D=lv["means3D"], means2D=lv["means2D"], means2D_abs=lv["means2D_abs"], opacities=lv["opacities"], shs=lv["shs"],
                     scales=lv["scales"], rotations=lv["rotations"], all_map=lv["all_map"])

    w0 = rasterizer._tex_writes[0]
    a = [t.clone() for t in fwd(st)]
    assert rasterizer._tex_writes[0] == w0 + 1
    b = fwd(st)
    assert rasterizer._tex_writes[0] == w0 + 1, "the same stack again: no second pack"
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    st.src_images.mul_(0.5)                           # in-place write: the version counter moves, the pack is redone
    c = fwd(st)
    assert rasterizer._tex_writes[0] == w0 + 2
    assert not torch.equal(c[5], a[5]) and torch.equal(c[0], a[0])          # warped colours changed, the render did not
    st2 = st._replace(src_images=st.src_images.clone())          # another object with the same bytes: packed again, same result
    d = fwd(st2)
    assert rasterizer._tex_writes[0] == w0 + 3
    for x, y in zip(c, d):
        assert torch.equal(x, y)
    old = rasterizer.TEX_CACHE
    try:
        rasterizer.TEX_CACHE = False
        e = fwd(st2)
        assert rasterizer._tex_writes[0] == w0 + 4
    finally:
        rasterizer.TEX_CACHE = old
    for x, y in zip(d, e):
        assert torch.equal(x, y)


def test_inference_tensors_and_unversioned_writes():
    """(a) under `torch.inference_mode()` a tensor has no version counter to read (RuntimeError, not AttributeError) -- such a stack is
    packed on every call instead of crashing the forward; (b) a write that does not move the counter (`.data.copy_`) is invisible to the cache
    until `rasterizer.invalidate_tex_cache()` is called."""
    from tests import hipref
    from tests.scenes import add_sources, scene as mk
    inp = add_sources(mk(P=1200, W=96, H=64, deg=1, seed=6, opacity="trained", planes=True, scale_mul=1.6), n_src=2, L=4)
    lv = hipref.leaf_inputs(inp, "cuda", requires_grad=False)

    def fwd(settings):
        return rasterizer.GaussianRasterizer(settings)(means3D=lv["means3D"], means2D=lv["means2D"], means2D_abs=lv["means2D_abs"], opacities=lv["opacities"],
                                                       shs=lv["shs"], scales=lv["scales"], rotations=lv["rotations"], all_map=lv["all_map"])
    st = hipref.settings_from(inp, "cuda")
    with torch.no_grad():
        want = [t.clone() for t in fwd(st)]
    with torch.inference_mode():
        st_inf = hipref.settings_from(inp, "cuda")          # created inside: inference tensors
        w0 = rasterizer._tex_writes[0]
        a = fwd(st_inf); b = fwd(st_inf)
        assert rasterizer._tex_writes[0] == w0 + 2, "inference tensors carry no version counter: packed per call"
        for x, y, z in zip(a, b, want):
            assert torch.equal(x, y) and torch.equal(x, z)
    with torch.no_grad():
        fwd(st)
        w0 = rasterizer._tex_writes[0]
        st.src_images.data.copy_(st.src_images * 0.5)          # no version bump: the cache cannot see it ...
        stale = fwd(st)
        assert rasterizer._tex_writes[0] == w0 and torch.equal(stale[5], want[5])
        rasterizer.invalidate_tex_cache()                      # ... until told
        fresh = fwd(st)
        assert rasterizer._tex_writes[0] == w0 + 1 and not torch.equal(fresh[5], want[5])
