"""The reference's `if (power > 0.0f) continue;` (forward.cu:420, backward.cu:645) in the oracle: where it fires, and that the tile cull
never changes a result on the conics for which it can (they are exempt from the cull: oracle conic_is_risky = csrc/common.h)."""
import numpy as np

import oracle
from tests.scenes import giant_needles, scene

PUBLIC = ("color", "radii", "final_T")


def _risky(ref):
    co = ref["conic_opacity"][ref["radii"] > 0]
    return co[:, 1] * co[:, 1] > np.float32(0.99999) * (co[:, 0] * co[:, 2])


def test_power_skip_never_fires_on_ordinary_and_plane_like_scenes():
    for an in (None, "plane", "needle", "mixed"):
        inp = scene(P=3000, W=160, H=112, deg=1, seed=5, opacity="trained", anisotropy=an)
        ref = oracle.forward(inp, cull=True)
        assert oracle.power_skips()[0] == 0, an
        assert not _risky(ref).any(), an          # the bound of conic_is_risky is what makes the fast path safe for all of them


def test_power_skip_fires_on_giant_needles_and_the_cull_leaves_them_alone():
    inp = giant_needles()
    full = oracle.forward(inp, cull=False)
    n_full = oracle.power_skips()[0]
    g = np.random.default_rng(0).normal(size=(3, inp["H"], inp["W"])).astype(np.float32)
    gb_full = oracle.backward(inp, full, g)
    assert n_full > 1000 and oracle.power_skips()[1] > 1000
    assert _risky(full).mean() > 0.5
    culled = oracle.forward(inp, cull=True)
    gb_cull = oracle.backward(inp, culled, g)
    for k in PUBLIC:
        assert np.array_equal(culled[k], full[k]), k
    for k in gb_full:
        assert np.allclose(gb_cull[k], gb_full[k], rtol=1e-6, atol=1e-9), k
    # a risky Gaussian keeps the reference's rectangle, every tile of it
    r = culled["rect4"].astype(np.int64); area = (r[:, 2] - r[:, 0]) * (r[:, 3] - r[:, 1])
    vis = culled["radii"] > 0
    co = culled["conic_opacity"]
    risky = vis & (co[:, 1] * co[:, 1] > np.float32(0.99999) * (co[:, 0] * co[:, 2]))
    assert np.array_equal(culled["tiles_touched"][risky], area[risky].astype(np.uint32))
    assert np.array_equal(culled["tiles_touched"][risky], full["tiles_touched"][risky])
