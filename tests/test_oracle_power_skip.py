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


def test_what_fma_contraction_leaves_undetermined():
    """The oracle against ITSELF with gcc free to contract a*b+c into fma (nvcc does by default, in an unknowable pattern): identical to
    ~1e-6 on a near-isotropic scene; on needles the two builds differ by more than the 1e-3 gradient bar in every gradient that passes
    through the inversion of cov2D -- the floor that tests/test_gpu_anisotropic.py measures its bars against."""
    from tests.metrics import l1, rel_l2
    res = {}
    for tag, an in (("iso", None), ("needle", "needle")):
        inp = scene(P=2500, W=160, H=112, deg=1, seed=31, opacity="trained", anisotropy=an)
        g = np.random.default_rng(1).normal(size=(3, 112, 160)).astype(np.float32)
        r0 = oracle.forward(inp, cull=True); g0 = oracle.backward(inp, r0, g)
        with oracle.variant("fma"):
            r1 = oracle.forward(inp, cull=True); g1 = oracle.backward(inp, r1, g)
        res[tag] = (l1(r1["color"], r0["color"]), {k: rel_l2(g1[k], g0[k]) for k in ("dL_dopacity", "dL_dscales", "dL_drotations", "dL_dmeans3D")})
    assert res["iso"][0] < 1e-6 and max(res["iso"][1].values()) < 1e-4, res["iso"]
    assert res["needle"][1]["dL_dscales"] > 1e-3 or res["needle"][1]["dL_drotations"] > 1e-3, res["needle"]
    assert res["needle"][0] < 1e-4          # the image bar of the north star still separates the two builds from a wrong image
