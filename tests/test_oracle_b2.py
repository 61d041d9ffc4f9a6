"""The oracle's median / warp backward (B2: backward.cu:692-771 + bilinearInterpolateBackward :55-109, quirks Q2-Q5) against the
second, independent restatement in tests/b2_restatement.py (float64 torch, tile-at-a-time, closed forms over the slot axis).
Losses touch ONLY the median depth and the warped colours, so every gradient the oracle returns here was produced by B2."""
import numpy as np
import pytest

import oracle
from tests.b2_restatement import b2_gradients
from tests.metrics import rel_l2
from tests.scenes import add_sources, scene


def _case(P, W, H, n_src, L, seed, scale_mul, which):
    base = scene(P=P, W=W, H=H, deg=1, seed=seed, opacity="trained", planes=True, scale_mul=scale_mul)
    # Q4 needs pixels whose buffer slot 0 stays empty: a few nearly opaque Gaussians whose plane faces AWAY from the camera
    # (ray/plane depth <= 0 -> never buffered) push T below 0.5 at once, so everything behind them lands in the "below" half
    pre = oracle.forward(base)
    vis = np.flatnonzero((pre["radii"] > 0) & (pre["means2D"][:, 0] < 0.3 * W))     # ... in the left part of the frame only
    back = vis[np.argsort(pre["depths"][vis])[:20]]                     # the visible Gaussians nearest to the camera
    base["opacities"] = base["opacities"].copy(); base["opacities"][back] = 0.97
    base["all_map"] = base["all_map"].copy(); base["all_map"][back, :3] = (0.0, 0.0, 1.0)
    inp = add_sources(base, n_src=n_src, L=L)
    inp["depth_thr"] = 0.3              # generous source validity: every slot count from 0 to n_src occurs on many pixels
    fwd = oracle.forward(inp)
    rng = np.random.default_rng(seed + 1)
    g_d = rng.normal(size=(1, H, W)).astype(np.float32) if which in ("depth", "both") else np.zeros((1, H, W), np.float32)
    g_w = rng.normal(size=(15, H, W)).astype(np.float32) if which in ("warp", "both") else np.zeros((15, H, W), np.float32)
    ob = oracle.backward(inp, fwd, np.zeros((3, H, W), np.float32), None, g_d, g_w)
    mine = b2_gradients(inp, fwd, g_d, g_w)
    return inp, fwd, ob, mine


def _well_conditioned(fwd):
    """Pixels whose buffer weights are not tiny: the block divides by sum_w and by the per-source weight sums, so on the
    others fp32 rounding of the oracle (the restatement is float64) is amplified without saying anything about the formulas."""
    vw = fwd["valid_src_w"]
    return (fwd["cache_sum_w"] > 1e-2) & ~(((vw > 0) & (vw < 1e-2)).any(0))


def _compare(ob, mine, tol):
    pairs = [("dL_dall_map", ob["dL_dall_map"], mine["dL_dall_map"]), ("dL_dmeans2D", ob["dL_dmeans2D"][:, :2], mine["dL_dmeans2D"]),
             ("dL_dconic", ob["dL_dconic"][:, [0, 1, 3]], mine["dL_dconic"]), ("dL_dopacity", ob["dL_dopacity"][:, 0], mine["dL_dopacity"])]
    for name, a, b in pairs:
        assert np.abs(b).max() > 0, name
        err = rel_l2(a, b)
        assert err < tol, "%s: oracle vs second restatement, relative L2 %.3e" % (name, err)
    assert not ob["dL_dall_map"][:, 3].any() and not ob["dL_dcolors"].any()


@pytest.mark.parametrize("which", ["depth", "warp", "both"])
@pytest.mark.parametrize("n_src,L,seed", [(3, 4, 71), (1, 5, 72), (5, 8, 73)])
def test_oracle_b2_equals_the_second_restatement(n_src, L, seed, which):
    """Bars: 1e-4 relative L2 where the divisions are well conditioned (measured 2e-6 .. 5e-5: what is left is the fp32
    cancellation in (depth - median) / sum_w), 1e-3 over all pixels (measured <= 4e-4, gradients up to 4e4 from buffer
    weights of 1e-5)."""
    inp, fwd, ob, mine = _case(900, 80, 64, n_src, L, seed, 2.0, which)
    H, W = inp["H"], inp["W"]
    # the scene must exercise what the block is about
    nvalid = (np.cumprod(fwd["valid_src_idx"] != -1, axis=0) > 0).sum(0)
    have = fwd["n_contrib"] > 0
    assert (nvalid[have] == 0).sum() > 20, "no pixel with zero valid sources (Q2: the median loss then yields NO plane gradient)"
    assert (nvalid == 1).sum() > 20
    if n_src >= 3:
        assert (nvalid >= 3).sum() > 15, "no pixel with three in-bounds sources (Q2: base depth gradient added three times)"
    if seed == 72:
        assert (have & (fwd["cache_low"] == 0)).sum() > 100, "no Q4 pixel (buffer slot 0 empty -> min contributor 0 -> block skipped)"
    _compare(ob, mine, 1e-3)
    good = _well_conditioned(fwd).reshape(1, H, W)
    assert good.mean() > 0.75
    rng = np.random.default_rng(seed + 2)
    g_d = (rng.normal(size=(1, H, W)) * good).astype(np.float32) if which in ("depth", "both") else np.zeros((1, H, W), np.float32)
    g_w = (rng.normal(size=(15, H, W)) * good).astype(np.float32) if which in ("warp", "both") else np.zeros((15, H, W), np.float32)
    _compare(oracle.backward(inp, fwd, np.zeros((3, H, W), np.float32), None, g_d, g_w), b2_gradients(inp, fwd, g_d, g_w), 1e-4)


def test_q2_and_q4_in_numbers():
    """Quirks stated as properties, on a scene built for them: (Q4) zeroing the incoming gradients of every pixel whose
    buffer slot 0 is empty changes nothing; (Q2) with the depth loss alone, pixels without a valid in-bounds source
    contribute no plane gradient although they do contribute to dL/dalpha."""
    inp, fwd, ob, mine = _case(900, 80, 64, 1, 5, 72, 2.0, "depth")
    H, W = inp["H"], inp["W"]
    g_d = np.random.default_rng(72).normal(size=(1, H, W)).astype(np.float32)
    q4 = (fwd["cache_low"] == 0).reshape(1, H, W)
    assert q4.sum() > 100
    masked = np.where(q4, 0.0, g_d).astype(np.float32)
    z15 = np.zeros((15, H, W), np.float32)
    a = oracle.backward(inp, fwd, np.zeros((3, H, W), np.float32), None, g_d, z15)
    b = oracle.backward(inp, fwd, np.zeros((3, H, W), np.float32), None, masked, z15)
    for k in ("dL_dall_map", "dL_dmeans2D", "dL_dopacity"):
        assert np.array_equal(a[k], b[k]), k
    nvalid = (np.cumprod(fwd["valid_src_idx"] != -1, axis=0) > 0).sum(0).reshape(1, H, W)
    only0 = np.where((nvalid == 0) & ~q4, g_d, 0.0).astype(np.float32)
    c = oracle.backward(inp, fwd, np.zeros((3, H, W), np.float32), None, only0, z15)
    assert not c["dL_dall_map"].any() and np.abs(c["dL_dopacity"]).max() > 0
    m = b2_gradients(inp, fwd, only0, z15)
    assert not m["dL_dall_map"].any() and rel_l2(c["dL_dopacity"][:, 0], m["dL_dopacity"]) < 1e-3
