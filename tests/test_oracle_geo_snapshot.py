"""The frozen GEO answer (tests/golden/oracle_geo.npz, written by tests/golden/make_oracle_geo_snapshot.py; SURVEY Appendix B `rand_small_geo`:
2 k Gaussians, 64 x 64, n_src 3, buffer_length 4 and 5, Q4 pixels forced): every public forward plane, the window state, and for all ten
gradients head rows + checksums -- with normal map, median depth and warped colours in the loss.
* CPU: the oracle still reproduces it (forward bit for bit, gradients to rounding: its OpenMP accumulation order is free);
* GPU: the HIP operator against the SAME committed numbers -- so the oracle's B2 (backward.cu:692-771) and the kernels' cannot drift together."""
import os

import numpy as np
import pytest

from tests.golden import make_oracle_geo_snapshot as snap
from tests.metrics import l1, rel_l2

GOLD = os.path.join(os.path.dirname(__file__), "golden", "oracle_geo.npz")


@pytest.mark.parametrize("L", snap.LS)
def test_oracle_reproduces_its_frozen_geo_answer(L):
    gold = np.load(GOLD)
    G = lambda k: gold["L%d_%s" % (L, k)]
    inp, g = snap.build(L)
    now = snap.snapshot(inp, g)
    assert int(now["num_rendered"]) == int(G("num_rendered"))
    for k in ("radii", "n_contrib", "cache_low", "cache_high", "valid_src_idx") + snap.PLANES:
        assert np.array_equal(now[k], G(k)), k
    assert (G("cache_low") == 0).sum() > 500 and ((G("cache_low") > 0) & (G("valid_src_idx")[0] >= 0)).sum() > 500, "the fixture must hold Q4 pixels and warped ones"
    for k in snap.GRADS:
        if float(G(k + "_l1norm")) == 0.0:
            assert float(now[k + "_l1norm"]) == 0.0, k
            continue
        assert rel_l2(now[k + "_head"], G(k + "_head")) < 1e-5, k
        assert abs(float(now[k + "_l1norm"]) - float(G(k + "_l1norm"))) < 1e-5 * float(G(k + "_l1norm")), k


@pytest.mark.gpu
@pytest.mark.parametrize("L", snap.LS)
@pytest.mark.parametrize("shape", ["tile", "quadrant"])
def test_hip_against_the_frozen_geo_answer(L, shape):
    import torch
    from ibgs_amd import rasterizer
    from tests import hipref
    gold = np.load(GOLD)
    G = lambda k: gold["L%d_%s" % (L, k)]
    inp, g = snap.build(L)
    H, W = inp["H"], inp["W"]
    old = rasterizer.WAVE_SHAPE
    try:
        rasterizer.WAVE_SHAPE = shape
        outs, lv, _ = hipref.run_forward(inp)
        ist = hipref.internal_state(outs, inp)
        o = hipref.to_np(outs)
        loss = sum((outs[k] * torch.as_tensor(v, device="cuda")).sum() for k, v in g.items())
        loss.backward()
    finally:
        rasterizer.WAVE_SHAPE = old
    assert int(outs["color"].grad_fn.num_rendered) == int(G("num_rendered"))
    assert np.array_equal(o["radii"], G("radii"))
    assert l1(o["color"], G("color")) < 1e-6 and l1(o["normal_map"], G("normal_map")) < 1e-6
    assert (ist["n_contrib"] != G("n_contrib")).mean() <= 5e-4
    assert np.array_equal(ist["low_high"][:, 0], G("cache_low")) and np.array_equal(ist["low_high"][:, 1], G("cache_high"))
    upto = lambda v: np.cumprod(v != -1, axis=0) > 0          # slots are only defined up to the -1 terminator (forward.cu:648-655)
    va, vb = upto(ist["valid_idx"]), upto(G("valid_src_idx").astype(np.int32))
    same = np.all((va == vb) & (~va | (ist["valid_idx"] == G("valid_src_idx"))), axis=0)
    assert (~same).sum() <= 2, "valid-source sets differ on %d pixels" % int((~same).sum())
    ok = same.reshape(H, W)
    for k, tol in (("median_depth", 1e-4), ("cam_feat", 1e-5), ("warped_image", 1e-5), ("min_depth_diff", 1e-5), ("camera_ray", 1e-5)):
        d = np.abs(o[k] - G(k))[:, ok]
        assert d.mean() / (np.abs(G(k)[:, ok]).mean() + 1e-9) < tol, k
    assert np.array_equal(o["use_first_src_frame_mask"][0][ok], G("use_first_src_frame_mask")[0][ok])
    names = {"dL_dmeans3D": "means3D", "dL_dmeans2D": "means2D", "dL_dmeans2D_abs": "means2D_abs", "dL_dsh": "shs", "dL_dopacity": "opacities",
             "dL_dscales": "scales", "dL_drotations": "rotations", "dL_dall_map": "all_map"}          # (colors_precomp / cov3D_precomp are not inputs here)
    P = inp["means3D"].shape[0]
    for k, leaf in names.items():
        a = lv[leaf].grad.detach().cpu().numpy().reshape(P, -1)
        want = G(k + "_head")
        assert rel_l2(a[:snap.HEAD, :want.shape[1]], want) < 1e-3, (k, rel_l2(a[:snap.HEAD, :want.shape[1]], want))          # BASELINE's gradient bar
        tot = float(np.abs(a[:, :want.shape[1]]).astype(np.float64).sum())
        assert abs(tot - float(G(k + "_l1norm"))) < 1e-3 * float(G(k + "_l1norm")), k
