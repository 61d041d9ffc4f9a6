"""A scene of the size class of the reference's garden recipe (`garden -r 4`, a 1297 x 840 frame -- neither side a multiple of the tile -- and
several million Gaussians after densification) against the oracle, forward and backward.  The dataset itself is not available; what this pins is that
nothing in the path depends on C3's sizes: 6 M Gaussians (P x 16 words of records, 32-bit list positions, per-wave tile sums, the 64-bit byte offsets of
1.1 GB of SH coefficients), 4 346 tiles (just above the switch from quadrant waves to tile waves), ragged right / bottom tiles."""
import numpy as np
import pytest
import torch

import oracle
from tests import hipref
from tests.metrics import l1, rel_l2
from tests.scenes import scene

pytestmark = pytest.mark.gpu


def test_six_million_gaussians_on_the_garden_frame_against_the_oracle():
    P, W, H = 6_000_000, 1297, 840
    inp = scene(P=P, W=W, H=H, deg=3, seed=17, opacity="trained", scale_mul=1.4)
    ref = oracle.forward(inp, cull=True)
    g = np.random.default_rng(1).standard_normal((3, H, W)).astype(np.float32)
    rb = oracle.backward(inp, ref, g)
    outs, lv, _ = hipref.run_forward(inp)
    ist = hipref.internal_state(outs, inp)
    assert ist["R"] == ref["num_rendered"] > 3 * 10**7
    assert np.array_equal(ist["point_list"], ref["point_list"]) and np.array_equal(ist["ranges"], ref["ranges"])
    assert np.array_equal(outs["radii"].cpu().numpy(), ref["radii"])
    col = outs["color"].detach().cpu().numpy()
    assert l1(col, ref["color"]) < 1e-6
    assert (ist["n_contrib"] == ref["n_contrib"]).mean() > 0.9999
    (outs["color"] * torch.as_tensor(g, device="cuda")).sum().backward()
    for k, v in {"dL_dmeans3D": "means3D", "dL_dsh": "shs", "dL_dopacity": "opacities", "dL_dscales": "scales", "dL_drotations": "rotations"}.items():
        assert rel_l2(lv[v].grad.cpu().numpy().reshape(np.asarray(rb[k]).shape), rb[k]) < 1e-3, k
    print("\n[large scene] P %d, %dx%d: R %d, visible %d" % (P, W, H, ist["R"], int((ref["radii"] > 0).sum())))
