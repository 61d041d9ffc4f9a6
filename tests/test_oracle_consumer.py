"""The oracle's geo outputs against what the reference's ONLY consumer of them made of them (SURVEY 8(c) cross-check 5).

tests/golden/consumer.npz was produced in the build container by tests/golden/make_consumer_fixture.py: oracle geo forward on a
small seeded scene -> reference `fuse_color` + seeded `ColorFusionResidualNet` on CPU (color_aggregation_network.py:156-246).
It holds the tensors the reference code built from cam_feat (20,H,W) / warped_image (15,H,W) / camera_ray (3,H,W) /
min_depth_diff / use_first_src_frame_mask and handed to its network, plus what came out.  Here (no reference needed):
  * today's oracle still produces the outputs the consumer accepted;
  * a numpy restatement of the consumer's tensor assembly, fed with the oracle outputs, reproduces the captured network inputs
    exactly -- i.e. slot k = channels 4k..4k+3 / 3k..3k+2, validity = "cam_feat slot sums > 0", residual = warped - render,
    ray layout -- which pins those LAYOUTS to reference code, not to this repository's reading of it;
  * the consumer's results were finite and used every slot level.
The GPU twin (tests/test_gpu_consumer.py) pushes the HIP outputs through the same restatement."""
import os

import numpy as np

import oracle
from tests.scenes import consumer_scene, fuse_color_inputs

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "consumer.npz")


def test_oracle_still_is_the_one_the_consumer_accepted():
    d = np.load(G)
    f = oracle.forward(consumer_scene())
    for k in ("color", "cam_feat", "warped_image", "min_depth_diff", "camera_ray", "use_first_src_frame_mask", "median_depth", "normal_map"):
        assert f[k].shape == d["oracle_" + k].shape, k
        np.testing.assert_allclose(f[k], d["oracle_" + k], rtol=0, atol=1e-6, err_msg=k)
    H, W = int(d["H"]), int(d["W"])
    assert f["cam_feat"].shape == (20, H, W) and f["warped_image"].shape == (15, H, W) and f["camera_ray"].shape == (3, H, W)
    assert f["min_depth_diff"].shape == (1, H, W) and f["use_first_src_frame_mask"].shape == (1, H, W)


def test_layouts_as_the_reference_consumer_reads_them():
    d = np.load(G)
    x, ray, c, levels = fuse_color_inputs(d["oracle_color"], d["oracle_cam_feat"], d["oracle_warped_image"], d["oracle_camera_ray"])
    assert levels == int(d["plain_levels"]) == 3
    assert np.array_equal(x, d["plain_x_views"]) and np.array_equal(ray, d["plain_ray_dir"]) and np.array_equal(c, d["plain_c_3dgs"])
    # every slot level carries pixels, level k never has more valid pixels than level k - 1 (slots are compacted, forward.cu:607-647)
    used = [(np.abs(d["plain_x_views"][:, k, 3:]).sum(-1) > 0).mean() for k in range(levels)]
    assert used[0] > 0.3 and used[0] >= used[1] >= used[2] > 0.0
    # valid_warp_mask = min_depth_diff < 0.999 (color_aggregation_network.py:198) marks exactly the pixels with a valid first slot
    vw = d["plain_valid_warp_mask"][0] > 0
    assert np.array_equal(vw, d["oracle_min_depth_diff"][0] < 0.999)
    assert np.array_equal(vw.reshape(-1), np.abs(d["plain_x_views"][:, 0, 3:]).sum(-1) > 0)
    # rays are unit vectors where a median depth exists
    n = np.linalg.norm(d["plain_ray_dir"], axis=1)
    assert np.all(np.abs(n[n > 0] - 1.0) < 1e-4)


def test_consumer_results_were_finite_with_and_without_exposure_correction():
    d = np.load(G)
    for name in ("plain", "exposure"):
        for k in ("image_pred", "residual"):
            a = d["%s_%s" % (name, k)]
            assert a.shape == (3, int(d["H"]), int(d["W"])) and np.isfinite(a).all() and np.abs(a).max() < 10.0
    # exposure correction fits an affine colour map on the pixels where use_first_src_frame_mask == 1: it changed the colours
    # the network saw, so the mask reached the consumer with a usable number of pixels
    m = d["oracle_use_first_src_frame_mask"][0] == 1
    assert 50 < m.sum() < m.size
    assert np.abs(d["exposure_c_3dgs"] - d["plain_c_3dgs"]).max() > 1e-3
