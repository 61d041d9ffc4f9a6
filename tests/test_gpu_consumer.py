"""HIP geo outputs on the scene of tests/golden/consumer.npz: the tensors the reference's consumer would build from them
(tests/scenes.fuse_color_inputs, pinned against reference code by tests/test_oracle_consumer.py) equal the ones it built from
the oracle's outputs when the fixture was made."""
import os

import numpy as np
import pytest

from tests import hipref
from tests.scenes import consumer_scene, fuse_color_inputs

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "consumer.npz")


@pytest.mark.gpu
def test_hip_outputs_reach_the_consumer_like_the_oracles():
    d = np.load(G)
    outs, _, _ = hipref.run_forward(consumer_scene(), debug=True, requires_grad=False)
    o = hipref.to_np(outs)
    x, ray, c, levels = fuse_color_inputs(o["color"], o["cam_feat"], o["warped_image"], o["camera_ray"])
    assert levels == int(d["plain_levels"])
    # validity of a slot is a threshold on an interpolated depth: allow a handful of pixels to flip, compare the rest tightly
    same = np.all((np.abs(x[:, :, 3:]).sum(-1) > 0) == (np.abs(d["plain_x_views"][:, :, 3:]).sum(-1) > 0), axis=1)
    assert same.mean() > 0.995
    assert np.abs(x[same] - d["plain_x_views"][same]).max() < 2e-4
    assert np.abs(ray[same] - d["plain_ray_dir"][same]).max() < 2e-5 and np.abs(c - d["plain_c_3dgs"]).max() < 1e-5
    assert np.array_equal(o["use_first_src_frame_mask"][0].reshape(-1)[same], d["oracle_use_first_src_frame_mask"][0].reshape(-1)[same])
    assert np.abs(o["min_depth_diff"][0].reshape(-1)[same] - d["oracle_min_depth_diff"][0].reshape(-1)[same]).max() < 2e-4
