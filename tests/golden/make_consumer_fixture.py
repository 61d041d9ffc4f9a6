"""Generates tests/golden/consumer.npz IN THE BUILD CONTAINER: the oracle's geo outputs on a small seeded scene are fed to the
reference's only consumer of those tensors -- `fuse_color` + a seeded `ColorFusionResidualNet` on CPU
(/root/reference/color_aggregation_network.py:156-246, imported read-only) -- and what the consumer built from them is recorded:

  * the exact tensors `fuse_color` handed to the network (captured with a forward pre-hook): per-view features (HW, levels, 7),
    ray directions (HW, 3), rendered colours (HW, 3)  -> pins the LAYOUT of cam_feat (20,H,W), warped_image (15,H,W),
    camera_ray (3,H,W) (SURVEY 8(c) cross-check 5, Appendix B `fuse_color_smoke`);
  * `image_pred`, `residual`, `valid_warp_mask`, `nb_valid_warp_level` it returned (finite, sensible);
  * the same with exposure correction on (exercises use_first_src_frame_mask through compute_exposure_affine_matrix);
  * the oracle outputs themselves, so that tests can tell whether today's oracle still is the one the consumer accepted.

The fixture is data only; nothing of the reference is copied.  Re-run:  python tests/golden/make_consumer_fixture.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference")

import oracle  # noqa: E402
from tests.scenes import consumer_scene  # noqa: E402
from color_aggregation_network import ColorFusionResidualNet, fuse_color  # noqa: E402

inp = consumer_scene()
H, W = inp["H"], inp["W"]
f = oracle.forward(inp)
pkg = {k: torch.tensor(f[src]) for k, src in (("render", "color"), ("cam_feat", "cam_feat"), ("warped_image", "warped_image"),
                                              ("min_depth_diff", "min_depth_diff"), ("camera_ray", "camera_ray"),
                                              ("use_first_src_frame_mask", "use_first_src_frame_mask"))}
torch.manual_seed(0)
net = ColorFusionResidualNet(height=H, width=W)
seen = {}
net.register_forward_pre_hook(lambda m, args: seen.update(x_views=args[0].detach().clone(), ray_dir=args[1].detach().clone(), c_3dgs=args[2].detach().clone()))
out = {"H": H, "W": W}
for name, expo in (("plain", False), ("exposure", True)):
    opts = types.SimpleNamespace(enable_exposure_correction=expo, nb_visible_src_frames=3, residual_resolution_scale=1.0)
    with torch.no_grad():
        r = fuse_color(pkg, net, None, None, None, 0, opts)
    assert r is not None and torch.isfinite(r["image_pred"]).all()
    out[name + "_image_pred"] = r["image_pred"].numpy()
    out[name + "_residual"] = r["residual"].numpy()
    out[name + "_valid_warp_mask"] = r["valid_warp_mask"].numpy()
    out[name + "_levels"] = np.int32(r["nb_valid_warp_level"])
    out[name + "_x_views"] = seen["x_views"].numpy()
    out[name + "_ray_dir"] = seen["ray_dir"].numpy()
    out[name + "_c_3dgs"] = seen["c_3dgs"].numpy()
for k in ("color", "cam_feat", "warped_image", "min_depth_diff", "camera_ray", "use_first_src_frame_mask", "median_depth", "normal_map"):
    out["oracle_" + k] = f[k]
np.savez_compressed(os.path.join(HERE, "consumer.npz"), **out)
print("consumer.npz written: levels", out["plain_levels"], "valid pixels", float(out["plain_valid_warp_mask"].mean()),
      "slots used per level", [(np.abs(f["warped_image"][3 * k:3 * k + 3]).sum(0) > 0).mean() for k in range(5)])
