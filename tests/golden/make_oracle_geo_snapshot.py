#!/usr/bin/env python3
"""Freezes the oracle's answer for the GEO path (SURVEY Appendix B `rand_small_geo`: 2 k Gaussians, 64 x 64, n_src = 3, buffer_length 4 and 5,
all four differentiable outputs in the loss) into tests/golden/oracle_geo.npz: every public forward plane, the per-pixel window state, and
for all ten gradients of the backward (rasterize_points.cu:209-219) head rows + float64 checksums.  Until round 4 only the geo FORWARD was
frozen (consumer.npz); with the B2 gradients frozen too, the oracle's median / warp backward (backward.cu:692-771, quirks Q2-Q5) and the
HIP path's can no longer drift together unnoticed.  Usage (repo root): python tests/golden/make_oracle_geo_snapshot.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from tests.scenes import add_sources, scene  # noqa: E402

HEAD = 192
LS = (4, 5)
PLANES = ("color", "normal_map", "median_depth", "cam_feat", "warped_image", "min_depth_diff", "camera_ray", "use_first_src_frame_mask")
GRADS = ("dL_dmeans3D", "dL_dmeans2D", "dL_dmeans2D_abs", "dL_dsh", "dL_dcolors", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dcov3D", "dL_dall_map")


def build(L):
    base = scene(P=2000, W=64, H=64, deg=2, seed=91, opacity="trained", planes=True, scale_mul=2.5)
    # the 3 visible Gaussians nearest to the camera become near-opaque planes that face AWAY from it (negative ray / plane depth: blended, never
    # buffered): behind their centres T falls below 0.5 with buffer slot 0 still empty -- quirk Q4 (forward.cu:515-516, backward.cu:693): no
    # median / warp gradient for those pixels
    pre = oracle.forward(base)
    vis = np.flatnonzero(pre["radii"] > 0)
    back = vis[np.argsort(pre["depths"][vis])[:3]]
    base["all_map"] = base["all_map"].copy(); base["all_map"][back, :3] = (0.0, 0.0, 1.0)
    base["opacities"] = base["opacities"].copy(); base["opacities"][back] = 0.97
    inp = add_sources(base, n_src=3, L=L, seed=17)
    r = np.random.default_rng(300 + L)
    H, W = inp["H"], inp["W"]
    g = {"color": r.standard_normal((3, H, W)).astype(np.float32), "normal_map": r.standard_normal((3, H, W)).astype(np.float32),
         "median_depth": r.standard_normal((1, H, W)).astype(np.float32), "warped_image": r.standard_normal((15, H, W)).astype(np.float32)}
    return inp, g


def snapshot(inp, g):
    f = oracle.forward(inp, cull=True)
    b = oracle.backward(inp, f, g["color"], g["normal_map"], g["median_depth"], g["warped_image"])
    out = {"num_rendered": np.int64(f["num_rendered"]), "radii": f["radii"], "n_contrib": f["n_contrib"], "cache_low": f["cache_low"], "cache_high": f["cache_high"],
           "valid_src_idx": f["valid_src_idx"].astype(np.int8)}
    for k in PLANES:
        out[k] = f[k]
    P = inp["means3D"].shape[0]
    for k in GRADS:
        a = np.asarray(b[k], np.float32).reshape(P, -1)
        out[k + "_head"] = a[:HEAD].copy()
        out[k + "_l1norm"] = np.float64(np.abs(a).astype(np.float64).sum())
        out[k + "_total"] = np.float64(a.astype(np.float64).sum())
    return out


if __name__ == "__main__":
    allv = {}
    for L in LS:
        inp, g = build(L)
        for k, v in snapshot(inp, g).items():
            allv["L%d_%s" % (L, k)] = v
    path = os.path.join(ROOT, "tests", "golden", "oracle_geo.npz")
    np.savez_compressed(path, **allv)
    print("wrote", path, os.path.getsize(path), "bytes")
