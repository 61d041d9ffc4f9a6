"""Stage-by-stage HIP-vs-oracle report (diagnostic companion of the -m gpu tests; run on a GPU box:
`python tests/gpu_report.py [P W H]`).  Prints one line per compared quantity."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from ibgs_amd import synthetic as syn  # noqa: E402
from tests import hipref  # noqa: E402


def cmp(name, a, b, exact=False):
    a = np.asarray(a); b = np.asarray(b)
    if a.shape != b.shape:
        print("%-28s SHAPE MISMATCH %s vs %s" % (name, a.shape, b.shape)); return
    if exact or a.dtype.kind in "iub":
        bad = int((a != b).sum())
        print("%-28s exact: %d / %d differ" % (name, bad, a.size))
    else:
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        den = np.abs(b).mean() + 1e-30
        bits = int((a.view(np.uint32) != b.view(np.uint32)).sum()) if a.dtype == np.float32 and b.dtype == np.float32 else -1
        print("%-28s max|d|=%.3e mean|d|=%.3e rel=%.3e  bitdiff=%d/%d" % (name, d.max() if d.size else 0, d.mean() if d.size else 0,
                                                                      (d.mean() / den) if d.size else 0, bits, a.size))


def relL2(a, b):
    a = a.astype(np.float64).ravel(); b = b.astype(np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def main():
    P, W, H = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (10000, 400, 400)
    geo = "--geo" in sys.argv
    inp = syn.make_scene(P, W, H, sh_degree=3, seed=1, with_planes=geo)
    if geo:
        cam = inp["_cam"]
        srcs = [syn.make_camera(W, H, azimuth_deg=a) for a in (8.0, -8.0, 16.0)]
        r2s, scp = syn.ref_to_src(cam, srcs)
        rng = np.random.default_rng(5)
        inp.update(render_geo=True, n_src=3, ref_to_src=r2s, src_cam_pos=scp,
                   src_images=rng.uniform(0, 1, (3, 3, H, W)).astype(np.float32),
                   src_depths=np.full((3, 1, H, W), 4.0, np.float32))
    cull = "--nocull" not in sys.argv
    from ibgs_amd import rasterizer
    rasterizer.TILE_CULL = cull
    t = time.time(); ref = oracle.forward(inp, cull=cull); print("oracle fwd %.2fs R=%d" % (time.time() - t, ref["num_rendered"]))
    outs, leaves, st = hipref.run_forward(inp, debug=True)
    torch.cuda.synchronize()
    ist = hipref.internal_state(outs, inp)
    o = hipref.to_np(outs)
    print("HIP R=%d" % ist["R"])
    cmp("radii", o["radii"], ref["radii"])
    cmp("tiles_touched", ist["tiles"], ref["tiles_touched"])
    cmp("depths", ist["depths"], ref["depths"])
    cmp("means2D", ist["rec"][:, 0:2], ref["means2D"])
    cmp("conic", ist["rec"][:, 4:7], ref["conic_opacity"][:, 0:3])
    cmp("rgb", ist["rec"][:, 8:11], ref["rgb"])
    cmp("cov3D", ist["cov3D"], ref["cov3D"])
    cb = ref["clamped"][:, 0] | (ref["clamped"][:, 1] << 1) | (ref["clamped"][:, 2] << 2)
    cmp("clamped", ist["clamped"], cb.astype(np.uint8))
    cmp("ranges", ist["ranges"], ref["ranges"])
    if ist["R"] == ref["num_rendered"]:
        cmp("point_list", ist["point_list"], ref["point_list"])
        cmp("tile keys", ist["sorted_tile_keys"], (ref["keys"] >> 32).astype(np.uint32))
    cmp("final_T", ist["final_T"], ref["final_T"])
    cmp("n_contrib", ist["n_contrib"], ref["n_contrib"])
    cmp("color", o["color"], ref["color"])
    if geo:
        for k in ("normal_map", "median_depth", "cam_feat", "warped_image", "min_depth_diff", "camera_ray", "use_first_src_frame_mask"):
            cmp(k, o[k], ref[k])
        cmp("sum_w", ist["sum_w"], ref["cache_sum_w"])
        cmp("low", ist["low_high"][:, 0], ref["cache_low"]); cmp("high", ist["low_high"][:, 1], ref["cache_high"])
        cmp("valid_idx[0]", ist["valid_idx"][0], ref["valid_src_idx"][0])
    # backward
    rng = np.random.default_rng(0)
    g = rng.normal(size=(3, H, W)).astype(np.float32)
    gn = rng.normal(size=(3, H, W)).astype(np.float32) if geo else None
    gd = rng.normal(size=(1, H, W)).astype(np.float32) if geo else None
    gw = rng.normal(size=(15, H, W)).astype(np.float32) if geo else None
    t = time.time(); rb = oracle.backward(inp, ref, g, gn, gd, gw); print("oracle bwd %.2fs" % (time.time() - t))
    loss = (outs["color"] * torch.as_tensor(g, device="cuda")).sum()
    if geo:
        loss = loss + (outs["normal_map"] * torch.as_tensor(gn, device="cuda")).sum() \
            + (outs["median_depth"] * torch.as_tensor(gd, device="cuda")).sum() \
            + (outs["warped_image"] * torch.as_tensor(gw, device="cuda")).sum()
    loss.backward()
    torch.cuda.synchronize()
    pairs = [("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("means2D_abs", "dL_dmeans2D_abs"), ("shs", "dL_dsh"),
             ("opacities", "dL_dopacity"), ("scales", "dL_dscales"), ("rotations", "dL_drotations")]
    if geo:
        pairs.append(("all_map", "dL_dall_map"))
    for lk, rk in pairs:
        gt = leaves[lk].grad
        if gt is None:
            print("%-28s grad is None" % lk); continue
        a = gt.cpu().numpy(); b = rb[rk]
        print("%-28s relL2=%.3e max|d|=%.3e  |ref|mean=%.3e" % ("grad " + lk, relL2(a, b), np.abs(a - b).max(), np.abs(b).mean()))


if __name__ == "__main__":
    main()
