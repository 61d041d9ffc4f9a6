"""Seeded scenes shared by the parity tests, the fixture generators under tests/golden/ and the tools."""
import numpy as np

import oracle
from ibgs_amd import synthetic as syn


def scene(P=4000, W=208, H=144, deg=3, seed=1, opacity="init", planes=False, scale_mul=1.0, anisotropy=None):
    inp = syn.make_scene(P, W, H, sh_degree=deg, seed=seed, opacity=opacity, with_planes=planes, anisotropy=anisotropy)
    if scale_mul != 1.0:
        inp["scales"] = (inp["scales"] * scale_mul).astype(np.float32)
        if planes:
            inp["all_map"] = syn.plane_all_map(inp["means3D"], inp["scales"], inp["rotations"], inp["_cam"])
    return inp


def giant_needles(P=300, W=208, H=144, seed=5, stretch=4.0, thin=0.05, deg=1, opacity="trained"):
    """Needles far longer than the frame and thinner than a pixel: the fp32 inversion of cov2D leaves conics within rounding of singular
    (some indefinite), the regime in which the reference's `power > 0` test (forward.cu:420, backward.cu:645) decides which pairs blend.
    The longest axis of every "needle" Gaussian is stretched again, the other two shrunk."""
    inp = scene(P=P, W=W, H=H, deg=deg, seed=seed, opacity=opacity, anisotropy="needle")
    s0 = inp["scales"]
    k = np.argmax(s0, axis=1); rows = np.arange(P)
    s = s0 * thin
    s[rows, k] = s0[rows, k] * stretch
    inp["scales"] = s.astype(np.float32)
    return inp


def add_sources(inp, n_src=3, L=4, seed=5, depth=None):
    W, H = inp["W"], inp["H"]
    srcs = [syn.make_camera(W, H, azimuth_deg=a) for a in (7.0, -7.0, 14.0, -14.0, 21.0)[:n_src]]
    r2s, scp = syn.ref_to_src(inp["_cam"], srcs)
    rng = np.random.default_rng(seed)
    if depth is None:   # plausible source depths: the oracle's own depth-only render of each source view
        deps = []
        for s in srcs:
            d = dict(inp); d.update(viewmatrix=s["viewmatrix"], projmatrix=s["projmatrix"], campos=s["campos"],
                                    render_geo=False, render_depth_only=True, buffer_length=4,
                                    all_map=syn.plane_all_map(inp["means3D"], inp["scales"], inp["rotations"], s))
            deps.append(oracle.forward(d)["median_depth"])
        depth = np.stack(deps)
    inp = dict(inp)
    inp.update(render_geo=True, n_src=n_src, buffer_length=L, ref_to_src=r2s, src_cam_pos=scp,
               src_images=rng.uniform(0, 1, (n_src, 3, H, W)).astype(np.float32), src_depths=depth.astype(np.float32), depth_thr=0.05)
    return inp


def consumer_scene():
    """The scene of tests/golden/consumer.npz (make_consumer_fixture.py): small enough for a committed fixture, H and W
    divisible by 4 (the reference's ColorFusionResidualNet pools twice), 3 sources so that every slot level is used."""
    return add_sources(scene(P=700, W=48, H=32, deg=1, seed=61, opacity="trained", planes=True, scale_mul=2.2), n_src=3, L=4)


def fuse_color_inputs(render, cam_feat, warped_image, camera_ray, nb_visible_src_frames=3):
    """numpy restatement of how the reference's only consumer reads the geo outputs (color_aggregation_network.py:156-198,
    231-233, no exposure correction, residual_resolution_scale 1): slot k of `warped_image` = channels 3k..3k+2, slot k of
    `cam_feat` = channels 4k..4k+3; the number of slot levels used = the count of levels whose warped colours are not all
    zero (capped); a slot of a pixel is valid iff the sum of its 4 cam_feat values is > 0; per-view features = [warped -
    render (masked), cam_feat].  Returns (x_views (HW, levels, 7), ray_dir (HW, 3), c_3dgs (HW, 3), levels)."""
    _, H, W = render.shape
    wl = warped_image.reshape(-1, 3, H, W).transpose(2, 3, 0, 1)             # (H, W, 5, 3)
    ft = cam_feat.reshape(-1, 4, H, W).transpose(2, 3, 0, 1)                 # (H, W, 5, 4)
    levels = min(int(np.count_nonzero(wl.sum(axis=(0, 1, 3)))), nb_visible_src_frames)
    ft, wl = ft[:, :, :levels], wl[:, :, :levels]
    valid = (ft.sum(-1, keepdims=True) > 0.0).astype(np.float32)
    resid = (wl - render.transpose(1, 2, 0)[:, :, None, :]) * valid
    x = np.concatenate([resid, ft], axis=-1).reshape(H * W, levels, 7)
    return x, camera_ray.reshape(3, -1).T, render.transpose(1, 2, 0).reshape(-1, 3), levels
