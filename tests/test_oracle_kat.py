"""Known-answer micro scenes for the oracle (the reference has no tests; these are hand-computable).

Camera: identity world->view (camera at the origin looking down +z), so a point (0,0,z0) projects to
the principal point ((W-1)/2, (H-1)/2) (pixel centres at integers, auxiliary.h:45-48)."""
import math

import numpy as np

import oracle
from ibgs_amd import synthetic as syn


def identity_camera(W, H, fovx=0.8):
    focal = W / (2.0 * math.tan(fovx / 2))
    fovy = 2.0 * math.atan(H / (2.0 * focal))
    vm = np.eye(4, dtype=np.float32)
    proj = syn.projection_matrix(0.01, 100.0, fovx, fovy)
    return {"W": W, "H": H, "tanfovx": math.tan(fovx / 2), "tanfovy": math.tan(fovy / 2),
            "viewmatrix": vm, "projmatrix": (vm @ proj.T).astype(np.float32), "campos": np.zeros(3, np.float32),
            "fx": focal, "fy": H / (2.0 * math.tan(fovy / 2))}


def base_inputs(cam, xyz, scales, opac, colors, bg=(0.2, 0.3, 0.4)):
    P = len(xyz)
    q = np.tile(np.array([[1.0, 0, 0, 0]], np.float32), (P, 1))
    return {"means3D": np.asarray(xyz, np.float32), "scales": np.asarray(scales, np.float32), "rotations": q,
            "opacities": np.asarray(opac, np.float32).reshape(P, 1), "colors_precomp": np.asarray(colors, np.float32),
            "bg": np.asarray(bg, np.float32), "W": cam["W"], "H": cam["H"], "tanfovx": cam["tanfovx"], "tanfovy": cam["tanfovy"],
            "viewmatrix": cam["viewmatrix"], "projmatrix": cam["projmatrix"], "campos": cam["campos"], "sh_degree": 0}


def test_single_isotropic_gaussian_analytic():
    W = H = 32
    cam = identity_camera(W, H)
    z0, s, o = 2.0, 0.05, 0.8
    col = np.array([[0.9, 0.5, 0.1]], np.float32)
    inp = base_inputs(cam, [[0, 0, z0]], [[s, s, s]], [o], col)
    f = oracle.forward(inp)
    sig2 = (cam["fx"] * s / z0) ** 2 + 0.3                      # EWA + 0.3 low-pass, forward.cu:148-149
    assert f["radii"][0] == math.ceil(3 * math.sqrt(sig2))
    np.testing.assert_allclose(f["means2D"][0], [(W - 1) / 2, (H - 1) / 2], atol=1e-4)
    assert f["num_rendered"] == f["tiles_touched"][0] == 4      # centre of a 2x2 tile grid, radius < 16
    ys, xs = np.mgrid[0:H, 0:W]
    r2 = (xs - (W - 1) / 2) ** 2 + (ys - (H - 1) / 2) ** 2
    alpha = np.minimum(0.99, o * np.exp(-0.5 * r2 / sig2))
    alpha = np.where(alpha < 1 / 255, 0.0, alpha)
    bg = inp["bg"]
    want = col[0][:, None, None] * alpha[None] + bg[:, None, None] * (1 - alpha)[None]
    np.testing.assert_allclose(f["color"], want, atol=2e-6)
    np.testing.assert_allclose(f["final_T"].reshape(H, W), 1 - alpha, atol=1e-6)
    assert set(np.unique(f["n_contrib"])) <= {0, 1}


def test_two_layers_blend_order_and_state():
    W = H = 16
    cam = identity_camera(W, H)
    inp = base_inputs(cam, [[0, 0, 3.0], [0, 0, 2.0]], [[0.4] * 3, [0.4] * 3], [0.5, 0.6],
                      [[1, 0, 0], [0, 1, 0]], bg=(0, 0, 1))
    f = oracle.forward(inp)
    # the nearer Gaussian (index 1) is blended first although it comes second in memory
    assert list(f["point_list"]) == [1, 0]
    cy = cx = 8   # any pixel: both Gaussians are huge on screen, alpha ~ opacity near the centre
    sig2 = [(cam["fx"] * 0.4 / z) ** 2 + 0.3 for z in (2.0, 3.0)]
    r2 = (cx - 7.5) ** 2 + (cy - 7.5) ** 2
    a_near = 0.6 * math.exp(-0.5 * r2 / sig2[0]); a_far = 0.5 * math.exp(-0.5 * r2 / sig2[1])
    want = np.array([a_far * (1 - a_near), a_near, (1 - a_near) * (1 - a_far)])
    np.testing.assert_allclose(f["color"][:, cy, cx], want, atol=2e-6)
    assert f["n_contrib"][cy * W + cx] == 2
    np.testing.assert_allclose(f["final_T"][cy * W + cx], (1 - a_near) * (1 - a_far), atol=1e-6)


def test_termination_rule_q7():
    """The Gaussian that would push T below 1e-4 is not blended and not counted (forward.cu:427, 491-492)."""
    W = H = 16
    cam = identity_camera(W, H)
    n = 6
    xyz = [[0, 0, 2.0 + 0.1 * k] for k in range(n)]
    inp = base_inputs(cam, xyz, [[0.5] * 3] * n, [1.0] * n, [[1, 1, 1]] * n, bg=(0, 0, 0))
    f = oracle.forward(inp)
    pix = 8 * W + 8
    T = np.float32(1.0); count = 0
    for k in range(n):
        sig2 = (cam["fx"] * 0.5 / (2.0 + 0.1 * k)) ** 2 + 0.3
        a = np.float32(min(0.99, math.exp(-0.5 * 0.5 / sig2)))   # opacity 1, pixel (8,8) is 0.5 px from the centre in x and y
        t2 = np.float32(T * (np.float32(1.0) - a))
        if t2 < np.float32(0.0001):
            break
        T = t2; count += 1
    assert f["n_contrib"][pix] == count
    np.testing.assert_allclose(f["final_T"][pix], T, rtol=1e-6)
    assert count < n


def plane_scene(L=4, n_src=2, W=48, H=32, z0=3.0):
    cam = identity_camera(W, H)
    gx, gy = np.meshgrid(np.linspace(-1.2, 1.2, 25), np.linspace(-0.8, 0.8, 17))
    xyz = np.stack([gx.ravel(), gy.ravel(), np.full(gx.size, z0)], 1).astype(np.float32)
    P = xyz.shape[0]
    inp = base_inputs(cam, xyz, [[0.12, 0.12, 0.01]] * P, [0.7] * P, np.full((P, 3), 0.5), bg=(0, 0, 0))
    am = np.zeros((P, 5), np.float32)
    am[:, 2] = -1.0; am[:, 3] = 1.0; am[:, 4] = z0          # plane z = z0 facing the camera: depth = dist/(+1)
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.stack([xs / W, ys / H, 0.25 + 0 * xs])[None].repeat(n_src, 0).astype(np.float32)
    eye = np.eye(4, dtype=np.float32).reshape(1, 16).repeat(n_src, 0)
    inp.update(all_map=am, render_geo=True, n_src=n_src, buffer_length=L, depth_thr=0.01,
               ref_to_src=eye, src_cam_pos=np.zeros((n_src, 3), np.float32), src_images=img,
               src_depths=np.full((n_src, 1, H, W), z0, np.float32))
    return inp, img


def test_fronto_parallel_plane_geo_outputs():
    for L in (4, 5):
        inp, img = plane_scene(L=L)
        H, W = inp["H"], inp["W"]
        f = oracle.forward(inp)
        hit = f["cache_sum_w"].reshape(H, W) > 1e-3
        assert hit.mean() > 0.9
        np.testing.assert_allclose(f["median_depth"][0][hit], 3.0, rtol=1e-5)
        # border pixels re-project to u = 0 -/+ rounding and may fall outside the in-bounds gate
        hit[0, :] = hit[-1, :] = False; hit[:, 0] = hit[:, -1] = False
        # identity ref->src: the median point re-projects onto its own pixel, so the warp returns the source image
        for k in range(2):
            np.testing.assert_allclose(f["warped_image"][3 * k:3 * k + 3][:, hit], img[k][:, hit], atol=2e-4)
            np.testing.assert_allclose(f["cam_feat"][4 * k:4 * k + 3][:, hit], 0.0, atol=1e-7)
            np.testing.assert_allclose(f["cam_feat"][4 * k + 3][hit], 1.0, atol=1e-5)
        assert np.all(f["cam_feat"][8:] == 0) and np.all(f["warped_image"][6:] == 0)     # slots >= n valid stay zero
        np.testing.assert_allclose(f["min_depth_diff"][0][hit], 0.0, atol=1e-5)
        assert np.all(f["use_first_src_frame_mask"][0][hit] == 1)
        vi = f["valid_src_idx"].reshape(5, H, W)
        assert np.all(vi[0][hit] == 0) and np.all(vi[1][hit] == 1) and np.all(vi[2][hit] == -1)
        # normal map is the un-normalised sum n * alpha * T: z component = -(1 - T_final)
        np.testing.assert_allclose(f["normal_map"][2], -(1 - f["final_T"].reshape(H, W)), atol=1e-5)
        # camera ray = normalised pixel ray (camera at the origin, identity rotation)
        cx, cy = W / 2, H / 2
        fx = W / (2 * inp["tanfovx"]); fy = H / (2 * inp["tanfovy"])
        ys, xs = np.mgrid[0:H, 0:W]
        ray = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones_like(xs, float)])
        ray /= np.linalg.norm(ray, axis=0, keepdims=True)
        np.testing.assert_allclose(f["camera_ray"][:, hit], ray[:, hit], atol=1e-5)


def test_depth_only_matches_geo_median_when_buffer_not_evicted():
    inp, _ = plane_scene(L=4)
    g = oracle.forward(inp)
    d = dict(inp); d.update(render_geo=False, render_depth_only=True)
    f = oracle.forward(d)
    H, W = inp["H"], inp["W"]
    hit = g["cache_sum_w"].reshape(H, W) > 1e-3
    np.testing.assert_allclose(f["median_depth"][0][hit], 3.0, rtol=1e-5)
    assert np.all(f["color"] == 0)                                       # depth-only skips colour


def test_invalid_source_depth_gives_no_slots():
    inp, _ = plane_scene()
    inp["src_depths"] = np.zeros_like(inp["src_depths"])                 # "zeros until first written"
    f = oracle.forward(inp)
    assert np.all(f["warped_image"] == 0) and np.all(f["cam_feat"] == 0)
    assert np.all(f["use_first_src_frame_mask"] == 0)
    assert np.all(f["min_depth_diff"] == 1.0)
    assert np.all(f["valid_src_idx"][0] == -1)


def test_sort_is_stable_on_depth_ties_q9():
    W = H = 16
    cam = identity_camera(W, H)
    xyz = [[0.0, 0, 2.0]] * 5 + [[0.0, 0, 1.5]]
    inp = base_inputs(cam, xyz, [[0.3] * 3] * 6, [0.3] * 6, np.eye(3).tolist() * 2)
    f = oracle.forward(inp)
    assert list(f["point_list"]) == [5, 0, 1, 2, 3, 4]


def test_empty_and_culled_inputs():
    W = H = 32
    cam = identity_camera(W, H)
    inp = base_inputs(cam, np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0,)), np.zeros((0, 3)))
    f = oracle.forward(inp)
    assert f["num_rendered"] == 0 and np.all(f["color"] == 0)           # P == 0: outputs stay zero
    inp = base_inputs(cam, [[0, 0, 0.1], [0, 0, -3.0], [50.0, 0, 1.0]], [[0.1] * 3] * 3, [0.9] * 3, [[1, 1, 1]] * 3)
    f = oracle.forward(inp)
    assert list(f["radii"]) == [0, 0, 0] and f["num_rendered"] == 0      # near plane / behind / off screen
    np.testing.assert_allclose(f["color"], np.broadcast_to(inp["bg"][:, None, None], (3, H, W)))
    b = oracle.backward(inp, f, np.ones((3, H, W), np.float32))
    assert all(np.all(v == 0) for v in b.values())
