"""Pins the oracle (and the host-side helpers) against values produced by the reference's own
importable python code (tests/golden/make_golden.py ran it in the build container):
SH polynomial, camera-matrix conventions, depth->normal, parity metric definitions."""
import os

import numpy as np
import torch

import oracle
from ibgs_amd import renderer, synthetic as syn
from tests import metrics

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_sh_polynomial_matches_reference_eval_sh():
    d = np.load(os.path.join(G, "sh_eval.npz"))
    shs = np.ascontiguousarray(d["sh"].transpose(0, 2, 1))      # (N,3,16) -> (N,16,3) rasterizer layout
    for deg in range(4):
        got = oracle.eval_sh(deg, shs, d["dirs"])
        np.testing.assert_allclose(got, d["deg%d" % deg], rtol=0, atol=2e-6)
        mine = renderer.eval_sh(deg, torch.tensor(d["sh"]), torch.tensor(d["dirs"])).numpy()
        np.testing.assert_allclose(mine, d["deg%d" % deg], rtol=0, atol=2e-6)


def test_camera_matrices_match_reference_helpers():
    d = np.load(os.path.join(G, "camera_mats.npz"))
    W, H, fovx = int(d["W"]), int(d["H"]), float(d["fovx"])
    for k in range(8):
        cam = syn.make_camera(W, H, fovx=fovx, azimuth_deg=45.0 * k)
        np.testing.assert_allclose(cam["R"], d["R%d" % k], atol=1e-6)
        np.testing.assert_allclose(cam["viewmatrix"], d["wvt%d" % k], atol=2e-6)
        np.testing.assert_allclose(cam["projmatrix"], d["full%d" % k], atol=2e-5)
        np.testing.assert_allclose(cam["campos"], d["center%d" % k], atol=5e-6)
        assert abs(cam["FoVy"] - float(d["fovy%d" % k])) < 1e-12


def test_projection_convention_w_equals_view_depth():
    # P[3][2] = 1  =>  p_hom.w == z_view; checked through the oracle's own projection
    cam = syn.make_camera(64, 48)
    pts = np.random.default_rng(0).uniform(-1, 1, (50, 3)).astype(np.float32)
    vm, pm = cam["viewmatrix"].reshape(-1), cam["projmatrix"].reshape(-1)
    z = vm[2] * pts[:, 0] + vm[6] * pts[:, 1] + vm[10] * pts[:, 2] + vm[14]
    w = pm[3] * pts[:, 0] + pm[7] * pts[:, 1] + pm[11] * pts[:, 2] + pm[15]
    np.testing.assert_allclose(w, z, rtol=1e-5, atol=1e-5)
    assert np.array_equal(oracle.mark_visible(pts, cam["viewmatrix"]), z > 0.2)


def test_depth_normal_matches_reference():
    d = np.load(os.path.join(G, "depth_normal.npz"))
    n = renderer.normal_from_depth_image(torch.tensor(d["depth"]), torch.tensor(d["K"])).numpy()
    np.testing.assert_allclose(n, d["normal"], atol=1e-5)


def test_metric_definitions():
    d = np.load(os.path.join(G, "metrics.npz"))
    np.testing.assert_allclose(metrics.psnr(d["a"], d["b"]), d["psnr"].reshape(-1), rtol=1e-5)
    assert abs(metrics.l1(d["a"], d["b"]) - float(d["l1"])) < 1e-7
    # SSIM (utils/loss_utils.py:34-64), the third parity metric of the reference's evaluation
    assert abs(metrics.ssim(d["a"], d["b"]) - float(d["ssim"])) < 2e-6
    np.testing.assert_allclose(metrics.ssim(d["a"], d["b"], size_average=False), d["ssim_per_image"], atol=2e-6)


def test_cov3d_matches_reference_covariance_activation():
    """tests/golden/cov3d.npz = build_scaling_rotation / strip_symmetric of the reference (gaussian_model.py:38-42,
    utils/general_utils.py:67-120).  The oracle's computeCov3D (forward.cu:118-153 restated) must give the same six
    numbers for the normalised quaternion, for both scale modifiers; so must the renderer's get_covariance glue."""
    import torch
    import oracle
    from ibgs_amd import simple_scene, synthetic as syn
    g = np.load(os.path.join(G, "cov3d.npz"))
    P = g["scales"].shape[0]
    q = g["quats"] / np.linalg.norm(g["quats"], axis=1, keepdims=True)
    cam = syn.make_camera(64, 48)
    rng = np.random.default_rng(0)
    base = {"means3D": (0.2 * rng.normal(size=(P, 3))).astype(np.float32), "shs": np.zeros((P, 1, 3), np.float32),
            "opacities": np.full(P, 0.5, np.float32), "scales": g["scales"], "rotations": q.astype(np.float32),
            "W": 64, "H": 48, "tanfovx": cam["tanfovx"], "tanfovy": cam["tanfovy"], "viewmatrix": cam["viewmatrix"],
            "projmatrix": cam["projmatrix"], "campos": cam["campos"], "bg": np.zeros(3, np.float32), "sh_degree": 0}
    for mod in (1.0, 0.5):
        inp = dict(base); inp["scale_modifier"] = mod
        out = oracle.forward(inp)
        want = g["cov6_mod%g" % mod]
        assert np.abs(out["cov3D"] - want).max() <= 2e-6 * np.abs(want).max()
    pc = simple_scene.SimpleGaussians({"means3D": base["means3D"], "shs": base["shs"], "opacities": base["opacities"],
                                       "scales": g["scales"], "rotations": g["quats"]}, sh_degree=0)
    got = pc.get_covariance(0.5).detach().numpy()
    assert np.abs(got - g["cov6_mod0.5"]).max() <= 2e-6 * np.abs(g["cov6_mod0.5"]).max()
    assert np.abs(pc.rotation_matrices().detach().numpy() - g["rotmat"]).max() < 2e-6
