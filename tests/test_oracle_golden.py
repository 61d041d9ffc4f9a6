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
