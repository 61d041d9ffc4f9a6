"""Host-side mirror of the reference operator: names, field order, validation, glue maths (CPU only)."""
import math

import numpy as np
import pytest
import torch

import diff_plane_rasterization as dpr
from ibgs_amd import rasterizer, renderer, simple_scene, synthetic as syn


def test_settings_fields_match_reference_order():
    # reference DPR/diff_plane_rasterization/__init__.py:252-276
    want = ["image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
            "ref_to_src_list", "src_cam_pos", "src_images", "src_rendered_depths", "nb_src_images", "buffer_length",
            "depth_error_threshold", "sh_degree", "campos", "prefiltered", "render_geo", "render_depth_only", "debug"]
    fields = list(dpr.GaussianRasterizationSettings._fields)
    assert fields[:len(want)] == want
    # ... followed only by this library's extensions, every one with a default: the reference's own 21-argument construction keeps working (`_settings()` below)
    assert fields[len(want):] == ["src_depth_slots"] and set(fields[len(want):]) <= set(dpr.GaussianRasterizationSettings._field_defaults)
    assert dpr.GaussianRasterizer is rasterizer.GaussianRasterizer
    assert hasattr(dpr, "_C") and hasattr(dpr._C, "rasterize_gaussians") and hasattr(dpr._C, "mark_visible")


def _settings():
    z = torch.zeros
    return dpr.GaussianRasterizationSettings(32, 32, 0.5, 0.5, z(3), 1.0, torch.eye(4), torch.eye(4), z(1, 16), z(1, 3),
                                             z(1, 3, 1024), z(1, 1, 1024), 1, 4, 0.01, 0, z(3), False, False, False, False)


def test_argument_validation_messages():
    r = dpr.GaussianRasterizer(_settings())
    m = torch.zeros(4, 3)
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, m, torch.zeros(4, 1), scales=m, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(m, m, m, torch.zeros(4, 1), shs=torch.zeros(4, 1, 3), colors_precomp=m, scales=m, rotations=torch.zeros(4, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m, m, m, torch.zeros(4, 1), colors_precomp=m)
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        r(m, m, m, torch.zeros(4, 1), colors_precomp=m, scales=m, rotations=torch.zeros(4, 4), cov3D_precomp=torch.zeros(4, 6))


def test_cpu_tensors_are_rejected_not_silently_computed():
    # the product path has no CPU fallback: a CPU means3D must raise, never route to the oracle
    r = dpr.GaussianRasterizer(_settings())
    m = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="HIP device"):
        r(m, m, m, torch.zeros(4, 1), colors_precomp=m, scales=m, rotations=torch.zeros(4, 4))
    with pytest.raises(RuntimeError, match=r"\(num_points, 3\)"):
        r(torch.zeros(4, 2), m, m, torch.zeros(4, 1), colors_precomp=m, scales=m, rotations=torch.zeros(4, 4))


def test_product_path_never_imports_the_oracle():
    import os, re
    root = os.path.dirname(os.path.abspath(rasterizer.__file__))
    for fn in os.listdir(root):
        if fn.endswith(".py"):
            src = open(os.path.join(root, fn)).read()
            assert not re.search(r"^\s*(import|from)\s+oracle\b", src, re.M), fn + " imports the oracle"


def test_plane_map_matches_numpy_construction():
    g = syn.make_gaussians(200, seed=3)
    cam = syn.make_camera(64, 48, azimuth_deg=70)
    pc = simple_scene.SimpleGaussians(g)
    vc = simple_scene.SimpleCamera(cam)
    with torch.no_grad():
        am = renderer._plane_map(pc, vc, False, pc.get_xyz).numpy()
    want = syn.plane_all_map(g["means3D"], g["scales"], g["rotations"], cam)
    np.testing.assert_allclose(am, want, atol=2e-5)
    assert np.all(am[:, 3] == 1.0) and np.all(am[:, 4] >= 0)
    # learnt normal + offset: flipped normal flips the offset sign (scene/gaussian_model.py:166-173)
    rng = np.random.default_rng(0)
    g["normal"] = rng.normal(size=(200, 3)).astype(np.float32); g["offset"] = rng.normal(size=(200, 1)).astype(np.float32)
    pc = simple_scene.SimpleGaussians(g)
    with torch.no_grad():
        am = renderer._plane_map(pc, vc, True, pc.get_xyz).numpy()
    want = syn.plane_all_map(g["means3D"], g["scales"], g["rotations"], cam, normal=g["normal"], offset=g["offset"])
    np.testing.assert_allclose(am, want, atol=2e-5)


def test_ref_to_src_convention():
    cams = simple_scene.orbit_cameras(64, 48, n_views=4)
    sc = simple_scene.SimpleScene(cams)
    ref = cams[0]
    chosen = [1, 2]
    world_to_src = sc.world_view_transforms[chosen]
    ref_to_world = ref.world_view_transform.T.inverse()
    r2s = (world_to_src @ ref_to_world.unsqueeze(0)).numpy()
    want, pos = syn.ref_to_src({"viewmatrix": ref.world_view_transform.numpy()},
                               [{"viewmatrix": cams[i].world_view_transform.numpy()} for i in chosen])
    np.testing.assert_allclose(r2s.reshape(2, 16), want, atol=1e-5)
    np.testing.assert_allclose(torch.inverse(world_to_src)[:, :3, 3].numpy(), pos, atol=1e-5)
    np.testing.assert_allclose(pos[0], cams[1].camera_center.numpy(), atol=1e-5)


def test_camera_objects():
    cams = simple_scene.orbit_cameras(80, 60, n_views=8)
    assert all(len(c.nearest_id) == 4 and c.uid not in c.nearest_id for c in cams)
    K, E = cams[0].get_calib_matrix_nerf()
    assert abs(float(K[0, 0]) - 80 / (2 * math.tan(cams[0].FoVx / 2))) < 1e-3
    np.testing.assert_allclose(E.numpy(), cams[0].world_view_transform.numpy().T)
