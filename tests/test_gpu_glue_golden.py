"""The plane map built INSIDE the preprocess kernels (plane_mode 1 = learnt normals + offsets, 2 = smallest-scale axis) against the
reference's own (P, 5) `all_map` -- tests/golden/glue.npz, produced by the reference's `render()` (gaussian_renderer/__init__.py:304-316 on
scene/gaussian_model.py:156-173) in the build container.  The kernels leave the plane of every Gaussian that has tiles in its 64-byte record
(ibgs_amd/csrc/common.h: n_cam at R_NX.., |d_cam| at R_DIST); it must be the reference's to fp32 rounding, flips included."""
import numpy as np
import pytest
import torch

from ibgs_amd import renderer
from tests import hipref
from tests.golden_glue import Glue

pytestmark = pytest.mark.gpu
G = Glue()


@pytest.mark.parametrize("name,learnt", [("geo_learnt", True), ("geo_axis_two_sources_app", False)])
def test_fused_plane_map_is_the_reference_s_all_map(name, learnt, monkeypatch):
    dev = torch.device("cuda")
    pc, cams = G.model(dev), G.cameras(dev)
    scene = G.scene(cams, dev)
    pipe, args = G.pipe_args(name)
    kw = G.call_kwargs(name)
    kw.pop("scaling_modifier", None)          # (the plane map does not depend on it; keeps every Gaussian's footprint as large as possible)
    monkeypatch.setattr(renderer, "FUSED_PLANE_MAP", True)
    cam = cams[int(G["case_%s_cam" % name])]
    out = renderer.render(cam, pc, scene, pipe, args, torch.as_tensor(G["bg"], device=dev), **kw)
    assert out["render"].grad_fn is not None
    ist = hipref.internal_state({"color": out["render"]}, {"means3D": G["raw_xyz"], "W": int(G["W"]), "H": int(G["H"])})
    want = G["case_%s_call0_arg_all_map" % name]
    vis = out["radii"].cpu().numpy() > 0
    assert vis.sum() > 0.5 * vis.size, "fixture scene: too few Gaussians on screen"
    rec = ist["rec"]
    n_err = np.abs(rec[vis, 12:15] - want[vis, :3]).max(); d_err = (np.abs(rec[vis, 7] - want[vis, 4]) / np.maximum(1.0, np.abs(want[vis, 4]))).max()
    print("\n[glue golden] %s: %d of %d Gaussians on screen, plane normal max |d| %.2e, plane distance max rel |d| %.2e" % (name, vis.sum(), vis.size, n_err, d_err))
    assert n_err < 2e-6 and d_err < 2e-6
    # ... and the reference's map fed through the plain `all_map` input gives the same render as the fused path
    monkeypatch.setattr(renderer, "FUSED_PLANE_MAP", False)
    monkeypatch.setattr(renderer, "_plane_map", lambda *a, **k: torch.as_tensor(want, device=dev))
    out2 = renderer.render(cam, pc, scene, pipe, args, torch.as_tensor(G["bg"], device=dev), **kw)
    for k in ("render", "rendered_normal", "median_intersected_depth"):
        a, b = out[k].detach().cpu().numpy(), out2[k].detach().cpu().numpy()
        assert np.abs(a - b).mean() <= 1e-5 * (np.abs(b).mean() + 1e-6) + 1e-7, k
