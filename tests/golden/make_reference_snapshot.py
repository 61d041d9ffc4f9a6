#!/usr/bin/env python3
"""NOT runnable in this repository's environment: it needs a CUDA machine with the REFERENCE's own rasterizer installed
(`pip install submodules/diff-plane-rasterization` inside a checkout of HoangChuongNguyen/ibgs) and NOT this repository's shim of the
same name on PYTHONPATH.  It closes the one gap DESIGN.md section 5 names: it runs the reference's CUDA operator on the seeded C1
inputs that tests/golden/make_oracle_snapshot.py freezes the oracle on, and writes the same fields to tests/golden/reference_c1.npz.
With that file committed, tests/test_oracle_snapshot.py::test_oracle_against_the_reference_snapshot compares the oracle with the
reference's own output (it is skipped while the file is absent).

Usage (CUDA box, repo root of THIS repository for the input generator, reference rasterizer importable):
    python tests/golden/make_reference_snapshot.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.golden import make_oracle_snapshot as snap  # noqa: E402  (inputs + field list; runs the C oracle for nothing else)


def main():
    import torch
    import diff_plane_rasterization as dpr
    if "ibgs_amd" in (getattr(dpr, "__file__", "") or "") or os.path.dirname(os.path.abspath(dpr.__file__)).startswith(ROOT):
        sys.exit("`diff_plane_rasterization` resolves to this repository's shim; install the reference's package and drop the shim from PYTHONPATH")
    if not torch.cuda.is_available() or torch.version.hip is not None:
        sys.exit("needs the reference's CUDA build")
    inp, g = snap.build()
    dev = "cuda"
    t = lambda a, **k: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev, **k)
    P, H, W = inp["means3D"].shape[0], int(inp["H"]), int(inp["W"])
    st = dpr.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=float(inp["tanfovx"]), tanfovy=float(inp["tanfovy"]), bg=t(inp["bg"]),
        scale_modifier=1.0, viewmatrix=t(inp["viewmatrix"]), projmatrix=t(inp["projmatrix"]),
        ref_to_src_list=torch.zeros(1, 16, device=dev), src_cam_pos=torch.zeros(1, 3, device=dev),
        src_images=torch.zeros(1, 3, H * W, device=dev), src_rendered_depths=torch.zeros(1, 1, H * W, device=dev),
        nb_src_images=1, buffer_length=4, depth_error_threshold=0.01, sh_degree=int(inp["sh_degree"]), campos=t(inp["campos"]),
        prefiltered=False, render_geo=False, render_depth_only=False, debug=False)
    leaves = {k: t(inp[k]).requires_grad_(True) for k in ("means3D", "shs", "scales", "rotations")}
    leaves["opacities"] = t(inp["opacities"]).reshape(P, 1).requires_grad_(True)
    m2d = torch.zeros(P, 3, device=dev, requires_grad=True); m2d_abs = torch.zeros(P, 3, device=dev, requires_grad=True)
    outs = dpr.GaussianRasterizer(st)(means3D=leaves["means3D"], means2D=m2d, means2D_abs=m2d_abs, opacities=leaves["opacities"],
                                      shs=leaves["shs"], colors_precomp=None, scales=leaves["scales"], rotations=leaves["rotations"],
                                      cov3D_precomp=None, all_map=t(inp["all_map"]) if inp.get("all_map") is not None else torch.zeros(P, 5, device=dev))
    color, radii = outs[0], outs[1]
    color.backward(t(g))
    out = {"color_crop": color.detach().cpu().numpy()[(slice(None),) + snap.CROP], "radii_head": radii.cpu().numpy()[:snap.HEAD * 8],
           "color_sum": np.float64(color.detach().double().sum().item()), "color_abs_sum": np.float64(color.detach().abs().double().sum().item())}
    grads = {"dL_dmeans3D": leaves["means3D"].grad, "dL_dsh": leaves["shs"].grad, "dL_dopacity": leaves["opacities"].grad,
             "dL_dscales": leaves["scales"].grad, "dL_drotations": leaves["rotations"].grad, "dL_dmeans2D": m2d.grad}
    for k, v in grads.items():
        a = v.detach().cpu().numpy().reshape(P, -1)
        out[k + "_head"] = a[:snap.HEAD].copy()
        out[k + "_abs_sum"] = np.float64(np.abs(a).astype(np.float64).sum())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "reference_c1.npz"), **out)
    print("wrote tests/golden/reference_c1.npz")


if __name__ == "__main__":
    main()
