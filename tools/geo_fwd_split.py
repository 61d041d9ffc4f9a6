"""How much of the geo forward is the per-pixel epilogue (warp into n_src sources)?  Forward-only kernel time for n_src = 1..5."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibgs_amd import _lib, synthetic as syn
from ibgs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
dev = torch.device("cuda")
c = syn.CONFIGS["C3"]; W, H, P = c["W"], c["H"], c["P"]
inp = syn.make_scene(P, W, H, sh_degree=3, seed=3)
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
lv = {k: t(inp[k]) for k in ("means3D", "shs", "scales", "rotations")}; lv["opacities"] = t(inp["opacities"]).reshape(P, 1)
z3 = torch.zeros(P, 3, device=dev)
am = t(syn.plane_all_map(inp["means3D"], inp["scales"], inp["rotations"], inp["_cam"]))
for n_src in (1, 2, 4, 5):
    cams = [syn.make_camera(W, H, azimuth_deg=45.0 * k) for k in range(1, n_src + 1)]
    r2s, scp = syn.ref_to_src(inp["_cam"], cams)
    st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=float(inp["tanfovx"]), tanfovy=float(inp["tanfovy"]), bg=torch.zeros(3, device=dev),
        scale_modifier=1.0, viewmatrix=t(inp["viewmatrix"]), projmatrix=t(inp["projmatrix"]), ref_to_src_list=t(r2s), src_cam_pos=t(scp),
        src_images=torch.rand(n_src, 3, H, W, device=dev), src_rendered_depths=torch.rand(n_src, 1, H, W, device=dev) * 4 + 1, nb_src_images=n_src,
        buffer_length=4, depth_error_threshold=0.01, sh_degree=3, campos=t(inp["campos"]), prefiltered=False, render_geo=True, render_depth_only=False, debug=False)
    rast = GaussianRasterizer(st)
    with torch.no_grad():
        for _ in range(3): rast(means3D=lv["means3D"], means2D=z3, means2D_abs=z3, opacities=lv["opacities"], shs=lv["shs"], scales=lv["scales"], rotations=lv["rotations"], all_map=am)
        _lib.timing_enable(["render_fwd"]); _lib.timing_collect()
        for _ in range(10): rast(means3D=lv["means3D"], means2D=z3, means2D_abs=z3, opacities=lv["opacities"], shs=lv["shs"], scales=lv["scales"], rotations=lv["rotations"], all_map=am)
        torch.cuda.synchronize(); tm = _lib.timing_collect(); _lib.timing_enable([])
    print("n_src=%d: geo render_fwd %.3f ms" % (n_src, tm["render_fwd"][0] / tm["render_fwd"][1]))
