"""A/B of the Python layer's host time on ONE box: ibgs_amd/rasterizer.py against an older copy (ibgs_amd/_exp/rasterizer_old.py, `git show <rev>:ibgs_amd/rasterizer.py`).
A tiny workload (200 Gaussians, 64 x 64: the GPU needs a few tens of microseconds) so that what is timed is the host: forward call and backward call, GPU drained
before each.  usage: python tools/host_ab.py"""
import importlib.util, os, statistics, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ibgs_amd import rasterizer as new, synthetic as syn
mods = {"current": new}
old_path = os.path.join(ROOT, "ibgs_amd", "_exp", "rasterizer_old.py")
if os.path.exists(old_path):
    spec = importlib.util.spec_from_file_location("rasterizer_old", old_path)
    old = importlib.util.module_from_spec(spec); spec.loader.exec_module(old)
    mods["old"] = old
dev = torch.device("cuda", 0)
inp = syn.make_scene(200, 64, 64, sh_degree=3, seed=1, opacity="trained")
t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
P = 200
lv = {k: t(inp[k]).requires_grad_(True) for k in ("means3D", "shs", "scales", "rotations")}
lv["opacities"] = t(inp["opacities"]).reshape(P, 1).requires_grad_(True)
m2 = torch.zeros(P, 3, device=dev, requires_grad=True); m2a = torch.zeros(P, 3, device=dev, requires_grad=True)
z = lambda *s: torch.zeros(*s, device=dev)
g = torch.randn(3, 64, 64, device=dev)
for rep in range(2):
    for name, mod in mods.items():
        st = mod.GaussianRasterizationSettings(image_height=64, image_width=64, tanfovx=float(inp["tanfovx"]), tanfovy=float(inp["tanfovy"]), bg=t(inp["bg"]), scale_modifier=1.0,
                                               viewmatrix=t(inp["viewmatrix"]), projmatrix=t(inp["projmatrix"]), ref_to_src_list=z(1, 16), src_cam_pos=z(1, 3), src_images=z(1, 3, 1),
                                               src_rendered_depths=z(1, 1, 1), nb_src_images=1, buffer_length=4, depth_error_threshold=0.01, sh_degree=3, campos=t(inp["campos"]),
                                               prefiltered=False, render_geo=False, render_depth_only=False, debug=False)
        rast = mod.GaussianRasterizer(st)
        tf, tb = [], []
        for it in range(400):
            for v in lv.values():
                v.grad = None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = rast(means3D=lv["means3D"], means2D=m2, means2D_abs=m2a, opacities=lv["opacities"], shs=lv["shs"], scales=lv["scales"], rotations=lv["rotations"])
            t1 = time.perf_counter()
            loss = (out[0] * g).sum()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            loss.backward()
            t3 = time.perf_counter()
            if it >= 50:
                tf.append((t1 - t0) * 1e6); tb.append((t3 - t2) * 1e6)
        print("%-8s forward call %.1f us (min %.1f), backward call %.1f us (min %.1f)  [medians of 350, GPU drained before each call, 200 Gaussians 64 x 64]"
              % (name, statistics.median(tf), min(tf), statistics.median(tb), min(tb)))
