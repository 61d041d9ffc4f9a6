"""How the geo forward kernel (render_fwd_kernel<1, 2, 4>: blend loop + median buffer + per-pixel epilogue, forward.cu:303-665) splits into its parts, on the bench's two
geo workloads (C3 init scene; the trainer-shaped `trained_geo` scene), by leaving parts out:

    colour        the colour forward kernel on the same lists (no median buffer, no epilogue): the floor of the blend loop
    geo n_src=k   the real thing with the first k of the workload's 4 sources (real renders of the same Gaussians: most pixels find them valid) -- k = 1..4
    geo, sources invalid   n_src = 4 with all-zero source depths: every source fails the depth test (forward.cu:596-610), so the epilogue does its median depth, its
                  world point and 4 depth fetches per pixel, but warps nothing and gathers no texel
Differences: (sources invalid) - colour = median buffer in the loop + the epilogue's fixed part; per valid source = slope of n_src 1..4 = the warp gathers
(L buffered points x one bilinear RGBA fetch each) + 7 plane stores.  usage: python tools/geo_fwd_split.py [steps]   (MI355X; prints a table, profiles/r06_geo_fwd_split.txt)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sys.argv = sys.argv[:1]
import bench
from ibgs_amd import _lib
from ibgs_amd.rasterizer import GaussianRasterizer

dev = torch.device("cuda", 0)


def fwd_ms(wl, rast):
    lv = wl.leaves
    call = lambda: rast(means3D=lv["means3D"], means2D=lv["means2D"], means2D_abs=lv["means2D_abs"], opacities=lv["opacities"], shs=lv["shs"], scales=lv["scales"],
                        rotations=lv["rotations"], all_map=lv.get("all_map"))
    with torch.no_grad():
        for _ in range(4):
            call()
        torch.cuda.synchronize()
        _lib.timing_enable(["render_fwd"]); _lib.timing_collect()
        for _ in range(steps):
            call()
        torch.cuda.synchronize()
        tm = _lib.timing_collect(); _lib.timing_enable([])
    return tm["render_fwd"][0] / max(tm["render_fwd"][1], 1)


for name, kw in (("C3-geo (init scene)", dict(opacity="init")),
                 ("trained_geo (plane-like, log-normal sizes, 30 % in one blob, trained opacities)", dict(opacity="trained", cluster=0.3, anisotropy="plane", scale_sigma=1.0))):
    wl = bench.Workload("C3", 0, dev, kw.pop("opacity"), True, True, 1234, **kw)
    st = wl.st
    rows = []
    rows.append(("colour kernel on the same lists", fwd_ms(wl, GaussianRasterizer(st._replace(render_geo=False)))))
    zero_dep = torch.zeros_like(st.src_rendered_depths)
    rows.append(("geo, n_src 4, every source invalid (zero source depths)", fwd_ms(wl, GaussianRasterizer(st._replace(src_rendered_depths=zero_dep)))))
    for k in (1, 2, 3, 4):
        sk = st._replace(nb_src_images=k, ref_to_src_list=st.ref_to_src_list[:k].contiguous(), src_cam_pos=st.src_cam_pos[:k].contiguous(),
                         src_images=st.src_images[:k].contiguous(), src_rendered_depths=st.src_rendered_depths[:k].contiguous())
        rows.append(("geo, n_src %d (valid sources)" % k, fwd_ms(wl, GaussianRasterizer(sk))))
    # how many (pixel, source) pairs are valid with the 4 real sources
    with torch.no_grad():
        outs = wl.rast(means3D=wl.leaves["means3D"], means2D=wl.leaves["means2D"], means2D_abs=wl.leaves["means2D_abs"], opacities=wl.leaves["opacities"], shs=wl.leaves["shs"],
                       scales=wl.leaves["scales"], rotations=wl.leaves["rotations"], all_map=wl.leaves.get("all_map"))
        valid = (outs[4].view(5, 4, wl.H, wl.W).abs().sum(1) > 0).float().sum(0).mean().item()          # cam_feat: non-zero rows = valid source slots
    print("\n%s: valid sources per pixel (of 4) %.2f" % (name, valid))
    base = rows[0][1]
    for label, ms in rows:
        print("   %-62s %.3f ms   (+%.3f over the colour kernel)" % (label, ms, ms - base))
    inv, g1, g4 = rows[1][1], rows[2][1], rows[5][1]
    print("   -> blend loop floor %.3f | median buffer + epilogue's fixed part %.3f | per valid source %.3f (n_src 1 -> 4: %.3f -> %.3f)" % (base, inv - base, (g4 - g1) / 3.0, g1, g4))
    del wl
    torch.cuda.empty_cache()
