"""Upper bound of what a per-tile depth bound in the front end could save, measured without building it: render a bench workload
once, mark every Gaussian that some tile's walk really REACHED (its position in that tile's list < how far the forward walked the list), drop the
visible Gaussians nobody reached from the inputs, and time the same step on what is left.  An exact bound can cull no more than that; a real one would
still run the geometry kernel over the culled Gaussians (added back below from the measured per-Gaussian cost).
usage: python tools/occlusion_potential.py [init|trained|trained_geo|geo] ..."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
modes = [a for a in sys.argv[1:]] or ["init", "trained_geo"]
sys.argv = sys.argv[:1]
import bench
from ibgs_amd import _lib, synthetic as syn
from tests import hipref

dev = torch.device("cuda", 0)
lib = _lib.load()


def stages(wl, n=5):
    for _ in range(8):
        wl.local_step()
    wall = bench.timed_wall_ms(wl.local_step, 20, warmup=3)
    _lib.timing_enable(_lib.STAGES)
    for _ in range(n):
        wl.local_step()
    torch.cuda.synchronize()
    st = {k: v[0] / n for k, v in _lib.timing_collect().items()}
    _lib.timing_enable([])
    return wall, st


for mode in modes:
    geo = "geo" in mode
    kw = dict(cluster=0.3, anisotropy="plane", scale_sigma=1.0) if mode.startswith("trained_geo") else {}
    opacity = "init" if mode in ("init", "geo") else "trained"
    wl = bench.Workload("C3", 0, dev, opacity, geo, False, 1234, **kw)
    wall0, st0 = stages(wl)
    for v in wl.leaves.values():
        v.grad = None
    outs = wl._call()
    inp = {"means3D": wl.inp["means3D"], "W": wl.W, "H": wl.H}
    ist = hipref.internal_state({"color": outs[0]}, inp)
    img = outs[0].grad_fn.saved_tensors[-1]
    tiles = ist["ranges"].shape[0]
    off, moff = lib.ibgs_img_offset(wl.W, wl.H, b"tile_walked"), lib.ibgs_img_offset(wl.W, wl.H, b"meta")
    ipt = int(img[moff:moff + 128].view(torch.int32)[10].item())
    walked = img[off:off + tiles * ipt * 4].view(torch.int32).view(tiles, ipt).max(dim=1).values.cpu().numpy().astype(np.int64)
    rg = ist["ranges"].astype(np.int64)
    n_list = rg[:, 1] - rg[:, 0]
    reach = np.minimum(walked + 64, n_list)          # the forward stages 64 (colour) / 16 (geo) entries per round beyond the last contributor: count a full round as reached
    needed = np.zeros(wl.P, bool)
    pl = ist["point_list"]
    for t in np.flatnonzero(reach > 0):
        needed[pl[rg[t, 0]:rg[t, 0] + reach[t]]] = True
    radii = outs[1].cpu().numpy()
    vis = radii > 0
    culled = vis & ~needed
    sat = (walked < n_list).mean()
    print("\n%s: P %d, visible %d, reached by some tile %d, visible but never reached %d (%.1f %% of the visible); R %d, entries walked %d (%.1f %%); tiles that stop before their list ends %.1f %%"
          % (mode, wl.P, vis.sum(), needed.sum(), culled.sum(), 100.0 * culled.sum() / max(vis.sum(), 1), ist["R"], walked.sum(), 100.0 * walked.sum() / max(ist["R"], 1), 100 * sat))
    keep = torch.as_tensor(~culled, device=dev)
    col0 = outs[0].detach().clone()
    # the same workload without the never-reached Gaussians
    for k in ("means3D", "shs", "opacities", "scales", "rotations", "means2D", "means2D_abs", "all_map"):
        if k in wl.leaves:
            wl.leaves[k] = wl.leaves[k].detach()[keep].contiguous().requires_grad_(True)
    wl.params = [wl.leaves[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    wl.P = int(keep.sum().item())
    wall1, st1 = stages(wl)
    with torch.no_grad():
        col1 = wl._call()[0]
    print("   image after the cull: max |d| %.3e (must be 0: nobody reached what went)" % float((col1 - col0).abs().max()))
    geom_per_g = st0["preprocess"] * 0.5 / wl.c["P"]          # ~half of the preprocess stage is the geometry kernel, which a real bound would still run for every Gaussian
    back = geom_per_g * culled.sum()
    print("   step wall %.3f -> %.3f ms (+ %.3f ms for the geometry kernel over the culled ones = %.3f: upper bound of the saving %.3f ms = %.1f %%)"
          % (wall0, wall1, back, wall1 + back, wall0 - wall1 - back, 100.0 * (wall0 - wall1 - back) / wall0))
    print("   stages before: " + " ".join("%s %.3f" % (k[:10], v) for k, v in st0.items() if v > 0))
    print("   stages after:  " + " ".join("%s %.3f" % (k[:10], v) for k, v in st1.items() if v > 0))
    del wl, outs
    torch.cuda.empty_cache()
