"""The depth-bound hint (include/ibgs_rast.h: depth_bound_hint) on the bench workloads, A/B: step time and stage times with rasterizer.DEPTH_BOUND off / on, how much
of the lists and of the depth sort's input the bound removed, how many tiles carry a finite bound, whether the repair pass ever ran.
usage: python tools/depth_bound_ab.py [init|trained|trained_geo|geo] ..."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
modes = [a for a in sys.argv[1:]] or ["init", "trained", "trained_geo"]
sys.argv = sys.argv[:1]
import bench
from ibgs_amd import _lib, rasterizer
from tests import hipref

dev = torch.device("cuda", 0)
lib = _lib.load()


def stages(wl, n=5):
    for _ in range(8):
        wl.local_step()
    wall = bench.timed_wall_ms(wl.local_step, 30, warmup=3)
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
    res = {}
    for on in (False, True, False, True):
        rasterizer.DEPTH_BOUND = on
        rasterizer._bound_hints.clear()
        wl = bench.Workload("C3", 0, dev, opacity, geo, False, 1234, **kw)
        wall, st = stages(wl)
        for v in wl.leaves.values():
            v.grad = None
        outs = wl._call()
        torch.cuda.synchronize()
        ist = hipref.internal_state({"color": outs[0]}, {"means3D": wl.inp["means3D"], "W": wl.W, "H": wl.H})
        img = outs[0].grad_fn.saved_tensors[-1]
        moff = lib.ibgs_img_offset(wl.W, wl.H, b"meta")
        meta = img[moff:moff + 128].view(torch.int32).cpu().numpy()
        line = "%-12s bound %-3s  step %.3f ms  | %s | R %d, listed %d (%.1f %%), sorted Gaussians %d" % (
            mode, "on" if on else "off", wall, " ".join("%s %.3f" % (k, v) for k, v in st.items() if v > 0.0005), int(outs[0].grad_fn.num_rendered), ist["R"],
            100.0 * ist["R"] / max(int(outs[0].grad_fn.num_rendered), 1), len(ist["order"]))
        if on and rasterizer._bound_hints:
            b = next(iter(rasterizer._bound_hints.values())).cpu().numpy()
            line += " | tiles with a bound %.1f %%, repair %d (%d tiles)" % (100.0 * np.isfinite(b).mean(), meta[12], meta[13])
        print(line, flush=True)
        del wl, outs
