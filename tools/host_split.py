"""Where the host time of one rasterizer call pair goes (round 5): the tiny workload of tools/host_ab.py (200 Gaussians, 64 x 64 -- the GPU needs a few
tens of microseconds, the host is what is timed), GPU drained before each call.  Splits each call into the time inside the C library
(`ibgs_forward` / `ibgs_backward`: argument checks + HIP launches + the R read-back) and the Python / torch layer around it (tensor allocation,
argument marshalling, autograd).  usage: python tools/host_split.py [--geo] [--graph]"""
import os, statistics, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ibgs_amd import _lib, rasterizer as rz, synthetic as syn
from tests import hipref
from tests.scenes import add_sources, scene as mk

GEO = "--geo" in sys.argv
if "--graph" in sys.argv:
    rz.GRAPHS = True
dev = torch.device("cuda", 0)
inp = mk(P=200, W=64, H=64, deg=3, seed=1, opacity="trained", planes=GEO)
if GEO:
    inp = add_sources(inp, n_src=3, L=4)
st = hipref.settings_from(inp, dev)
lv = hipref.leaf_inputs(inp, dev)
rast = rz.GaussianRasterizer(st)
lib = _lib.load()
acc = {"fwd": [], "bwd": []}
real_f, real_b = lib.ibgs_forward, lib.ibgs_backward


class Timed:
    def __init__(self, fn, key):
        self.fn, self.key = fn, key

    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.fn(*a); acc[self.key].append((time.perf_counter() - t0) * 1e6)
        return r


names = ("color", "normal_map", "median_depth", "warped_image")
g = {"color": torch.randn(3, 64, 64, device=dev), "normal_map": torch.randn(3, 64, 64, device=dev), "median_depth": torch.randn(1, 64, 64, device=dev),
     "warped_image": torch.randn(15, 64, 64, device=dev)}
# layer by layer: time inside _C.rasterize_gaussians / _backward (marshalling + the C call), inside the autograd Function's forward / backward (the former + ctx
# bookkeeping), and the call as the trainer sees it (the latter + torch's dispatch: Module.__call__, Function.apply, the autograd engine's thread hop)
layers = {"C.fwd": [], "C.bwd": [], "Fn.fwd": [], "Fn.bwd": []}


def timed_static(cls, name, key):
    fn = getattr(cls, name)

    def wrapper(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); layers[key].append((time.perf_counter() - t0) * 1e6)
        return r
    setattr(cls, name, staticmethod(wrapper))
    return fn


for timed in (False, True):
    if timed:
        lib.ibgs_forward, lib.ibgs_backward = Timed(real_f, "fwd"), Timed(real_b, "bwd")
        timed_static(rz._CModule, "rasterize_gaussians", "C.fwd"); timed_static(rz._CModule, "rasterize_gaussians_backward", "C.bwd")
        timed_static(rz._RasterizeGaussians, "forward", "Fn.fwd"); timed_static(rz._RasterizeGaussians, "backward", "Fn.bwd")
    tf, tb = [], []
    for it in range(400):
        for v in lv.values():
            if v is not None:
                v.grad = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = rast(means3D=lv["means3D"], means2D=lv["means2D"], means2D_abs=lv["means2D_abs"], opacities=lv["opacities"], shs=lv["shs"], scales=lv["scales"],
                   rotations=lv["rotations"], all_map=lv["all_map"])
        t1 = time.perf_counter()
        loss = (out[0] * g["color"]).sum()
        if GEO:
            loss = loss + (out[2] * g["normal_map"]).sum() + (out[3] * g["median_depth"]).sum() + (out[5] * g["warped_image"]).sum()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        loss.backward()
        t3 = time.perf_counter()
        if it >= 50:
            tf.append((t1 - t0) * 1e6); tb.append((t3 - t2) * 1e6)
    med = statistics.median
    if not timed:
        print("%s: forward call %.1f us (min %.1f), backward call %.1f us (min %.1f), pair %.1f us  [medians of 350, GPU drained before each call]"
              % ("geo" if GEO else "colour", med(tf), min(tf), med(tb), min(tb), med(tf) + med(tb)))
    else:
        cf, cb = med(acc["fwd"][50:]), med(acc["bwd"][50:])
        print("   inside the C library: ibgs_forward %.1f us, ibgs_backward %.1f us; Python / torch / autograd around them: forward %.1f us, backward %.1f us"
              % (cf, cb, med(tf) - cf, med(tb) - cb))
        m = {k: med(v[50:]) for k, v in layers.items()}
        print("   layers (medians, us): forward  call %.1f > Function.forward %.1f > _C.rasterize_gaussians %.1f > ibgs_forward %.1f" % (med(tf), m["Fn.fwd"], m["C.fwd"], cf))
        print("                         backward call %.1f > Function.backward %.1f > _C.rasterize_gaussians_backward %.1f > ibgs_backward %.1f   (the call includes the loss's own backward)"
              % (med(tb), m["Fn.bwd"], m["C.bwd"], cb))
lib.ibgs_forward, lib.ibgs_backward = real_f, real_b
