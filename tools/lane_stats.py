"""How full are the lanes of the colour blend kernels?    Diagnostic build of the library with -DIBGS_COUNT_LANES:
every wave counts the list entries it walks, the entries that at least one of its pixels blends, the lanes that EXECUTE a (pixel, Gaussian)
evaluation (64 per quadrant evaluation that is not skipped as a whole) and the lanes whose pixel really blends.  One forward + backward of
the C3-sized scene per workload.   usage: python tools/lane_stats.py [init] [trained] [trained_scene]"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ibgs_amd import _build

lib = "/tmp/libibgs_lanes.so"
objs = []
for name in _build.SOURCES:
    o = os.path.join(_build.OBJ, name + ".o")
    if name in ("render_fwd", "render_bwd"):
        o = "/tmp/lanes_%s.o" % name
        subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DIBGS_COUNT_LANES", "-c",
                               os.path.join(_build.CSRC, name + ".hip"), "-o", o] + _build.EXTRA.get(name, []))
    objs.append(o)
subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)

import torch
from ibgs_amd import _lib
_lib.LIB_PATH = lib
from ibgs_amd import synthetic as syn
from tests import hipref

raw = ctypes.CDLL(lib)
c = syn.CONFIGS["C3"]
cases = {"init": dict(opacity="init"), "trained": dict(opacity="trained"),
         "trained_scene": dict(opacity="trained", anisotropy="plane", scale_sigma=1.0, cluster=0.3)}
for name in ([a for a in sys.argv[1:] if a in cases] or list(cases)):
    inp = syn.make_scene(c["P"], c["W"], c["H"], sh_degree=3, seed=c["seed"], **cases[name])
    buf = (ctypes.c_ulonglong * 4)()
    raw.ibgs_debug_lanes_fwd(buf, 1); raw.ibgs_debug_lanes_bwd(buf, 1)
    outs, lv, _ = hipref.run_forward(inp)
    (outs["color"] * torch.randn(3, c["H"], c["W"], device="cuda")).sum().backward()
    torch.cuda.synchronize()
    R = int(outs["color"].grad_fn.num_rendered)
    for which in ("fwd", "bwd"):
        getattr(raw, "ibgs_debug_lanes_" + which)(buf, 1)
        walked, proc, ex, useful = [int(x) for x in buf]
        print("C3 %-13s %s: R %.2f M, walked %.2f M entries (%.1f %% of the lists), %.2f M of them blended by some pixel (%.1f %%); quadrant evaluations %.2f M "
              "(%.2f per blended entry of 4); lanes executing %.1f M, useful %.1f M = %.1f %% (useful pairs per walked entry: %.1f of 256)"
              % (name, which, R / 1e6, walked / 1e6, 100.0 * walked / max(R, 1), proc / 1e6, 100.0 * proc / max(walked, 1), ex / 64e6,
                 ex / 64.0 / max(proc, 1), ex / 1e6, useful / 1e6, 100.0 * useful / max(ex, 1), useful / max(walked, 1)))
    del outs, lv
    torch.cuda.empty_cache()
