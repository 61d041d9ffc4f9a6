"""Quick timing loop for profiling (not a test): python tests/gpu_timing.py [C1|C2|C3] [--geo] [--iters N] [--fwd]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibgs_amd import synthetic as syn  # noqa: E402
from tests import hipref  # noqa: E402


def main():
    cfg = "C3"
    for a in sys.argv[1:]:
        if a in syn.CONFIGS:
            cfg = a
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 5
    fwd_only = "--fwd" in sys.argv
    opacity = "trained" if "--trained" in sys.argv else "init"
    if "--nocull" in sys.argv:
        from ibgs_amd import rasterizer
        rasterizer.TILE_CULL = False
    c = syn.CONFIGS[cfg]
    inp = syn.make_scene(c["P"], c["W"], c["H"], sh_degree=c["sh_degree"], seed=c["seed"], opacity=opacity)
    dev = "cuda"
    st = hipref.settings_from(inp, dev)
    lv = hipref.leaf_inputs(inp, dev, requires_grad=not fwd_only)
    from ibgs_amd.rasterizer import GaussianRasterizer
    rast = GaussianRasterizer(st)
    target = torch.rand(3, c["H"], c["W"], device=dev)
    for it in range(iters + 2):
        if it == 2:
            torch.cuda.synchronize(); t0 = time.time()
        for v in lv.values():
            if v is not None and v.grad is not None:
                v.grad = None
        outs = rast(means3D=lv["means3D"], means2D=lv["means2D"], means2D_abs=lv["means2D_abs"], opacities=lv["opacities"],
                    shs=lv["shs"], scales=lv["scales"], rotations=lv["rotations"])
        if not fwd_only:
            loss = (outs[0] - target).abs().mean()
            loss.backward()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / iters
    R = outs[0].grad_fn.num_rendered if not fwd_only else -1
    print("cfg %s  %s  %.3f ms/iter  R=%s" % (cfg, "fwd" if fwd_only else "fwd+bwd", dt * 1e3, R))


if __name__ == "__main__":
    main()
