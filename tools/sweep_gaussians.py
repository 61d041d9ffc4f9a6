"""BASELINE metric as a curve: train-step ms (rasterizer fwd + L1 loss + bwd) at 1920x1080, SH 3, against the number of Gaussians."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibgs_amd import rasterizer, synthetic as syn
from tests import hipref

W, H = 1920, 1080
print("| Gaussians | R (tile entries) | forward ms | fwd+bwd ms | fps |\n|---|---|---|---|---|")
for P in (100_000, 250_000, 500_000, 1_000_000, 2_000_000, 5_000_000):
    inp = syn.make_scene(P, W, H, sh_degree=3, seed=3)
    lv = hipref.leaf_inputs(inp, "cuda"); st = hipref.settings_from(inp, "cuda")
    rast = rasterizer.GaussianRasterizer(st)
    tgt = torch.rand(3, H, W, device="cuda")
    call = lambda: rast(means3D=lv["means3D"], means2D=lv["means2D"], means2D_abs=lv["means2D_abs"], opacities=lv["opacities"], shs=lv["shs"],
                        scales=lv["scales"], rotations=lv["rotations"])
    def step():
        for v in lv.values():
            if v is not None: v.grad = None
        torch.nn.functional.l1_loss(call()[0], tgt).backward()
    def fwd():
        with torch.no_grad(): call()
    res = []
    for fn in (fwd, step):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30): fn()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 30 * 1e3)
    print("| %d | %d | %.2f | %.2f | %.0f |" % (P, rasterizer.LAST_NUM_RENDERED, res[0], res[1], 1000.0 / res[1]), flush=True)
    del lv, rast, tgt; torch.cuda.empty_cache()
