"""Diagnostic: HIP colour backward vs oracle on a small scene, per gradient and as ratio statistics."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from ibgs_amd import rasterizer
from tests import hipref
from tests.scenes import scene
from tests.metrics import rel_l2

for shape in ("tile",):
    rasterizer.WAVE_SHAPE = shape
    inp = scene(P=4000, deg=0, seed=10, opacity=sys.argv[1] if len(sys.argv) > 1 else "init")
    alt = {k: v for k, v in inp.items() if k not in ("shs",)}
    alt["colors_precomp"] = np.random.default_rng(1).uniform(0, 1, (4000, 3)).astype(np.float32)
    ref = oracle.forward(alt, cull=True)
    outs, lv, _ = hipref.run_forward(alt)
    g = np.random.default_rng(2).normal(size=(3, inp["H"], inp["W"])).astype(np.float32)
    (outs["color"] * torch.as_tensor(g, device="cuda")).sum().backward()
    rb = oracle.backward(alt, ref, g)
    for lk, rk in (("colors_precomp", "dL_dcolors"), ("opacities", "dL_dopacity"), ("means2D", "dL_dmeans2D"), ("means2D_abs", "dL_dmeans2D_abs"), ("means3D", "dL_dmeans3D"), ("scales", "dL_dscales")):
        a = lv[lk].grad.cpu().numpy(); b = rb[rk].reshape(a.shape)
        m = np.abs(b) > 1e-6 * np.abs(b).max()
        ratio = a[m] / b[m]
        print(shape, lk, "relL2 %.3e" % rel_l2(a, b), "ratio median %.4f p10 %.4f p90 %.4f" % (np.median(ratio), np.percentile(ratio, 10), np.percentile(ratio, 90)), "nonzero hip %d oracle %d" % ((a != 0).sum(), (b != 0).sum()))
