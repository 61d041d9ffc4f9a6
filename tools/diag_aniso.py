"""Per-gradient parity breakdown on anisotropic scenes (diagnostic; prints rel L2 per output, split by conic conditioning)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from tests import hipref
from tests.metrics import rel_l2
from tests.scenes import scene, giant_needles
from tests.test_gpu_parity import GRAD_PAIRS, rnd

def diag(tag, inp):
    H, W = inp["H"], inp["W"]
    g = rnd((3, H, W), 1)
    ref = oracle.forward(inp, cull=True)
    gb = oracle.backward(inp, ref, g)
    outs, lv, _ = hipref.run_forward(inp)
    (outs["color"] * torch.as_tensor(g, device="cuda")).sum().backward()
    co = ref["conic_opacity"]; vis = ref["radii"] > 0
    cond = np.full(co.shape[0], 1.0)
    cond[vis] = 1.0 - co[vis, 1] ** 2 / (co[vis, 0] * co[vis, 2])          # 1 - rho^2
    print("==", tag, "R", ref["num_rendered"], "skips", oracle.power_skips())
    for lk, rk in GRAD_PAIRS:
        if lv.get(lk) is None or lv[lk].grad is None: continue
        a = lv[lk].grad.cpu().numpy().reshape(co.shape[0], -1); b = gb[rk].reshape(co.shape[0], -1)
        if np.abs(b).max() == 0: continue
        line = "  %-12s all %.2e" % (lk, rel_l2(a, b))
        for lo, hi in ((1e-1, 2), (1e-2, 1e-1), (1e-3, 1e-2), (1e-4, 1e-3), (1e-5, 1e-4), (-1, 1e-5)):
            m = vis & (cond > lo) & (cond <= hi)
            if m.sum(): line += " | (%g,%g] n=%d %.1e" % (lo, hi, m.sum(), rel_l2(a[m], b[m]))
        print(line)

diag("needle", scene(P=4000, W=208, H=144, deg=3, seed=31, opacity="trained", anisotropy="needle"))
diag("plane", scene(P=4000, W=208, H=144, deg=3, seed=31, opacity="trained", anisotropy="plane"))
diag("giant 4/0.05", giant_needles(stretch=4.0, thin=0.05))
diag("giant 2/0.05", giant_needles(stretch=2.0, thin=0.05))
