"""Randomised parity sweep on the MI355X (not part of the test suite): random sizes / degrees / modes against the oracle.
python tools/fuzz_parity.py [n_cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from ibgs_amd import rasterizer, synthetic as syn
from tests import hipref
from tests.metrics import l1, rel_l2
from tests.test_gpu_parity import add_sources, scene

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {"color": 0.0, "grad": 0.0, "ncontrib": 1.0}
bad = 0
for case in range(n_cases):
    P = int(rng.choice([1, 2, 7, 63, 64, 65, 300, 1500, 4000, 9000]))
    W, H = int(rng.integers(8, 320)), int(rng.integers(8, 240))
    deg = int(rng.integers(0, 4)); geo = bool(rng.integers(0, 3) == 0)
    opacity = str(rng.choice(["init", "trained"])); smul = float(rng.choice([0.5, 1.0, 2.5]))
    rasterizer.WAVE_SHAPE = [None, "tile", "quadrant"][int(rng.integers(0, 3))]
    inp = scene(P=P, W=W, H=H, deg=deg, seed=int(rng.integers(0, 10**6)), opacity=opacity, planes=geo, scale_mul=smul)
    if geo:
        inp = add_sources(inp, n_src=int(rng.integers(1, 6)), L=int(rng.integers(1, 9)))
    ref = oracle.forward(inp, cull=True)
    outs, lv, _ = hipref.run_forward(inp)
    ist = hipref.internal_state(outs, inp)
    o = hipref.to_np(outs)
    ok = ist["R"] == ref["num_rendered"] and np.array_equal(ist["point_list"], ref["point_list"]) and np.array_equal(o["radii"], ref["radii"])
    dc = l1(o["color"], ref["color"])
    nc = float((ist["n_contrib"] == ref["n_contrib"]).mean())
    g = rng.standard_normal((3, H, W)).astype(np.float32)
    if geo:      # every differentiable geo output takes part
        gn = rng.standard_normal((3, H, W)).astype(np.float32); gdp = rng.standard_normal((1, H, W)).astype(np.float32)
        gw = rng.standard_normal((15, H, W)).astype(np.float32)
        ((outs["color"] * torch.as_tensor(g, device="cuda")).sum() + (outs["normal_map"] * torch.as_tensor(gn, device="cuda")).sum()
         + (outs["median_depth"] * torch.as_tensor(gdp, device="cuda")).sum() + (outs["warped_image"] * torch.as_tensor(gw, device="cuda")).sum()).backward()
        rb = oracle.backward(inp, ref, g, gn, gdp, gw)
    else:
        (outs["color"] * torch.as_tensor(g, device="cuda")).sum().backward()
        rb = oracle.backward(inp, ref, g)
    gr = 0.0
    names = {"dL_dmeans3D": "means3D", "dL_dopacity": "opacities", "dL_dscales": "scales"}
    if geo:
        names["dL_dall_map"] = "all_map"
    for k, v in names.items():
        a = lv[v].grad.cpu().numpy().reshape(np.asarray(rb[k]).shape)
        if np.abs(rb[k]).sum() > 0:
            gr = max(gr, float(rel_l2(a, rb[k])))
    worst["color"] = max(worst["color"], dc); worst["grad"] = max(worst["grad"], gr); worst["ncontrib"] = min(worst["ncontrib"], nc)
    flag = ok and dc < 1e-5 and gr < (2e-2 if geo else 5e-3) and nc > 0.995
    bad += not flag
    print("%s case %2d: P=%5d %3dx%3d deg=%d geo=%d shape=%-8s R=%7d | lists %s colour L1 %.1e n_contrib eq %.4f grad relL2 %.1e"
          % ("ok  " if flag else "FAIL", case, P, W, H, deg, geo, rasterizer.WAVE_SHAPE, ist["R"], ok, dc, nc, gr), flush=True)
print("worst:", worst, "failures:", bad)
sys.exit(1 if bad else 0)
