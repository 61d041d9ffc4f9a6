"""Where does a fused-glue fuzz case part from the float64 build?  (tools/fuzz_fused.py SEED, case INDEX)
python tools/diag_fused_case.py 9105 1
Prints, per parameter gradient, how much of |HIP - f64|^2 the worst Gaussians carry, and the pixels at which the forward outputs (median depth, warped image,
valid-source count) of the HIP path and of the fp32 oracle differ from the float64 build's by more than rounding -- decisions on rounded floats that fell the other way."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from tests import fuzz_cases as fc
from tests.metrics import rel_l2
from tests.test_gpu_fused_planes import _oracle_chain, _run, _scene

seed, index = int(sys.argv[1]), int(sys.argv[2])
c = fc.fused_case(seed, index)
print("case", c)
dev, g, cams, scene, pipe, args, bg = _scene(P=c["P"], W=c["W"], H=c["H"], seed=c["seed"])
planes = {} if os.environ.get("DIAG_SAME_PLANES", "1") == "1" else None          # 1 (default): the oracle is evaluated at the plane map the kernels built (see _oracle_chain)
o_fus, g_fus = _run(True, c["learnt"], g, dev, cams, scene, pipe, args, bg, planes_out=planes)
ref32, g32 = _oracle_chain(c["learnt"], g, dev, cams, scene, bg, planes=planes)
with oracle.variant("f64"):
    ref64, g64 = _oracle_chain(c["learnt"], g, dev, cams, scene, bg, planes=planes)
if planes is not None:
    from ibgs_amd import renderer as _r, simple_scene as _ss
    print("plane map of the kernels vs the torch glue: see tests/test_gpu_glue_golden.py; rows with tiles: %d of %d" % (planes["have"].sum(), planes["have"].size))
H, W = c["H"], c["W"]
names = ["_xyz", "_rotation", "_scaling", "_opacity"]
for n in names:
    dh = (g_fus[n] - g64[n]).reshape(c["P"], -1); do = (g32[n] - g64[n]).reshape(c["P"], -1)
    eh, eo = (dh ** 2).sum(1), (do ** 2).sum(1)
    top = np.argsort(-eh)[:5]
    print("%-10s HIP %.2e oracle32 %.2e | worst Gaussians (HIP): %s | (oracle32): %s" % (n, rel_l2(g_fus[n], g64[n]), rel_l2(g32[n], g64[n]),
          ", ".join("%d: %.0f%%" % (i, 100 * eh[i] / eh.sum()) for i in top), ", ".join("%d: %.0f%%" % (i, 100 * eo[i] / eo.sum()) for i in np.argsort(-eo)[:5])))
hip = {"median_depth": o_fus["median_intersected_depth"].cpu().numpy().reshape(-1), "warped_image": o_fus["warped_image"].cpu().numpy().reshape(15, -1),
       "color": o_fus["render"].cpu().numpy().reshape(3, -1), "normal_map": o_fus["rendered_normal"].cpu().numpy().reshape(3, -1)}
for k in hip:
    a, b32, b64 = hip[k], np.asarray(ref32[k]).reshape(hip[k].shape), np.asarray(ref64[k]).reshape(hip[k].shape)
    dh = np.abs(a - b64).reshape(-1, H * W).max(0); do = np.abs(b32 - b64).reshape(-1, H * W).max(0)
    thr = 1e-3 * max(1e-6, np.abs(b64).max())
    ph, po = np.flatnonzero(dh > thr), np.flatnonzero(do > thr)
    print("%-14s pixels off by > %.1e: HIP %d %s | oracle32 %d %s" % (k, thr, ph.size, [(int(p % W), int(p // W), float("%.3g" % dh[p])) for p in ph[:6]], po.size, [(int(p % W), int(p // W), float("%.3g" % do[p])) for p in po[:6]]))
# which Gaussians sit on those pixels?  (the oracle's lists)
if "point_list" in ref64:
    pass
