import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tests.test_gpu_fused_planes import _scene, _run
from tests.metrics import rel_l2
dev, g, cams, scene, pipe, args, bg = _scene()
for learnt in (True, False):
    o_ref, g_ref = _run(False, learnt, g, dev, cams, scene, pipe, args, bg)
    o_fus, g_fus = _run(True, learnt, g, dev, cams, scene, pipe, args, bg)
    o_ref2, g_ref2 = _run(False, learnt, g, dev, cams, scene, pipe, args, bg)
    print(learnt, {n: (float(rel_l2(g_fus[n], g_ref[n])), float(rel_l2(g_ref2[n], g_ref[n]))) for n in g_ref if g_ref[n] is not None and g_fus[n] is not None})
