"""Rectangle statistics of a bench workload (CPU oracle, test infrastructure): how many Gaussians take the binning's large-rectangle path
(more than 8 tiles wide or high), how many waves of 64 consecutive depth ranks hold at least one of them, cells reached per Gaussian.
usage: python tools/rect_stats.py [init|trained] [config]"""
import sys
import numpy as np
import oracle
from ibgs_amd import synthetic as syn

opacity = sys.argv[1] if len(sys.argv) > 1 else "init"
cfg = syn.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "C3"]
inp = syn.make_scene(cfg["P"], cfg["W"], cfg["H"], sh_degree=3, seed=1, opacity=opacity)
ref = oracle.forward(inp, cull=True)
r = ref["rect4"].astype(np.int64)
print("rect4 columns sample", r[ref["tiles_touched"] > 0][:3])
alive = ref["tiles_touched"] > 0
x0, y0, x1, y1 = (r[:, k] for k in range(4))
if (x1 >= x0).all() and not (r[:, 2] >= r[:, 0]).all():
    pass
w, h = (x1 - x0)[alive], (y1 - y0)[alive]
big = (w > 8) | (h > 8)
huge = (w * h) > 256
print("opacity", opacity, "P", cfg["P"], "alive with tiles", int(alive.sum()), "R", int(ref["num_rendered"]))
print("large path (w > 8 or h > 8): %d = %.2f%%; unmasked (area > 256 tiles): %d" % (big.sum(), 100.0 * big.mean(), huge.sum()))
order = np.argsort(ref["depths"][alive], kind="stable")
bs = big[order]
n64 = len(bs) // 64
print("waves of 64 ranks with >= 1 large: %.1f%%, mean large per wave %.2f, max %d" % (100.0 * bs[:n64 * 64].reshape(n64, 64).any(1).mean(), bs[:n64 * 64].reshape(n64, 64).sum(1).mean(), bs[:n64 * 64].reshape(n64, 64).sum(1).max()))
cx = (x1[alive] - 1) // 8 - x0[alive] // 8 + 1
cy = (y1[alive] - 1) // 8 - y0[alive] // 8 + 1
cells = cx * cy
print("cells in the rectangle per Gaussian: mean %.2f; large ones: mean %.1f max %d; w,h percentiles (50,90,99,99.9):" % (cells.mean(), cells[big].mean() if big.any() else 0, cells.max()),
      np.percentile(w, [50, 90, 99, 99.9]), np.percentile(h, [50, 90, 99, 99.9]))
