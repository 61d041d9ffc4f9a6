"""Per-tile work statistics of the C3 scene (diagnostic)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibgs_amd import synthetic as syn
from tests import hipref
c = syn.CONFIGS["C3"]
inp = syn.make_scene(c["P"], c["W"], c["H"], sh_degree=3, seed=3, opacity="trained" if "--trained" in sys.argv else "init")
outs, lv, st = hipref.run_forward(inp)
ist = hipref.internal_state(outs, inp)
W, H = c["W"], c["H"]; gx, gy = (W + 15) // 16, (H + 15) // 16
nc = ist["n_contrib"].reshape(H, W)
pad = np.zeros((gy * 16, gx * 16), np.uint32); pad[:H, :W] = nc
tmax = pad.reshape(gy, 16, gx, 16).max(axis=(1, 3))
tlen = (ist["ranges"][:, 1] - ist["ranges"][:, 0]).reshape(gy, gx)
print("R", ist["R"], "list len mean/max", tlen.mean(), tlen.max())
print("processed (max n_contrib per tile) mean %.1f p50 %.1f p90 %.1f p99 %.1f max %d" % (tmax.mean(), np.percentile(tmax, 50), np.percentile(tmax, 90), np.percentile(tmax, 99), tmax.max()))
print("mean n_contrib per pixel", nc.mean(), "fraction processed/list", tmax.sum() / tlen.sum())
rows = tmax.mean(axis=1)
print("per tile-row mean work:", np.round(rows[::4]).astype(int))
# band mapping: 8 XCD bands of contiguous tiles
flat = tmax.reshape(-1); per = (flat.size + 7) // 8
print("work per XCD band:", [int(flat[k * per:(k + 1) * per].sum()) for k in range(8)])
print("work per XCD row-interleave:", [int(tmax[k::8].sum()) for k in range(8)])
print("work per XCD round-robin:", [int(flat[k::8].sum()) for k in range(8)])
