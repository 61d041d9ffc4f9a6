"""How much would a 32 x 16 super-tile backward share?  (docs/EXPERIMENTS.md section 7, round 3: the measured basis of the bound on VERDICT item 3.)
Runs the C3 colour backward in deterministic mode, whose slab holds one row per (list entry, wave of its tile) -- rows the backward never
wrote stay zero -- and counts, among the PROCESSED (Gaussian, tile) entries, those whose Gaussian is also processed in the horizontal
partner tile (tiles 2k, 2k + 1 of a row) and in the vertical one."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibgs_amd import rasterizer, synthetic as syn
from tests import hipref

for opacity in ("init", "trained"):
    c = syn.CONFIGS["C3"]
    inp = syn.make_scene(c["P"], c["W"], c["H"], sh_degree=3, seed=c["seed"], opacity=opacity)
    rasterizer.DETERMINISTIC = True; rasterizer.KEEP_DET_SCRATCH = True
    outs, lv, _ = hipref.run_forward(inp)
    ist = hipref.internal_state(outs, inp)
    g = torch.randn(3, c["H"], c["W"], device="cuda")
    (outs["color"] * g).sum().backward()
    torch.cuda.synchronize()
    R = ist["R"]
    det = rasterizer._CModule.last_det
    slab = det[: R * 64].view(torch.float32).view(R, 16)          # one wave per tile at 1080p: row = list position
    processed = (slab != 0).any(dim=1)
    ranges = torch.as_tensor(ist["ranges"].astype(np.int64), device="cuda")
    gx = (c["W"] + 15) // 16
    counts = ranges[:, 1] - ranges[:, 0]
    tile_of = torch.repeat_interleave(torch.arange(ranges.shape[0], device="cuda"), counts)
    ids = torch.as_tensor(ist["point_list"].astype(np.int64), device="cuda")
    t, gid = tile_of[processed], ids[processed]
    ntiles = ranges.shape[0]
    key = gid * ntiles + t
    keys_sorted = torch.sort(key).values

    def has(k):
        pos = torch.searchsorted(keys_sorted, k).clamp(max=keys_sorted.numel() - 1)
        return keys_sorted[pos] == k
    tx, ty = t % gx, t // gx
    partner_h = torch.where(tx % 2 == 0, t + 1, t - 1)
    ok_h = (torch.where(tx % 2 == 0, tx + 1, tx - 1) < gx)
    shared_h = has(gid * ntiles + partner_h) & ok_h
    partner_v = torch.where(ty % 2 == 0, t + gx, t - gx)
    ok_v = partner_v < ntiles
    shared_v = has(gid * ntiles + partner_v.clamp(max=ntiles - 1)) & ok_v
    n = int(processed.sum())
    print("C3 opacity=%s: R %d list entries, %d processed by the backward (%.1f %%); of those %.1f %% have their Gaussian processed in the horizontal partner tile, %.1f %% in the vertical one"
          % (opacity, R, n, 100.0 * n / R, 100.0 * float(shared_h.float().mean()), 100.0 * float(shared_v.float().mean())))
    print("   => reduce + atomic launches with 32 x 16 super-tiles: %.0f %% of today's (perfect sharing would be 50 %%)" % (100.0 * (1.0 - 0.5 * float(shared_h.float().mean()))))
    rasterizer.DETERMINISTIC = False; rasterizer.KEEP_DET_SCRATCH = False
    del outs, lv, det, slab
    torch.cuda.empty_cache()
