"""How evenly does the colour backward's work fall on the 1 024 SIMDs?  (docs/EXPERIMENTS.md section 7, round 3.)
One wave per tile, every tile resident at once (8 160 waves on 8 192 slots at 1080p): the kernel ends when the most loaded SIMD is done.
Per-tile work from the deterministic backward's slab (which list entries were PROCESSED, = reduced and added) and from the forward's
n_contrib (how far each tile's list is WALKED): cost model 230 instructions per processed entry, 60 per walked-only entry (ISA counts).
Placement models: workgroup i -> XCD i % 8, then round-robin over that XCD's 128 SIMDs; and a random placement.  Prints max / mean load."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibgs_amd import rasterizer, synthetic as syn
from tests import hipref

cluster = float(sys.argv[sys.argv.index("--cluster") + 1]) if "--cluster" in sys.argv else 0.0
args = [a for a in sys.argv[1:] if a in ("init", "trained")] or ["init", "trained"]
for opacity in args:
    c = syn.CONFIGS["C3"]
    inp = syn.make_scene(c["P"], c["W"], c["H"], sh_degree=3, seed=c["seed"], opacity=opacity)
    if cluster > 0:          # as bench.py --cluster
        k = int(cluster * c["P"])
        inp["means3D"] = inp["means3D"].copy(); inp["means3D"][:k] = inp["means3D"][:k] * 0.3 + np.array([0.5, 0.25, 0.0], np.float32)
    rasterizer.DETERMINISTIC = True; rasterizer.KEEP_DET_SCRATCH = True
    outs, lv, _ = hipref.run_forward(inp)
    ist = hipref.internal_state(outs, inp)
    g = torch.randn(3, c["H"], c["W"], device="cuda")
    (outs["color"] * g).sum().backward()
    torch.cuda.synchronize()
    R = ist["R"]
    det = rasterizer._CModule.last_det
    slab = det[: R * 64].view(torch.float32).view(R, 16)
    processed = (slab != 0).any(dim=1)
    ranges = ist["ranges"].astype(np.int64)
    W, H = c["W"], c["H"]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    counts = ranges[:, 1] - ranges[:, 0]
    tile_of = np.repeat(np.arange(ranges.shape[0]), counts)
    proc = np.bincount(tile_of[processed.cpu().numpy()], minlength=gx * gy).astype(np.float64)
    nc = np.zeros((gy * 16, gx * 16), np.int64); nc[:H, :W] = ist["n_contrib"].reshape(H, W)
    top = np.minimum(nc.reshape(gy, 16, gx, 16).max(axis=(1, 3)).reshape(-1), counts).astype(np.float64)
    work = 230.0 * proc + 60.0 * (top - proc)
    nt = gx * gy
    # launch order of the tiles: blocks of 8 x 8 tiles, row-major inside a block, blocks row-major (TMAP_BLOCK; the XCD round-robin goes over consecutive workgroups)
    order = []
    for by in range(0, gy, 8):
        for bx in range(0, gx, 8):
            for ty in range(by, min(by + 8, gy)):
                for tx in range(bx, min(bx + 8, gx)):
                    order.append(ty * gx + tx)
    order = np.array(order)
    w_launch = work[order]
    nsimd = 1024
    mean = work.sum() / nsimd
    load_rr = np.zeros(nsimd)
    for i, w in enumerate(w_launch):
        xcd = i % 8; k = i // 8
        load_rr[xcd * 128 + k % 128] += w
    rng = np.random.default_rng(0)
    load_rand = np.zeros(nsimd); np.add.at(load_rand, rng.integers(0, nsimd, nt), w_launch)
    # greedy, longest first (what a work queue sorted by list length approaches)
    import heapq
    h = [0.0] * nsimd; heapq.heapify(h)
    for w in np.sort(work)[::-1]:
        heapq.heappush(h, heapq.heappop(h) + w)
    lpt = max(h)
    # greedy in launch order (a work queue without sorting)
    h = [0.0] * nsimd; heapq.heapify(h)
    for w in w_launch:
        heapq.heappush(h, heapq.heappop(h) + w)
    fifo = max(h)
    # the placement as measured (tools/wave_trace.py --placement): workgroups i and i + 1024 share a SIMD when every workgroup is one wave and all are
    # resident.  Today: launch order = tile order (rr map).  Snake: tiles sorted by a key, rank r -> round r // 1024, class r % 1024 (reversed in odd rounds)
    def classes(perm):          # perm[i] = tile launched as workgroup i
        load = np.zeros(1024); np.add.at(load, np.arange(len(perm)) % 1024, work[perm]); return load.max() / (work.sum() / 1024)
    def snake(key):
        r = np.argsort(-key, kind="stable"); perm = np.empty(len(r), np.int64)
        for k in range((len(r) + 1023) // 1024):
            seg = r[k * 1024:(k + 1) * 1024]
            perm[k * 1024:k * 1024 + len(seg)] = seg if k % 2 == 0 else np.concatenate([seg[::-1], []]).astype(np.int64) if len(seg) == 1024 else seg[::-1]
        return perm
    def snake_classes(key):          # odd rounds run the classes backwards (a short last round still starts at the far end)
        r = np.argsort(-key, kind="stable"); load = np.zeros(1024)
        for i, t in enumerate(r):
            k, c = divmod(i, 1024)
            load[c if k % 2 == 0 else 1023 - c] += work[t]
        return load.max() / (work.sum() / 1024)
    print("   measured placement model (workgroup %% 1024): tile order %.3f (mean / max = %.3f); snake by true work %.3f; snake by list length %.3f; snake by walked length %.3f"
          % (classes(np.arange(nt)), 1.0 / classes(np.arange(nt)), snake_classes(work), snake_classes(counts.astype(np.float64)), snake_classes(top)))
    # ... and with the ISSUE RATE in the model: the waves of a SIMD share its VALU (processor sharing); two or more waves together issue an instruction
    # every 2.65 cycles, a single wave only every 5.1 (profiles/r02_probe_xlane.txt) -- so the stretch the heaviest tile of a SIMD runs ALONE costs double
    def finish(weights):          # weights of the tiles of one SIMD -> time (in units of work at the shared rate)
        w = np.sort(np.asarray(weights, np.float64))
        if len(w) == 0: return 0.0
        if len(w) == 1: return w[0] * 5.1 / 2.65
        return w[:-1].sum() + w[-2] + (w[-1] - w[-2]) * 5.1 / 2.65          # everything is shared until the second heaviest is done, the rest runs alone
    def span(assign):          # assign[c] = list of tiles of class c
        return max(finish(work[a]) for a in assign)
    r = np.argsort(-top, kind="stable")
    def arrange(dirs, pairs=False):
        cls = [[] for _ in range(1024)]
        if pairs:          # the two heaviest of a class adjacent in rank: ranks 2k, 2k + 1 -> class k; the remaining strata by `dirs`
            for k in range(1024):
                cls[k] += [r[2 * k], r[2 * k + 1]] if 2 * k + 1 < len(r) else []
            rest = r[2048:]; base = 0
        else:
            rest = r; base = 0
        for i, t in enumerate(rest):
            k, c = divmod(i, 1024)
            cls[c if dirs[k % len(dirs)] == "F" else 1023 - c].append(t)
        return cls
    ideal = work.sum() / 1024
    tile_order_cls = [list(range(c, nt, 1024)) for c in range(1024)]
    print("   with the single-wave issue rate in the model, span / (total work / 1024): tile order %.3f; snake %.3f; adjacent pairs + snake of the rest %.3f; pairs + rest 'RFRFRF' %.3f; pairs + 'RRFFRF' %.3f"
          % (span(tile_order_cls) / ideal, span(arrange("FR")) / ideal, span(arrange("FR", True)) / ideal, span(arrange("RF", True)) / ideal, span(arrange("RRFFRF", True)) / ideal))
    print("   walked length per tile: mean %.1f, cv %.3f, max / mean %.2f%s" % (top.mean(), top.std() / top.mean(), top.max() / top.mean(), ("  (cluster %g)" % cluster) if cluster else ""))
    # hybrid decomposition (not built): tiles heavier than thr x the mean as four quadrant waves of `f` x the tile wave's cost each, everything dealt out in snake order
    def hybrid(thr, f=0.36):
        wts = []
        for t in range(nt):
            if work[t] > thr * work.mean(): wts += [work[t] * f] * 4
            else: wts.append(work[t])
        wts = np.sort(np.asarray(wts))[::-1]
        cls = [[] for _ in range(1024)]
        for i, wgt in enumerate(wts):
            k, c = divmod(i, 1024)
            cls[c if k % 2 == 0 else 1023 - c].append(wgt)
        return max(finish(a) for a in cls) / (work.sum() / 1024), len(wts) - nt
    print("   hybrid (model): " + "; ".join("tiles > %.2f x mean split: span %.3f (+%d waves)" % ((thr,) + hybrid(thr)) for thr in (1.05, 1.15, 1.3, 1.6)))
    print("C3 opacity=%s: tiles %d, walked entries %.2f M, processed %.2f M; work per tile: mean %.0f, cv %.2f, max %.0f (%.1f x mean)"
          % (opacity, nt, top.sum() / 1e6, proc.sum() / 1e6, work.mean(), work.std() / work.mean(), work.max(), work.max() / work.mean()))
    print("   per-SIMD load, max / mean: round-robin placement %.2f, random placement %.2f, work queue in launch order %.2f, work queue longest first %.2f"
          % (load_rr.max() / mean, load_rand.max() / mean, fifo / mean, lpt / mean))
    rasterizer.DETERMINISTIC = False; rasterizer.KEEP_DET_SCRATCH = False
    del outs, lv, det, slab
    torch.cuda.empty_cache()
