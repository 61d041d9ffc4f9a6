"""Where do the blend kernels' two work decompositions cross over?  One wave per 16 x 16 tile against one wave per 8 x 8 quadrant, C3's Gaussians
on frames of different sizes (the library switches on the number of tiles).  usage: python tools/sweep_wave_shape.py [P]"""
import gc, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0]] + sys.argv[1:]
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
import bench
from ibgs_amd import _lib, rasterizer, synthetic as syn
_lib.load()
dev = torch.device("cuda", 0)
SIZES = {"coarse": ((1280, 720), (1120, 630), (960, 540), (800, 800), (800, 450), (640, 360), (400, 400)),
         "fine": ((1280, 720), (1248, 704), (1280, 672), (1152, 720), (1216, 672), (1200, 640), (1120, 630))}
for W, H in SIZES[sys.argv[2] if len(sys.argv) > 2 else "coarse"]:
    nt = ((W + 15) // 16) * ((H + 15) // 16)
    for opacity in ("init", "trained"):
        row = []
        for shape in ("tile", "quadrant", None):          # None: the library's own choice (below 4 096 tiles: per tile, the hybrid kernels)
            rasterizer.WAVE_SHAPE = shape
            syn.CONFIGS["_sweep"] = dict(P=P, W=W, H=H, sh_degree=3, seed=3)
            wl = bench.Workload("_sweep", 0, dev, opacity, False, False, 1234)
            for _ in range(6):
                wl.local_step()
            gc.collect(); gc.disable()          # (a generation-2 collection inside the timed steps costs 40-55 ms in a process that holds torch)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 40
            for _ in range(n):
                wl.local_step()
            torch.cuda.synchronize()
            row.append((time.perf_counter() - t0) / n * 1e3)
            del wl
            torch.cuda.empty_cache()
        print("%4dx%-4d %5d tiles  P %d  %-7s  tile %.3f ms   quadrant %.3f ms   library %.3f ms   library/best %.2f" % (W, H, nt, P, opacity, row[0], row[1], row[2], row[2] / min(row[0], row[1])), flush=True)
