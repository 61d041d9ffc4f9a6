"""Stage times of the blend kernels for the three wave-shape choices (tile, quadrant, the library's own = hybrid below 4 096 tiles).
usage: python tools/hybrid_stages.py W H [opacity [cluster [geo]]]   (IBGS_HYBRID_THETA in the environment changes the hybrid's threshold)"""
import gc, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
W, H = int(sys.argv[1]), int(sys.argv[2])
opacity = sys.argv[3] if len(sys.argv) > 3 else "init"
cluster = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
geo = len(sys.argv) > 5 and sys.argv[5] == "geo"          # render_geo passes: tile = half-tile waves; the library picks by the frame size alone
sys.argv = sys.argv[:1]
import bench
from ibgs_amd import _lib, rasterizer, synthetic as syn
_lib.load()
dev = torch.device("cuda", 0)
for shape in (("tile", "quadrant", None) if "IBGS_HYBRID_THETA" not in os.environ else (None,)):
    rasterizer.WAVE_SHAPE = shape
    syn.CONFIGS["_sweep"] = dict(P=1_000_000, W=W, H=H, sh_degree=3, seed=3)
    wl = bench.Workload("_sweep", 0, dev, opacity, geo, False, 1234, cluster=cluster, anisotropy="plane" if geo else None)
    for _ in range(6):
        wl.local_step()
    gc.collect(); gc.disable()          # (a generation-2 collection inside the timed steps costs 40-55 ms in a process that holds torch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        wl.local_step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    _lib.timing_enable(_lib.STAGES)
    for _ in range(5):
        wl.local_step()
    torch.cuda.synchronize()
    st = {k: v[0] / 5.0 for k, v in _lib.timing_collect().items()}
    _lib.timing_enable([])
    print("%dx%d %-7s cluster %.1f theta %s %s %-8s step %.3f  fwd %.3f  bwd %.3f  window %.3f" % (W, H, opacity, cluster, os.environ.get("IBGS_HYBRID_THETA", "-"), "geo" if geo else "   ", shape or "library", ms, st.get("render_fwd", 0), st.get("render_bwd", 0), st.get("geo_window", 0)), flush=True)
    del wl
    torch.cuda.empty_cache()
