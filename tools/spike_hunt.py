"""Where do the rare 50 ms steps of the trained_geo workload come from?  Runs the workload for N steps, GPU drained after every step, and prints the
steps that took more than 5 ms with what changed around them (allocator statistics, garbage collector counters, hint misses)."""
import gc, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from ibgs_amd import rasterizer as rz
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
dev = torch.device("cuda", 0)
if mode == "after_geo":          # as the default bench line: another geo workload lives and dies first
    w0 = bench.Workload("C3", 0, dev, "init", True, False, 1234)
    for _ in range(30):
        w0.local_step()
    del w0
    torch.cuda.empty_cache()
wl = bench.Workload("C3", 0, dev, "trained", True, False, 1234, cluster=0.3, anisotropy="plane", scale_sigma=1.0)
if mode in ("nogc", "long", "nosync"):
    gc.disable()
gc.collect(); gc.freeze()
prev = torch.cuda.memory_stats()
T0 = time.perf_counter()
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 80):
    t0 = time.perf_counter()
    wl.local_step()
    if mode != "nosync":
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    st = torch.cuda.memory_stats() if mode not in ("long", "nosync") else prev
    if dt > (20.0 if mode in ("long", "nosync") else 5.0):
        print("t = %.2f s, step %d: %.1f ms | hipMalloc calls +%d, frees +%d, alloc retries +%d | gc counts %s | hint misses %d | reserved %.2f GB"
              % (time.perf_counter() - T0, i, dt, st["num_device_alloc"] - prev["num_device_alloc"], st["num_device_free"] - prev["num_device_free"], st["num_alloc_retries"] - prev["num_alloc_retries"],
                 gc.get_count(), rz.HINT_MISSES, st["reserved_bytes.all.current"] / 2 ** 30))
    prev = st
print("done", mode)
