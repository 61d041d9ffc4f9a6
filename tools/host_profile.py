"""Host time of one rasterizer step: a C1-sized workload (the GPU needs ~0.35 ms for it, so the step is host-bound),
N steps forward + L1 loss + backward: wall time per step with the GPU idle-waiting, and cProfile's top entries.
usage: python tools/host_profile.py [steps] [--geo]"""
import cProfile, io, os, pstats, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 300
geo = "--geo" in sys.argv
dev = torch.device("cuda", 0)
wl = bench.Workload("C1", 0, dev, "trained", geo, False, 7)
for _ in range(20):
    wl.local_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    wl.local_step()
t_host = (time.perf_counter() - t0) / n * 1e3          # the host's own time per step (it never waits for the GPU here except inside the forward's R read-back)
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / n * 1e3
print("C1%s, %d steps: %.3f ms per step until the host is done queueing, %.3f ms until the GPU is done" % (" geo" if geo else "", n, t_host, t_all))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    wl.local_step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print("\n".join(l for l in s.getvalue().split("\n") if l.strip())[:6000])
