"""Which torch operators (not the library's launches) a trainer-shaped iteration calls, and from where: aten ops with their Python call sites, sorted by the device
time of the kernels they launch.  usage: python tools/train_iter_ops.py [full] [sh_factored]   (MI355X)"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
full = "full" in sys.argv; shf = "sh_factored" in sys.argv
sys.argv = sys.argv[:1]
import bench
from ibgs_amd import synthetic as syn
from ibgs_amd.optim import FusedAdam
from torch.profiler import ProfilerActivity, profile
dev = torch.device("cuda", 0)
ti = bench.TrainIteration(dev, syn.CONFIGS["C3"], FusedAdam, full=full, sh_factored=shf)
for _ in range(10):
    ti()
torch.cuda.synchronize()
n = 4
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(n):
        ti()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=4):
    dt = float(getattr(e, "self_device_time_total", 0.0))
    if e.key.startswith("aten::") and dt > 0:
        st = [x for x in (e.stack or []) if "dist-packages" not in x and "site-packages" not in x and "<built-in" not in x][:2]
        rows.append((dt / n, e.count / n, e.key, " <- ".join(x.strip()[-80:] for x in st)))
for dt, cnt, name, where in sorted(rows, reverse=True)[:40]:
    print("%7.1f us x%4.1f  %-28s %s" % (dt, cnt, name, where))
