"""What one trainer-shaped iteration (bench.py: TrainIteration) spends its GPU time on: every kernel the GPU runs during 6 iterations, grouped by name --
count per iteration, microseconds per iteration -- and the wall clock beside their sum.  usage: python tools/train_iter_profile.py [torch_adam] [full] [sh_factored]
(full: train.py's steady-state losses -- normal consistency + multi-view photometric L1 -- instead of L1 on `render` alone: TrainIteration(full=True); sh_factored: the SH coefficients updated straight from
the backward's factors, no dense dL/dsh: TrainIteration(sh_factored=True))"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch_adam = "torch_adam" in sys.argv
full = "full" in sys.argv
shf = "sh_factored" in sys.argv
sys.argv = sys.argv[:1]
import bench
from ibgs_amd import synthetic as syn
from ibgs_amd.optim import FusedAdam
from torch.profiler import ProfilerActivity, profile
dev = torch.device("cuda", 0)
ti = bench.TrainIteration(dev, syn.CONFIGS["C3"], torch.optim.Adam if torch_adam else FusedAdam, full=full, sh_factored=shf)
print("TrainIteration(full=%s, %s%s)" % (full, "torch.optim.Adam" if torch_adam else "FusedAdam", ", sh_factored" if shf else ""))
wall = bench.timed_wall_ms(ti, 16, warmup=10)
n = 6
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(n):
        ti()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        a = agg[e.name[:110]]; a[0] += 1; a[1] += float(getattr(e, "device_time_total", 0.0))
tot = sum(v[1] for v in agg.values()); cnt = sum(v[0] for v in agg.values())
print("wall %.3f ms per iteration; %d kernels / copies per iteration, %.3f ms in them (sum); the library's (ibgs::) %.3f ms in %d launches"
      % (wall, cnt // n, tot / n * 1e-3, sum(v[1] for k, v in agg.items() if "ibgs::" in k) / n * 1e-3, sum(v[0] for k, v in agg.items() if "ibgs::" in k) // n))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%7.1f us  x%5.1f  %s" % (v[1] / n, v[0] / n, k))
