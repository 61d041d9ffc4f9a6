import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1]
import bench
from ibgs_amd import rasterizer
rasterizer.DEPTH_BOUND = True
wl = bench.Workload("C3", 0, torch.device("cuda", 0), "init", False, False, 1234)
for _ in range(12):
    wl.local_step()
torch.cuda.synchronize()
