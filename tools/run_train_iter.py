"""Runs bench.py's TrainIteration a few times (for rocprofv3 / PMC passes: no torch.profiler inside).  usage: python3 tools/run_train_iter.py [n] [--no-split-sh]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6
nosplit = "--no-split-sh" in sys.argv
sys.argv = sys.argv[:1]
import bench
from ibgs_amd import renderer, synthetic as syn
from ibgs_amd.optim import FusedAdam
renderer.SPLIT_SH = not nosplit
ti = bench.TrainIteration(torch.device("cuda", 0), syn.CONFIGS["C3"], FusedAdam)
for _ in range(n):
    ti()
torch.cuda.synchronize()
