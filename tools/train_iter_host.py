"""Host time of one trainer-shaped iteration (bench.py: TrainIteration): run at C1 size (10 000 Gaussians, 400 x 400), where the GPU is idle most of the time, the wall
clock per iteration is what the HOST needs to queue it -- the floor under the C3-sized iteration's wall clock whatever the kernels do.  usage: python tools/train_iter_host.py"""
import os, sys, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1]
import bench
from ibgs_amd import synthetic as syn
from ibgs_amd.optim import FusedAdam
dev = torch.device("cuda", 0)
for full, shf in ((False, False), (False, True), (True, False), (True, True)):
    ti = bench.TrainIteration(dev, syn.CONFIGS["C1"], FusedAdam, full=full, sh_factored=shf)
    wall = bench.timed_wall_ms(ti, 200, warmup=30)
    ksum = bench.gpu_kernel_sum_ms(ti, 5)
    print("TrainIteration(full=%s, sh_factored=%s) at C1 size: wall %.3f ms per iteration (kernel sum %.3f ms): the host's share" % (full, shf, wall, ksum if ksum is not None else float("nan")))
    if full and shf:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(100):
            ti()
        torch.cuda.synchronize(); pr.disable()
        st = pstats.Stats(pr); st.sort_stats("tottime")
        st.print_stats(18)
