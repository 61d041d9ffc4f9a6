"""How long the reference's optimiser step takes at C3 size on the MI355X (row 4 of SURVEY 8(f)): torch.optim.Adam over the
eight Gaussian parameter groups (scene/gaussian_model.py:227-236), default (foreach) versus fused=True."""
import time, torch
P = 1_000_000
dev = "cuda"
shapes = {"xyz": (P, 3), "f_dc": (P, 1, 3), "f_rest": (P, 15, 3), "opacity": (P, 1), "scaling": (P, 3), "rotation": (P, 4), "normal": (P, 3), "offset": (P, 1)}
for fused in (False, True):
    params = [{"params": [torch.nn.Parameter(torch.randn(s, device=dev))], "lr": 1e-3, "name": n} for n, s in shapes.items()]
    try:
        opt = torch.optim.Adam(params, lr=0.0, eps=1e-15, fused=fused)
    except Exception as ex:
        print("fused=%s unavailable: %s" % (fused, ex)); continue
    for g in params:
        g["params"][0].grad = torch.randn_like(g["params"][0])
    for _ in range(3): opt.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): opt.step()
    torch.cuda.synchronize()
    print("Adam fused=%s: %.3f ms per step (63 floats x %d Gaussians)" % (fused, (time.perf_counter() - t0) / 20 * 1e3, P))

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibgs_amd.optim import FusedAdam
params = [{"params": [torch.nn.Parameter(torch.randn(s, device=dev))], "lr": 1e-3, "name": n} for n, s in shapes.items()]
opt = FusedAdam(params, lr=0.0, eps=1e-15)
for g in params:
    g["params"][0].grad = torch.randn_like(g["params"][0])
for _ in range(3): opt.step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): opt.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("ibgs_amd.optim.FusedAdam (one launch): %.3f ms per step = %.2f TB/s of the 7 floats/parameter it must move" % (dt * 1e3, 7 * 4 * 63 * P / dt / 1e12))
