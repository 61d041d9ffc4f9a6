"""The geo forward on the trained scene takes 0.32 or 0.36 ms from one run to the next.  Does it depend on WHERE its buffers lie?  One process, the workload
built several times with a dummy allocation of a different size in front each time (the caching allocator then hands out different addresses); per build: the
render_fwd stage time (hipEvent stage timer, 20 forward+backward steps) and the addresses of the tensors the kernel touches.  usage: python tools/geo_fwd_bimodal.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1]
import bench
from ibgs_amd import _lib

dev = torch.device("cuda", 0)
keep = []
for rnd, pad_mb in enumerate((0, 1, 3, 7, 64, 0, 129, 2, 0)):
    torch.cuda.empty_cache()
    pad = torch.empty(pad_mb << 20, dtype=torch.uint8, device=dev) if pad_mb else None
    wl = bench.Workload("C3", 0, dev, "trained", True, True, 1234, cluster=0.3, anisotropy="plane", scale_sigma=1.0)
    for _ in range(5):
        wl.local_step()
    torch.cuda.synchronize()
    _lib.timing_enable(["render_fwd", "render_bwd", "geo_window"]); _lib.timing_collect()
    for _ in range(20):
        wl.local_step()
    torch.cuda.synchronize()
    tm = _lib.timing_collect(); _lib.timing_enable([])
    st = wl.st
    addrs = {"src_images": st.src_images.data_ptr(), "src_depths": st.src_rendered_depths.data_ptr(), "means3D": wl.leaves["means3D"].data_ptr()}
    print("build %d (pad %3d MB): render_fwd %.4f ms  render_bwd %.4f  geo_window %.4f | %s" % (rnd, pad_mb, tm["render_fwd"][0] / 20.0, tm["render_bwd"][0] / 20.0,
          tm["geo_window"][0] / 20.0, " ".join("%s %#x (mod 2M %#x)" % (k, v, v & 0x1FFFFF) for k, v in addrs.items())), flush=True)
    if rnd % 2 == 0:
        keep.append(pad)          # keep some pads alive so that later builds cannot land on the same blocks
    del wl
