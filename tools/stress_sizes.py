"""Size stress on the MI355X (not part of the test suite): list invariants, finiteness and timing at P = 5 M (the
reference's max_all_points) and at 4K resolution."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibgs_amd import synthetic as syn
from tests import hipref
from tests.test_gpu_renderer import _check_lists

for P, W, H in ((5_000_000, 1920, 1080), (1_000_000, 3840, 2160), (200_000, 7680, 4320)):
    inp = syn.make_scene(P, W, H, sh_degree=3, seed=4)
    outs, lv, _ = hipref.run_forward(inp)
    ist = hipref.internal_state(outs, inp)
    _check_lists(ist, ist["depths"])
    col = outs["color"]
    assert torch.isfinite(col).all()
    g = torch.randn_like(col)
    (col * g).sum().backward()
    assert all(torch.isfinite(v.grad).all() for k, v in lv.items() if v is not None and v.grad is not None)
    def step():
        for v in lv.values():
            if v is not None: v.grad = None
        o, l2, _ = None, None, None
    torch.cuda.synchronize()
    from ibgs_amd.rasterizer import GaussianRasterizer
    st = hipref.settings_from(inp, "cuda")
    rast = GaussianRasterizer(st)
    def it():
        for v in lv.values():
            if v is not None: v.grad = None
        o = rast(means3D=lv["means3D"], means2D=lv["means2D"], means2D_abs=lv["means2D_abs"], opacities=lv["opacities"], shs=lv["shs"],
                 scales=lv["scales"], rotations=lv["rotations"])
        (o[0] * g).sum().backward()
    for _ in range(2): it()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): it()
    torch.cuda.synchronize()
    print("P=%d %dx%d: R=%d tiles=%d  fwd+bwd %.2f ms  peak mem %.2f GB" % (P, W, H, ist["R"], ((W + 15) // 16) * ((H + 15) // 16),
          (time.perf_counter() - t0) / 5 * 1e3, torch.cuda.max_memory_allocated() / 2**30), flush=True)
    del outs, lv, ist, col, g, rast
    torch.cuda.empty_cache()
