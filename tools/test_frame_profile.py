"""What the test-time frame of bench.py (`test_frame`: 4 source depth maps in one batched depth-only pass + the geo pass, no gradients) spends its GPU time on:
every kernel of 6 frames grouped by name.  usage: python tools/test_frame_profile.py"""
import os, sys, collections, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = sys.argv[:1]
import bench
from ibgs_amd import renderer, simple_scene, synthetic as syn
from torch.profiler import ProfilerActivity, profile
dev = torch.device("cuda", 0)
c = syn.CONFIGS["C3"]
P, W, H = c["P"], c["W"], c["H"]
g = syn.make_gaussians(P, c["seed"], sh_degree=3, max_coeffs=16, opacity="init")
rng = np.random.default_rng(0)
g["normal"] = rng.normal(size=(P, 3)).astype(np.float32); g["offset"] = (0.01 * rng.normal(size=(P, 1))).astype(np.float32)
pc = simple_scene.SimpleGaussians(g, sh_degree=3, device=dev)
cams = simple_scene.orbit_cameras(W, H, n_views=8, device=dev, nearest=4)
scene = simple_scene.SimpleScene(cams, images=torch.rand(8, 3, H, W, device=dev), device=dev)
pipe, args = simple_scene.default_pipe(), simple_scene.default_args()
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    fn = lambda: renderer.render(cams[0], pc, scene, pipe, args, bg, True, 4, 4, render_geo=True, do_render_src_depth=True, return_depth_normal=False)
    wall = bench.timed_wall_ms(fn, 16, warmup=5)
    n = 6
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        a = agg[e.name[:110]]; a[0] += 1; a[1] += float(getattr(e, "device_time_total", 0.0))
tot = sum(v[1] for v in agg.values()); cnt = sum(v[0] for v in agg.values())
print("wall %.3f ms per frame; %d kernels / copies per frame, %.3f ms in them" % (wall, cnt // n, tot / n * 1e-3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%7.1f us  x%5.1f  %s" % (v[1] / n, v[0] / n, k))
