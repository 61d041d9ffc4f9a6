"""Step time of renderer.render() (geo, 4 sources) + backward at C3 size with the plane-map glue in torch
(reference behaviour) versus fused into the preprocess kernels.  Run on the MI355X box: python tools/bench_render_glue.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibgs_amd import renderer, simple_scene, synthetic as syn

P, W, H = 1_000_000, 1920, 1080
dev = torch.device("cuda")
g = syn.make_gaussians(P, 3, sh_degree=3, max_coeffs=16, opacity="init")
rng = np.random.default_rng(0)
g["normal"] = rng.normal(size=(P, 3)).astype(np.float32); g["offset"] = (0.01 * rng.normal(size=(P, 1))).astype(np.float32)
pc = simple_scene.SimpleGaussians(g, sh_degree=3, device=dev)
cams = simple_scene.orbit_cameras(W, H, n_views=8, device=dev, nearest=4)
scene = simple_scene.SimpleScene(cams, images=torch.rand(8, 3, H, W, device=dev), device=dev)
pipe, args = simple_scene.default_pipe(), simple_scene.default_args()
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    for j in cams[0].nearest_id:
        scene.rendered_depth_list[j] = renderer.render_depth(cams[j], pc, scene, pipe, args, bg, True, 4, 4)
tgt = torch.rand(3, H, W, device=dev)
for learnt in (True, False):
    for fused in (False, True):
        renderer.FUSED_PLANE_MAP = fused
        def step():
            pc.zero_grad(set_to_none=True)
            out = renderer.render(cams[0], pc, scene, pipe, args, bg, learnt_normal=learnt, nb_src_frames=4, buffer_length=4,
                                  render_geo=True, return_depth_normal=False)
            loss = (out["render"] - tgt).abs().mean() + out["rendered_normal"].abs().mean() + out["median_intersected_depth"].abs().mean() \
                + (out["warped_image"] - 0.5).abs().mean()
            loss.backward()
        for _ in range(3): step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        with torch.no_grad():
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): renderer.render_depth(cams[1], pc, scene, pipe, args, bg, learnt, 4, 4)
            torch.cuda.synchronize(); dd = (time.perf_counter() - t0) / 10
        print("learnt_normal=%s fused=%s: render()+backward %.3f ms, render_depth() %.3f ms" % (learnt, fused, dt * 1e3, dd * 1e3))

# ---- the source-view depth passes of one test-time frame, looped vs batched ----
renderer.FUSED_PLANE_MAP = True
src = [cams[j] for j in cams[0].nearest_id[:4]]
with torch.no_grad():
    for name, fn in (("4 x render_depth()", lambda: torch.stack([renderer.render_depth(c, pc, scene, pipe, args, bg, True, 4, 4) for c in src])),
                     ("render_depth_batch(4)", lambda: renderer.render_depth_batch(src, pc, scene, pipe, args, bg, True, 4, 4))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        print("%s: %.3f ms" % (name, (time.perf_counter() - t0) / 10 * 1e3))
    for name, batch in (("render() test-time frame, batched source depths", True), ("render() test-time frame, looped source depths", False)):
        import ibgs_amd.renderer as R
        orig = R.render_depth_batch
        if not batch:
            R.render_depth_batch = lambda cs, *a, **k: torch.stack([R.render_depth(c, *a, **k) for c in cs])
        fn = lambda: R.render(cams[0], pc, scene, pipe, args, bg, True, 4, 4, render_geo=True, do_render_src_depth=True, return_depth_normal=False)
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        print("%s: %.3f ms" % (name, (time.perf_counter() - t0) / 10 * 1e3))
        R.render_depth_batch = orig
