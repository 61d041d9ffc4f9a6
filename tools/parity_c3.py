"""Full-size parity datapoint (not in the test suite: the oracle needs ~10 s on the box's host cores): HIP vs oracle on the
C3 bench workload -- image L1 / max / PSNR difference, per-pixel counter agreement, gradient relative L2.

python tools/parity_c3.py                 the near-isotropic bench scene, init and trained-like opacities (profiles/r02_parity_c3.txt)
python tools/parity_c3.py plane needle    anisotropic variants of the same scene (synthetic.make_gaussians), with the oracle's own
                                          fma / no-fma difference beside every number, its count of `power > 0` skips, and both fp32 sides against
                                          the float64 build of the oracle (the arbiter: |HIP - f64| against |oracle fp32 - f64| per gradient)
python tools/parity_c3.py [geo] trainer   the scene of bench.py's `trained_geo` line (plane-like, heavy-tailed, clustered, trained opacities)
python tools/parity_c3.py geo plane       the geo path (4 sources, L = 4; source images random, source depths the oracle's own depth-only
                                          renders at a quarter of the views' resolution upsampled -- the oracle walks 2 M pixels per pass)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from ibgs_amd import synthetic as syn
from tests import hipref
from tests.metrics import l1, psnr, rel_l2

c = syn.CONFIGS["C3"]
modes = sys.argv[1:]
GEO = "geo" in modes
modes = [m for m in modes if m != "geo"]
cases = [(None, "init"), (None, "trained")] if not modes else [(m, "trained") for m in modes]
names = {"dL_dmeans3D": "means3D", "dL_dsh": "shs", "dL_dopacity": "opacities", "dL_dscales": "scales", "dL_drotations": "rotations", "dL_dmeans2D": "means2D"}
for aniso, opacity in cases:
    if aniso == "trainer":          # the `trained_geo` scene of bench.py: plane-like, log-normal sizes (sigma 1), 30 % in one blob, trained opacities
        inp = syn.make_scene(c["P"], c["W"], c["H"], sh_degree=3, seed=c["seed"], opacity=opacity, anisotropy="plane", scale_sigma=1.0, cluster=0.3, with_planes=GEO)
    else:
        inp = syn.make_scene(c["P"], c["W"], c["H"], sh_degree=3, seed=c["seed"], opacity=opacity, anisotropy=aniso, with_planes=GEO)
    if GEO:
        W, H = c["W"], c["H"]
        srcs = [syn.make_camera(W, H, azimuth_deg=a) for a in (7.0, -7.0, 14.0, -14.0)]
        r2s, scp = syn.ref_to_src(inp["_cam"], srcs)
        deps = []
        for s_ in srcs:          # plausible source depths: the oracle's depth-only render of each source view
            d = dict(inp); d.update(viewmatrix=s_["viewmatrix"], projmatrix=s_["projmatrix"], campos=s_["campos"], render_depth_only=True, buffer_length=4,
                                    all_map=syn.plane_all_map(inp["means3D"], inp["scales"], inp["rotations"], s_))
            deps.append(oracle.forward(d)["median_depth"])
        inp.update(render_geo=True, n_src=4, buffer_length=4, ref_to_src=r2s, src_cam_pos=scp, depth_thr=0.05,
                   src_images=np.random.default_rng(5).uniform(0, 1, (4, 3, H, W)).astype(np.float32), src_depths=np.stack(deps).astype(np.float32))
    t0 = time.time(); ref = oracle.forward(inp, cull=True); t1 = time.time()
    skips_f = oracle.power_skips()[0]
    g = np.random.default_rng(1).standard_normal((3, c["H"], c["W"])).astype(np.float32)
    gn = gd = gw = None
    if GEO:
        r_ = np.random.default_rng(9)
        gn = r_.standard_normal((3, c["H"], c["W"])).astype(np.float32); gd = r_.standard_normal((1, c["H"], c["W"])).astype(np.float32)
        gw = r_.standard_normal((15, c["H"], c["W"])).astype(np.float32)
    rb = oracle.backward(inp, ref, g, gn, gd, gw); t2 = time.time()
    outs, lv, _ = hipref.run_forward(inp)
    ist = hipref.internal_state(outs, inp)
    col = outs["color"].detach().cpu().numpy()
    loss = (outs["color"] * torch.as_tensor(g, device="cuda")).sum()
    if GEO:
        loss = loss + (outs["normal_map"] * torch.as_tensor(gn, device="cuda")).sum() + (outs["median_depth"] * torch.as_tensor(gd, device="cuda")).sum() \
            + (outs["warped_image"] * torch.as_tensor(gw, device="cuda")).sum()
        names["dL_dall_map"] = "all_map"
        o = {k: v.detach().cpu().numpy() for k, v in outs.items()}
        va = np.cumprod(ist["valid_idx"] != -1, axis=0) > 0; vb = np.cumprod(ref["valid_src_idx"] != -1, axis=0) > 0
        same = np.all((va == vb) & (~va | (ist["valid_idx"] == ref["valid_src_idx"])), axis=0).reshape(c["H"], c["W"])
        print("   geo: valid-source sets equal on %.5f of pixels (%d differ), a first source valid on %.3f; median buffer windows equal on %.5f"
              % (same.mean(), int((~same).sum()), (ref["valid_src_idx"][0] >= 0).mean(),
                 ((ist["low_high"][:, 0] == ref["cache_low"]) & (ist["low_high"][:, 1] == ref["cache_high"])).mean()))
        for k in ("normal_map", "median_depth", "warped_image", "cam_feat", "camera_ray", "min_depth_diff"):
            dd = np.abs(o[k] - ref[k])[:, same]
            print("        %-14s mean |d| %.2e (rel %.2e) max %.2e" % (k, dd.mean(), dd.mean() / (np.abs(ref[k][:, same]).mean() + 1e-12), dd.max()))
    loss.backward()
    tgt = np.random.default_rng(2).random(col.shape).astype(np.float32)
    co = ref["conic_opacity"][ref["radii"] > 0]
    asp = (co[:, 0] * co[:, 2]) / np.maximum(co[:, 0] * co[:, 2] - co[:, 1] ** 2, 1e-30)
    print("C3 anisotropy=%s opacity=%s: oracle fwd %.1f s bwd %.1f s | R %d == %d: %s | lists equal: %s" % (aniso, opacity, t1 - t0, t2 - t1, ist["R"], ref["num_rendered"],
          ist["R"] == ref["num_rendered"], np.array_equal(ist["point_list"], ref["point_list"])))
    print("   footprints with a c / det > 25 (2D aspect beyond ~10:1): %.2f %%, > 1e3: %.3f %%; near-singular (reference-expression branch): %d; oracle `power > 0` skips fwd %d bwd %d"
          % (100 * (asp > 25).mean(), 100 * (asp > 1e3).mean(), int((co[:, 1] ** 2 > np.float32(0.99999) * co[:, 0] * co[:, 2]).sum()), skips_f, oracle.power_skips()[1]))
    print("   image: mean L1 %.3e  max |d| %.3e  PSNR(vs common target) HIP %.4f dB oracle %.4f dB | n_contrib equal on %.5f of pixels"
          % (l1(col, ref["color"]), np.abs(col - ref["color"]).max(), psnr(col, tgt)[0], psnr(ref["color"], tgt)[0],
             (ist["n_contrib"] == ref["n_contrib"]).mean()))
    print("   grads rel L2:", {k: float("%.2e" % rel_l2(lv[v].grad.cpu().numpy().reshape(np.asarray(rb[k]).shape), rb[k])) for k, v in names.items()})
    if aniso is not None:
        with oracle.variant("fma"):
            r1 = oracle.forward(inp, cull=True); b1 = oracle.backward(inp, r1, g, gn, gd, gw)
        print("   the oracle against its own fma-contracted build: image mean L1 %.3e max %.3e, n_contrib equal on %.5f, lists equal: %s"
              % (l1(r1["color"], ref["color"]), np.abs(r1["color"] - ref["color"]).max(), (r1["n_contrib"] == ref["n_contrib"]).mean(),
                 r1["num_rendered"] == ref["num_rendered"] and np.array_equal(r1["point_list"], ref["point_list"])))
        print("   ... grads rel L2:", {k: float("%.2e" % rel_l2(b1[k], rb[k])) for k in names})
        # the float64 build of the same C source as the arbiter (round 4): how far is each fp32 side from it?
        with oracle.variant("f64"):
            r64 = oracle.forward(inp, cull=True); b64 = oracle.backward(inp, r64, g, gn, gd, gw)
        hip = {k: lv[v].grad.cpu().numpy().reshape(np.asarray(rb[k]).shape).astype(np.float64) for k, v in names.items()}
        print("   against the float64 build: image mean L1 HIP %.3e oracle fp32 %.3e" % (l1(col, r64["color"]), l1(ref["color"], r64["color"])))
        print("   ... grads rel L2, HIP | oracle fp32 (ratio):", {k: "%.1e | %.1e (%.2f)" % (rel_l2(hip[k], b64[k]), rel_l2(rb[k], b64[k]), rel_l2(hip[k], b64[k]) / max(rel_l2(rb[k], b64[k]), 1e-30)) for k in names})
    del outs, lv
    torch.cuda.empty_cache()
