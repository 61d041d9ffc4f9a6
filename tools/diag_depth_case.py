"""Replays one case of tools/fuzz_depth_batch.py (W H n_views learnt L) and says, per view, where the depth-only pass and the oracle part: how many pixels, in which tiles,
and what the two sides hold there.  usage: python tools/diag_depth_case.py W H n learnt L [numpy]"""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from ibgs_amd import renderer, synthetic as syn
from tests.test_gpu_depth_batch import _setup
from tests import hipref
W, H, n, learnt, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4])), int(sys.argv[5])
dev, pc, cams, scene, pipe, args, bg = _setup(3000, W, H, max(n, 3), seed=W + n)
views = cams[:n]
if n >= 3:
    views[1].FoVx *= 0.8; views[1].FoVy *= 0.8
g = dict(pc._gnp)
if "numpy" not in sys.argv:          # (default: the oracle is fed the model's activated values, like the kernels; `numpy` as a sixth argument: the arrays the model was built from)
    with torch.no_grad():
        g["opacities"] = pc.get_opacity.cpu().numpy().reshape(np.asarray(g["opacities"]).shape); g["scales"] = pc.get_scaling.cpu().numpy(); g["rotations"] = pc.get_rotation.cpu().numpy()
with torch.no_grad():
    for v, cam in enumerate(views):
        a = renderer.render_depth(cam, pc, scene, pipe, args, bg, learnt, 3, L).cpu().numpy().reshape(H, W)
        camd = {"viewmatrix": cam.world_view_transform.cpu().numpy(), "campos": cam.camera_center.cpu().numpy()}
        am = syn.plane_all_map(g["means3D"], g["scales"], g["rotations"], camd, normal=g["normal"] if learnt else None, offset=g["offset"] if learnt else None)
        inp = {"means3D": g["means3D"], "shs": g["shs"], "opacities": g["opacities"], "scales": g["scales"], "rotations": g["rotations"],
               "all_map": am, "W": W, "H": H, "tanfovx": math.tan(cam.FoVx * 0.5), "tanfovy": math.tan(cam.FoVy * 0.5),
               "viewmatrix": camd["viewmatrix"], "projmatrix": cam.full_proj_transform.cpu().numpy(), "campos": camd["campos"],
               "bg": np.zeros(3, np.float32), "sh_degree": 1, "render_depth_only": True, "buffer_length": L}
        for cull in (False, True):
            ref = oracle.forward(inp, cull=cull)
            r = ref["median_depth"].reshape(H, W)
            off = np.abs(a - r) > 1e-3 * (1 + np.abs(r))
            ys, xs = np.nonzero(off)
            tiles = sorted(set(zip((ys // 16).tolist(), (xs // 16).tolist())))
            print("view %d oracle cull=%d: %d pixels off (%.2e), in %d tiles %s | R oracle %d" % (v, cull, int(off.sum()), float(off.mean()), len(tiles), tiles[:12], ref["num_rendered"]))
            if off.sum() and not cull:
                for (y, x) in list(zip(ys.tolist(), xs.tolist()))[:6]:
                    print("     pixel (%d, %d): HIP %.6g oracle %.6g" % (x, y, a[y, x], r[y, x]))
                if off.sum() > 10:          # the same inputs through the plain op (explicit all_map), lists against the oracle's
                    with torch.enable_grad():
                        outs, lv, _ = hipref.run_forward(inp)
                    ist = hipref.internal_state(outs, inp)
                    b = outs["median_depth"].cpu().numpy().reshape(H, W)
                    print("     plain op with the explicit plane map: %d pixels differ from render_depth's, %d from the oracle's | R %d vs %d, lists equal %s, radii equal %s"
                          % (int((np.abs(b - a) > 0).sum()), int((np.abs(b - r) > 1e-3 * (1 + np.abs(r))).sum()), ist["R"], ref["num_rendered"],
                             ist["R"] == ref["num_rendered"] and np.array_equal(ist["point_list"], ref["point_list"]), np.array_equal(outs["radii"].cpu().numpy(), ref["radii"])))
                    rh, ro = ist["ranges"], np.asarray(ref["ranges"]).reshape(-1, 2)
                    bad_t = np.nonzero((rh != ro).any(1))[0]
                    print("     tiles whose range differs: %d %s" % (len(bad_t), bad_t[:12].tolist()))
                    # whose plane do the off pixels show?  (forward.cu:524-530: depth = -dist / (n . ray + 1e-8), ray = ((x - W/2) / fx, (y - H/2) / fy, 1))
                    fx, fy = W / (2.0 * inp["tanfovx"]), H / (2.0 * inp["tanfovy"])
                    amd = am.astype(np.float64)
                    for (y, x) in list(zip(ys.tolist(), xs.tolist()))[:4] + list(zip(ys.tolist(), xs.tolist()))[-2:]:
                        ray = np.array([(x - W / 2.0) / fx, (y - H / 2.0) / fy, 1.0])
                        t = (y // 16) * ((W + 15) // 16) + x // 16
                        ids = ist["point_list"][ist["ranges"][t, 0]:ist["ranges"][t, 1]]
                        nr = amd[ids, :3] @ ray
                        dep = -amd[ids, 4] / (nr + 1e-8)
                        io, ih = int(np.argmin(np.abs(dep - r[y, x]))), int(np.argmin(np.abs(dep - a[y, x])))
                        print("     pixel (%d, %d): oracle %.6g = Gaussian %d (n . ray %.3e, dist %.3e, list position %d); render_depth %.6g closest to Gaussian %d (n . ray %.3e, dist %.3e, position %d, its depth by this formula %.6g)"
                              % (x, y, r[y, x], ids[io], nr[io], amd[ids[io], 4], io, a[y, x], ids[ih], nr[ih], amd[ids[ih], 4], ih, dep[ih]))
                    nc_h, nc_o = ist["n_contrib"].reshape(H, W), np.asarray(ref["n_contrib"]).reshape(H, W)
                    print("     n_contrib differs on %d pixels; on the off pixels: HIP %s oracle %s" % (int((nc_h != nc_o).sum()), nc_h[off][:8].tolist(), nc_o[off][:8].tolist()))
