"""Randomised parity sweep on the MI355X (not part of the test suite): random sizes / degrees / modes against the oracle.
python tools/fuzz_parity.py [n_cases] [seed] [only] [big]
("big" as the fourth argument: frames of 1 000 ... 3 600 tiles, the library's own choice of wave shape -- the hybrid kernels' range; "trained": those
frames with random anisotropy / cluster / log-normal sizes, synthetic.make_gaussians' knobs)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from ibgs_amd import rasterizer, synthetic as syn
from tests import fuzz_cases as fc, hipref
from tests.metrics import l1, rel_l2

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
only = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] not in ("-", "none") else None          # replay ONE case of the sequence and print where the two sides part
BIG = sys.argv[4] if len(sys.argv) > 4 and sys.argv[4] in ("big", "trained") else None          # "trained": big frames + random anisotropy / cluster / size spread
rasterizer.DETERMINISTIC = os.environ.get("FUZZ_DETERMINISTIC") == "1"          # (fixed summation order in the backward: separates float-atomic scatter from systematic distance)
worst = {"color": 0.0, "grad": 0.0, "ncontrib": 1.0}
bad = 0
for case in range(n_cases):
    # the draws live in tests/fuzz_cases.py (tests/test_gpu_fuzz_pins.py replays single cases of a sweep from there)
    c = fc.draw_parity(rng, BIG)
    P, W, H, deg, geo, sseed = c["P"], c["W"], c["H"], c["deg"], c["geo"], c["sseed"]
    rasterizer.WAVE_SHAPE = c["wave_shape"]
    if only is not None and case != only:          # consume the same draws as the full run
        fc.draw_parity_grads(rng, c)
        continue
    inp = fc.build_parity(c)
    if BIG == "trained":
        print("     case %d: %s" % (case, c["knobs"]))
    cull = c["cull"]
    rasterizer.TILE_CULL = cull
    ref = oracle.forward(inp, cull=cull)
    outs, lv, _ = hipref.run_forward(inp)
    ist = hipref.internal_state(outs, inp)
    o = hipref.to_np(outs)
    ok = ist["R"] == ref["num_rendered"] and np.array_equal(ist["point_list"], ref["point_list"]) and np.array_equal(o["radii"], ref["radii"])
    dc = l1(o["color"], ref["color"])
    nc = float((ist["n_contrib"] == ref["n_contrib"]).mean())
    gd = fc.draw_parity_grads(rng, c)
    g = gd["color"]
    if geo:      # every differentiable geo output takes part
        gn, gdp, gw = gd["normal_map"], gd["median_depth"], gd["warped_image"]
        ((outs["color"] * torch.as_tensor(g, device="cuda")).sum() + (outs["normal_map"] * torch.as_tensor(gn, device="cuda")).sum()
         + (outs["median_depth"] * torch.as_tensor(gdp, device="cuda")).sum() + (outs["warped_image"] * torch.as_tensor(gw, device="cuda")).sum()).backward()
        rb = oracle.backward(inp, ref, g, gn, gdp, gw)
    else:
        (outs["color"] * torch.as_tensor(g, device="cuda")).sum().backward()
        rb = oracle.backward(inp, ref, g)
    gr = 0.0; explained = False
    names = {"dL_dmeans3D": "means3D", "dL_dopacity": "opacities", "dL_dscales": "scales"}
    if geo:
        names["dL_dall_map"] = "all_map"
    for k, v in names.items():
        a = lv[v].grad.cpu().numpy().reshape(np.asarray(rb[k]).shape)
        if np.abs(rb[k]).sum() > 0:
            gr = max(gr, float(rel_l2(a, rb[k])))
    if gr > 1e-3:          # explain it: every gradient, beside the oracle's own fma / no-fma difference on the same case
        with oracle.variant("fma"):
            r1 = oracle.forward(inp, cull=cull)
            b1 = oracle.backward(inp, r1, g, gn, gdp, gw) if geo else oracle.backward(inp, r1, g)
        allk = {"dL_dmeans3D": "means3D", "dL_dmeans2D": "means2D", "dL_dopacity": "opacities", "dL_dsh": "shs", "dL_dscales": "scales", "dL_drotations": "rotations"}
        if geo: allk["dL_dall_map"] = "all_map"
        # ill-conditioned Gaussians (needles, giant planes: conics within 1e-4 of singular): a gradient counts as explained when the kernels are no farther
        # from the oracle than three times what the oracle's own fma / no-fma builds differ by (the bar of tests/test_gpu_anisotropic.py before the float64 arbiter)
        explained = all(rel_l2(lv[v].grad.cpu().numpy().reshape(np.asarray(rb[k]).shape), rb[k]) <= max(1e-3, 3.0 * rel_l2(b1[k], rb[k])) for k, v in allk.items() if np.abs(rb[k]).sum() > 0)
        if not explained:          # the arbiter of tests/test_gpu_anisotropic.py: the float64 build of the same C source.  HIP may be at most twice as far from it as the fp32 oracle is
            with oracle.variant("f64"):
                r64 = oracle.forward(inp, cull=cull)
                b64 = oracle.backward(inp, r64, g, gn, gdp, gw) if geo else oracle.backward(inp, r64, g)
            f64 = lambda k: np.asarray(b64[k]).reshape(np.asarray(rb[k]).shape)
            with oracle.variant("acc32"):          # the fp32 oracle with its gradient sums kept in float, as the reference's atomicAdd keeps them (the oracle proper sums in double)
                r32 = oracle.forward(inp, cull=cull)
                b32 = oracle.backward(inp, r32, g, gn, gdp, gw) if geo else oracle.backward(inp, r32, g)
            pairs = {v: (rel_l2(lv[v].grad.cpu().numpy().reshape(np.asarray(rb[k]).shape), f64(k)), rel_l2(rb[k], f64(k)), rel_l2(np.asarray(b1[k]).reshape(np.asarray(rb[k]).shape), f64(k)),
                         rel_l2(np.asarray(b32[k]).reshape(np.asarray(rb[k]).shape), f64(k)))
                     for k, v in allk.items() if np.abs(rb[k]).sum() > 0}
            # both fp32 builds of the oracle are evaluations of the reference's algorithm (nvcc contracts by default): the farther of the two sets the scale
            explained = all(a64 <= max(1e-3, 2.0 * max(o64, t64, s64)) for a64, o64, t64, s64 in pairs.values())
            print("     case %d vs the float64 build (HIP | oracle fp32 | its fma twin | oracle with float sums): " % case + ", ".join("%s %.1e|%.1e|%.1e|%.1e" % (v, p[0], p[1], p[2], p[3]) for v, p in pairs.items())
                  + ("  -> within the arbiter's bar" if explained else "  -> OUTSIDE"))
        print("     case %d detail (HIP vs oracle | oracle fma vs no-fma): " % case + ", ".join(
            "%s %.1e|%.1e" % (v, rel_l2(lv[v].grad.cpu().numpy().reshape(np.asarray(rb[k]).shape), rb[k]), rel_l2(b1[k], rb[k])) for k, v in allk.items() if np.abs(rb[k]).sum() > 0))
        k = "dL_dall_map" if geo else "dL_dscales"
        a = lv[names.get(k, "scales")].grad.cpu().numpy().reshape(np.asarray(rb[k]).shape); e = np.abs(a - rb[k]).reshape(a.shape[0], -1).sum(1)
        i = int(np.argmax(e))
        co = ref["conic_opacity"][i]
        print("     worst Gaussian %d: %s HIP %s oracle %s twin %s | 1 - b^2/(ac) = %.2e, radius %d" % (i, k, a[i].ravel()[:5], np.asarray(rb[k])[i].ravel()[:5], np.asarray(b1[k])[i].ravel()[:5], 1 - co[1] ** 2 / max(co[0] * co[2], 1e-30), ref["radii"][i]))
    if only is not None:
        HW = H * W
        if not ok:          # where the lists part
            print("     R HIP %d oracle %d; radii equal %s; tiles_touched equal %s" % (ist["R"], ref["num_rendered"], np.array_equal(o["radii"], ref["radii"]), np.array_equal(ist["tiles"], ref["tiles_touched"])))
            dt = np.flatnonzero(ist["tiles"] != ref["tiles_touched"])
            for i in dt[:6]:
                print("     Gaussian %d: tiles HIP %d oracle %d, rect %s, radius %d, mean2D %s conic_opacity %s tmask %s" % (i, ist["tiles"][i], ref["tiles_touched"][i], ref["rect4"][i], ref["radii"][i], ref["means2D"][i], ref["conic_opacity"][i], ref["tmask"][i]))
            if np.array_equal(ist["ranges"], ref["ranges"]):
                d = np.flatnonzero(ist["point_list"] != ref["point_list"])
                print("     ranges equal; %d list positions differ, first %s" % (d.size, d[:8]))
                for j in d[:4]:
                    t = int(np.searchsorted(ref["ranges"][:, 1], j, side="right"))
                    a, b = int(ist["point_list"][j]), int(ref["point_list"][j])
                    print("       pos %d (tile %d): HIP id %d depth %.9g | oracle id %d depth %.9g" % (j, t, a, ref["depths"][a], b, ref["depths"][b]))
            else:
                dr = np.flatnonzero((ist["ranges"] != ref["ranges"]).any(axis=1))
                print("     ranges differ at %d tiles, first %s" % (dr.size, dr[:8]))
                for t in dr[:3]:
                    ha = set(ist["point_list"][ist["ranges"][t, 0]:ist["ranges"][t, 1]].tolist()); oa = set(ref["point_list"][ref["ranges"][t, 0]:ref["ranges"][t, 1]].tolist())
                    print("       tile %d (x %d y %d): HIP %d entries, oracle %d; only HIP %s only oracle %s" % (t, t % ((W + 15) // 16), t // ((W + 15) // 16), len(ha), len(oa), sorted(ha - oa)[:5], sorted(oa - ha)[:5]))
                    for i in (sorted(ha - oa) + sorted(oa - ha))[:3]:
                        print("         Gaussian %d: rect %s radius %d mean2D %s conic_opacity %s tmask %s tiles HIP %d oracle %d" % (i, ref["rect4"][i], ref["radii"][i], ref["means2D"][i], ref["conic_opacity"][i], ref["tmask"][i], ist["tiles"][i], ref["tiles_touched"][i]))
        print("     n_contrib differs at pixels", np.flatnonzero(ist["n_contrib"] != ref["n_contrib"])[:10], "final_T max diff %.2e" % np.abs(ist["final_T"] - ref["final_T"]).max())
        dcol = np.abs(o["color"] - ref["color"]).max(0).ravel()
        for pix in np.argsort(-dcol)[:3]:
            print("     colour diff %.2e at pixel %d (x %d y %d): n_contrib HIP %d oracle %d, final_T %.6g / %.6g" % (dcol[pix], pix, pix % W, pix // W, ist["n_contrib"][pix], ref["n_contrib"][pix], ist["final_T"][pix], ref["final_T"][pix]))
        if geo:
            for k in ("median_depth", "normal_map", "warped_image", "min_depth_diff"):
                dk = np.abs(o[k] - ref[k]).max(0).ravel(); pix = int(np.argmax(dk))
                print("     %s max diff %.2e at pixel %d; low/high HIP %s oracle (%d, %d); sum_w %.6g / %.6g; valid HIP %s oracle %s" % (k, dk[pix], pix, ist["low_high"][pix], ref["cache_low"][pix], ref["cache_high"][pix],
                      ist["sum_w"][pix], ref["cache_sum_w"][pix], ist["valid_idx"][:, pix], ref["valid_src_idx"][:, pix]))
    worst["color"] = max(worst["color"], dc); worst["grad"] = max(worst["grad"], gr); worst["ncontrib"] = min(worst["ncontrib"], nc)
    flag = ok and dc < 1e-5 and (gr < (2e-2 if geo else 5e-3) or explained) and nc > 0.995
    bad += not flag
    print("%s case %2d: P=%5d %3dx%3d deg=%d geo=%d cull=%d shape=%-8s R=%7d | lists %s colour L1 %.1e n_contrib eq %.4f grad relL2 %.1e"
          % ("ok  " if flag else "FAIL", case, P, W, H, deg, geo, cull, rasterizer.WAVE_SHAPE, ist["R"], ok, dc, nc, gr), flush=True)
print("worst:", worst, "failures:", bad)
sys.exit(1 if bad else 0)
