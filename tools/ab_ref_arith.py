"""A/B of how the blend backward sums the pairs of near-singular conics (csrc/render_bwd.hip: RA_LFORM -- the default -- against RA_ASSOC = IBGS_FLAG_REF_ARITH, the
reference's own association) on the cases the round-5 sweeps left outside the float64 arbiter's bar: tools/fuzz_parity.py 120 9103 - trained, cases 9 and 117;
tools/fuzz_fused.py 100 9105, cases 1 and 38.  Deterministic backward unless AB_ATOMICS=1 (then AB_REPEAT runs: the float atomics' own scatter).
    [AB_REF_ARITH=1] [AB_ATOMICS=1 AB_REPEAT=8] [IBGS_LIB=<another build>] python tools/ab_ref_arith.py [parity|fused|all]
(the oracle's builds of a case are cached under /tmp between runs; columns: HIP | oracle fp32 | its fma twin | the oracle with float sums, as distances from the float64 build)"""
import os, pickle, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from ibgs_amd import rasterizer
from tests import fuzz_cases as fc, hipref
from tests.metrics import l1, rel_l2

which = sys.argv[1] if len(sys.argv) > 1 else "all"
rasterizer.DETERMINISTIC = os.environ.get("AB_ATOMICS") != "1"
rasterizer.REF_ARITH = os.environ.get("AB_REF_ARITH") == "1"
REPEAT = int(os.environ.get("AB_REPEAT", "1"))
print("# sums of near-singular conics: %s; deterministic=%s lib=%s" % ("the reference's association (IBGS_FLAG_REF_ARITH)" if rasterizer.REF_ARITH else "l-form (default)", rasterizer.DETERMINISTIC,
                                                                        os.environ.get("IBGS_LIB", "current")), flush=True)


def cached(name, fn):
    path = "/tmp/abra_%s.pkl" % name
    if os.path.exists(path):
        return pickle.load(open(path, "rb"))
    v = fn()
    pickle.dump(v, open(path, "wb"))
    return v


if which in ("parity", "all"):
    for seed, index in ((9103, 9), (9103, 117)):
        c, inp, g = fc.parity_case(seed, index, "trained")
        ob = cached("p%d_%d" % (seed, index), lambda: {b: fb[1] for b, fb in fc.oracle_builds(inp, g, c["cull"]).items()})
        ob = {b: (None, v) for b, v in ob.items()}
        rasterizer.TILE_CULL = c["cull"]; rasterizer.WAVE_SHAPE = c["wave_shape"]
        for rep in range(REPEAT):
            outs, lv, _ = hipref.run_forward(inp)
            loss = (outs["color"] * torch.as_tensor(g["color"], device="cuda")).sum()
            if c["geo"]:
                loss = loss + (outs["normal_map"] * torch.as_tensor(g["normal_map"], device="cuda")).sum() + (outs["median_depth"] * torch.as_tensor(g["median_depth"], device="cuda")).sum() \
                    + (outs["warped_image"] * torch.as_tensor(g["warped_image"], device="cuda")).sum()
            loss.backward()
            hip = {v: lv[v].grad.cpu().numpy() for v in list(fc.ALL_GRADS.values()) + (["all_map"] if c["geo"] else [])}
            pairs = fc.arbiter_pairs(hip, ob, c["geo"])
            print("parity %d/%d (%s; P %d %dx%d geo %d): ratio %.2f | " % (seed, index, c.get("knobs"), c["P"], c["W"], c["H"], c["geo"], fc.arbiter_ratio(pairs))
                  + ", ".join("%s %.1e|%.1e|%.1e|%.1e" % ((v,) + p) for v, p in pairs.items()), flush=True)

if which in ("fused", "all"):
    from tests.test_gpu_fused_planes import _oracle_chain, _run, _scene
    for seed, index in ((9105, 1), (9105, 38)):
        c = fc.fused_case(seed, index)
        dev, g, cams, scene, pipe, args, bg = _scene(P=c["P"], W=c["W"], H=c["H"], seed=c["seed"])
        o_fus, g_fus = _run(True, c["learnt"], g, dev, cams, scene, pipe, args, bg)

        def both():
            _, g32 = _oracle_chain(c["learnt"], g, dev, cams, scene, bg)
            with oracle.variant("f64"):
                _, g64 = _oracle_chain(c["learnt"], g, dev, cams, scene, bg)
            with oracle.variant("fma"):
                _, gfm = _oracle_chain(c["learnt"], g, dev, cams, scene, bg)
            with oracle.variant("acc32"):
                _, gac = _oracle_chain(c["learnt"], g, dev, cams, scene, bg)
            return g32, g64, gfm, gac
        g32, g64, gfm, gac = cached("f%d_%d" % (seed, index), both)
        names = ["_xyz", "_rotation", "_scaling", "_opacity", "_features_dc"] + (["_normal", "_offset"] if c["learnt"] else [])
        e = {n: (rel_l2(g_fus[n], g64[n]), rel_l2(g32[n], g64[n]), rel_l2(gfm[n], g64[n]), rel_l2(gac[n], g64[n])) for n in names if g64[n] is not None and np.abs(g64[n]).sum() > 0}
        ratio = max(p[0] / max(5e-3 / 2.0, max(p[1:])) for p in e.values())
        print("fused %d/%d (P %d %dx%d learnt %d): ratio %.2f | " % (seed, index, c["P"], c["W"], c["H"], c["learnt"], ratio) + ", ".join("%s %.1e|%.1e|%.1e|%.1e" % ((n,) + p) for n, p in e.items()), flush=True)
