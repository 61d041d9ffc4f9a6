"""The random cases of tools/fuzz_parity.py and tools/fuzz_fused.py as importable generators: the sweeps draw from them and tests/test_gpu_fuzz_pins.py
replays single cases of a sweep by (seed, index) -- the cases the round-5 sweeps left outside the float64 arbiter's bar are pinned there.

A case is everything the sweep draws for it from ONE numpy generator, in the sweep's order; replaying case k consumes the draws of cases 0 .. k - 1."""
import numpy as np

import oracle
from ibgs_amd import synthetic as syn
from tests.metrics import rel_l2


# ---- tools/fuzz_parity.py --------------------------------------------------------------------------------------------------------------------------
def draw_parity(rng, big):
    """The parameters of the next case (big: None | "big" | "trained").  The upstream gradients are drawn by draw_parity_grads afterwards."""
    c = {}
    c["P"] = int(rng.choice([1, 2, 7, 63, 64, 65, 300, 1500, 4000, 9000]))
    c["W"], c["H"] = int(rng.integers(8, 320)), int(rng.integers(8, 240))
    if big:
        c["P"] = int(rng.choice([3000, 9000, 20000, 40000]))
        c["W"], c["H"] = int(rng.integers(480, 1281)), int(rng.integers(360, 721))
    c["deg"] = int(rng.integers(0, 4)); c["geo"] = bool(rng.integers(0, 3) == 0)
    c["opacity"] = str(rng.choice(["init", "trained"])); c["smul"] = float(rng.choice([0.5, 1.0, 2.5]))
    c["wave_shape"] = [None, "tile", "quadrant"][int(rng.integers(0, 3))]
    if big:
        c["wave_shape"] = None
    c["sseed"] = int(rng.integers(0, 10**6))
    c["n_src"], c["Lb"] = (int(rng.integers(1, 6)), int(rng.integers(1, 9))) if c["geo"] else (1, 4)
    c["big"] = big
    c["cull"] = not (big and c["sseed"] % 4 == 0)          # big modes: a quarter of the cases on the reference's AABB lists
    return c


def draw_parity_grads(rng, c):
    H, W = c["H"], c["W"]
    g = {"color": rng.standard_normal((3, H, W)).astype(np.float32)}
    if c["geo"]:      # every differentiable geo output takes part
        g["normal_map"] = rng.standard_normal((3, H, W)).astype(np.float32)
        g["median_depth"] = rng.standard_normal((1, H, W)).astype(np.float32)
        g["warped_image"] = rng.standard_normal((15, H, W)).astype(np.float32)
    return g


def build_parity(c):
    """The oracle-style input dict of a drawn case."""
    from tests.test_gpu_parity import add_sources, scene
    if c["big"] == "trained":          # the trained generator's knobs as well (drawn from a generator of their own: the other modes' sequences stay what they were)
        r2 = np.random.default_rng(c["sseed"])
        aniso = [None, "plane", "needle", "mixed"][int(r2.integers(0, 4))]; cl = float(r2.choice([0.0, 0.3, 0.5])); sig = float(r2.choice([0.0, 1.0]))
        inp = syn.make_scene(c["P"], c["W"], c["H"], sh_degree=c["deg"], seed=c["sseed"], opacity=c["opacity"], with_planes=c["geo"], anisotropy=aniso, scale_sigma=sig, cluster=cl)
        mul = c["smul"] * (3.0 if sig == 0.0 else 6.0)
        inp["scales"] = (inp["scales"] * mul).astype(np.float32)
        if c["geo"]:
            inp["all_map"] = syn.plane_all_map(inp["means3D"], inp["scales"], inp["rotations"], inp["_cam"])
        c["knobs"] = "anisotropy %s cluster %.1f sigma %.0f" % (aniso, cl, sig)
    else:
        inp = scene(P=c["P"], W=c["W"], H=c["H"], deg=c["deg"], seed=c["sseed"], opacity=c["opacity"], planes=c["geo"], scale_mul=c["smul"] * (3.0 if c["big"] else 1.0))
    if c["geo"]:
        inp = add_sources(inp, n_src=c["n_src"], L=c["Lb"])
    return inp


def _pinned_rng(seed, index, big):
    """The generator as it stands right before case `index` is drawn, from tests/golden/fuzz_pins.json (tests/golden/make_fuzz_pins.py) when the case is pinned there."""
    import json, os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_pins.json")
    if os.path.exists(path):
        for e in json.load(open(path)):
            if e["kind"] == "parity" and (e["seed"], e["index"], e["mode"]) == (seed, index, big):
                rng = np.random.default_rng(0)
                rng.bit_generator.state = {"bit_generator": e["bit_generator"], "state": {"state": int(e["state"]["state"]), "inc": int(e["state"]["inc"])},
                                           "has_uint32": e["has_uint32"], "uinteger": e["uinteger"]}
                return rng
    return None


def parity_case(seed, index, big, replay=False):
    """Case `index` of `tools/fuzz_parity.py N seed - big`: (parameters, input dict, upstream gradients).  replay: draw cases 0 .. index - 1 even when the
    generator state of the case is pinned (what the pin is checked against)."""
    rng = None if replay else _pinned_rng(seed, index, big)
    if rng is None:
        rng = np.random.default_rng(seed)
        for _ in range(index):
            draw_parity_grads(rng, draw_parity(rng, big))
    c = draw_parity(rng, big)
    inp = build_parity(c)
    return c, inp, draw_parity_grads(rng, c)


ALL_GRADS = {"dL_dmeans3D": "means3D", "dL_dmeans2D": "means2D", "dL_dopacity": "opacities", "dL_dsh": "shs", "dL_dscales": "scales", "dL_drotations": "rotations"}


def oracle_builds(inp, g, cull, builds=("plain", "fma", "f64", "acc32")):
    """{build: (forward dict, backward dict)} of the oracle's builds on one case."""
    out = {}
    for b in builds:
        with oracle.variant(b):
            r = oracle.forward(inp, cull=cull)
            out[b] = (r, oracle.backward(inp, r, g["color"], g.get("normal_map"), g.get("median_depth"), g.get("warped_image")))
    return out


def arbiter_pairs(hip_grads, ob, geo):
    """Per gradient: (HIP, oracle fp32, its fma twin, the oracle with float sums) as relative L2 distances from the float64 build."""
    allk = dict(ALL_GRADS)
    if geo:
        allk["dL_dall_map"] = "all_map"
    b64 = ob["f64"][1]
    pairs = {}
    for k, v in allk.items():
        ref = np.asarray(ob["plain"][1][k])
        if np.abs(ref).sum() == 0:
            continue
        f64 = np.asarray(b64[k]).reshape(ref.shape)
        pairs[v] = tuple([rel_l2(np.asarray(hip_grads[v]).reshape(ref.shape), f64)] + [rel_l2(np.asarray(ob[b][1][k]).reshape(ref.shape), f64) for b in ("plain", "fma", "acc32")])
    return pairs


def arbiter_ratio(pairs, floor=1e-3):
    """max over the gradients of |HIP - f64| / max(floor, the farthest fp32 build of the oracle): <= 2 is the arbiter's bar (tests/test_gpu_anisotropic.py: F64_K)."""
    return max(p[0] / max(floor, max(p[1:])) for p in pairs.values())


# ---- tools/fuzz_fused.py ---------------------------------------------------------------------------------------------------------------------------
def draw_fused(rng):
    return {"P": int(rng.choice([500, 2500, 6000])), "W": int(rng.integers(96, 520)), "H": int(rng.integers(80, 340)), "seed": int(rng.integers(0, 10**6)), "learnt": bool(rng.integers(0, 2))}


def fused_case(seed, index):
    rng = np.random.default_rng(seed)
    for _ in range(index):
        draw_fused(rng)
    return draw_fused(rng)


def fused_eval(c, mask=None, builds=("plain", "fma", "f64", "acc32")):
    """(forward outputs and parameter gradients of the HIP path through renderer.render with the fused plane glue, {build: (oracle outputs, gradients)} of the oracle
    chain evaluated at the plane map the KERNELS built -- tests/test_gpu_fused_planes._oracle_chain(planes=...): the float64 arbiter must look at the same inputs)"""
    from tests.test_gpu_fused_planes import _oracle_chain, _run, _scene
    dev, g, cams, scene, pipe, args, bg = _scene(P=c["P"], W=c["W"], H=c["H"], seed=c["seed"])
    planes = {}
    o_fus, g_fus = _run(True, c["learnt"], g, dev, cams, scene, pipe, args, bg, planes_out=planes, mask=mask)
    ob = {}
    for b in builds:
        with oracle.variant(b):
            ob[b] = _oracle_chain(c["learnt"], g, dev, cams, scene, bg, planes=planes, mask=mask)
    hip = {"median_depth": o_fus["median_intersected_depth"].cpu().numpy(), "warped_image": o_fus["warped_image"].cpu().numpy(), "color": o_fus["render"].cpu().numpy(),
           "normal_map": o_fus["rendered_normal"].cpu().numpy()}
    return hip, g_fus, ob


def flipped_pixels(out, out64, H, W):
    """pixels at which a forward output differs from the float64 build's by more than rounding: a decision on a rounded float fell the other way"""
    m = np.zeros(H * W, bool)
    for k in ("median_depth", "warped_image", "color", "normal_map"):
        a, b = np.asarray(out[k]).reshape(-1, H * W), np.asarray(out64[k]).reshape(-1, H * W)
        m |= np.abs(a - b).max(0) > 1e-3 * max(1e-6, float(np.abs(b).max()))
    return m


def fused_names(c):
    return ["_xyz", "_rotation", "_scaling", "_opacity", "_features_dc"] + (["_normal", "_offset"] if c["learnt"] else [])


def fused_ratio(c, g_fus, ob, floor=1e-3):
    """(max over the parameter gradients of |HIP - f64| / max(floor, farthest fp32 oracle build), the per-gradient distances)"""
    g64 = ob["f64"][1]
    e = {n: (rel_l2(g_fus[n], g64[n]),) + tuple(rel_l2(ob[b][1][n], g64[n]) for b in ob if b != "f64") for n in fused_names(c) if g64[n] is not None and np.abs(g64[n]).sum() > 0}
    return max(p[0] / max(floor, max(p[1:])) for p in e.values()), e


def fused_verdict(c):
    """The arbiter's verdict on one fused-glue case: dict(ratio, flips_hip, flips_oracle, masked_ratio or None).  Flipped pixels (decisions that fall differently from
    float64's) are counted; when the kernels and the fp32 oracle flipped DIFFERENT pixels the gradients are compared again with those pixels' upstream gradients masked."""
    H, W = c["H"], c["W"]
    hip, g_fus, ob = fused_eval(c)
    fh, fo = flipped_pixels(hip, ob["f64"][0], H, W), flipped_pixels(ob["plain"][0], ob["f64"][0], H, W)
    r0, e0 = fused_ratio(c, g_fus, ob)
    out = {"ratio": r0, "detail": e0, "flips_hip": [(int(p % W), int(p // W)) for p in np.flatnonzero(fh)], "flips_oracle": [(int(p % W), int(p // W)) for p in np.flatnonzero(fo)],
           "masked_ratio": None, "color_l1": float(np.abs(hip["color"] - np.asarray(ob["plain"][0]["color"]).reshape(hip["color"].shape)).mean()),
           "normal_l1": float(np.abs(hip["normal_map"] - np.asarray(ob["plain"][0]["normal_map"]).reshape(hip["normal_map"].shape)).mean())}
    if (fh ^ fo).any():
        hip2, g2, ob2 = fused_eval(c, mask=(fh | fo).reshape(H, W))
        out["masked_ratio"], out["masked_detail"] = fused_ratio(c, g2, ob2)
    return out
