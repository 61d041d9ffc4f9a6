"""The four cases the round-5 sweeps left outside the float64 arbiter's bar, pinned where the driver sees them ("What's weak" 1c):

  tools/fuzz_parity.py 120 9103 - trained, cases 9 and 117      needle / giant-plane scenes: dL/dscales, dL/dmeans3D, dL/drotations 2.6-5 x as far from the float64
                                                                  build as the fp32 oracle's builds
  tools/fuzz_fused.py 100 9105, cases 1 and 38                   fused plane glue: 4.4 x / 5.5 x

What they were made of and what closed them (profiles/r06_ref_arith_ab.txt):
  * 9 and 117: the chain dL/dconic -> dL/dcov2D of backward.cu:405-420 on the sums of near-singular conics -- ANY fp32 evaluation of it scatters 0.8 .. 3.9 x around the
    fp32 oracle's own distance, by summation order alone.  The kernels now sum those Gaussians in l-space (csrc/render_bwd.hip: l_moments), where nothing cancels:
    0.35 x / 0.80 x, the same number in every run with float atomics.  IBGS_FLAG_REF_ARITH (the reference's own association, SURVEY Q1 as a switch) is run beside it.
  * fused 1: ONE pixel sees a plane edge-on (n . ray ~ 1e-5): the plane map the kernels build differs from the torch glue's by an ulp of the normal, which is per cents of
    that pixel's depth.  An input effect: with the float64 arbiter looking at the SAME plane map the kernels are exactly as far from it as the fp32 oracle (ratio 1.00).
  * fused 38: one pixel whose source-validity decision falls differently in each fp32 evaluation (the kernels flip pixel (65, 45) against float64, the fp32 oracle pixel
    (184, 185)): a discontinuity of the reference's function.  Flipped pixels are COUNTED; with their upstream gradients masked the gradients agree.
The arbiter (tests/test_gpu_anisotropic.py): a float evaluation is as good as its distance from the float64 build of the oracle; bar: HIP <= 2 x the fp32 oracle builds."""
import numpy as np
import pytest
import torch

import oracle
from ibgs_amd import rasterizer
from tests import fuzz_cases as fc, hipref
from tests.metrics import l1, rel_l2

pytestmark = pytest.mark.gpu

# measured ratios (profiles/r06_ref_arith_ab.txt; the r05 library on the same box: 1.3-3.3 and 1.9-2.9 over six runs with float atomics)
PARITY_PINS = {(9103, 9): 0.35, (9103, 117): 0.80}


def _hip_grads(inp, g, c):
    outs, lv, _ = hipref.run_forward(inp)
    ist = hipref.internal_state(outs, inp)
    col = outs["color"].detach().cpu().numpy()
    loss = (outs["color"] * torch.as_tensor(g["color"], device="cuda")).sum()
    if c["geo"]:
        for k in ("normal_map", "median_depth", "warped_image"):
            loss = loss + (outs[k] * torch.as_tensor(g[k], device="cuda")).sum()
    loss.backward()
    torch.cuda.synchronize()
    return col, ist, {v: lv[v].grad.cpu().numpy() for v in list(fc.ALL_GRADS.values()) + (["all_map"] if c["geo"] else [])}


@pytest.mark.parametrize("seed,index", sorted(PARITY_PINS))
def test_needle_scenes_of_the_trained_sweep(seed, index):
    c, inp, g = fc.parity_case(seed, index, "trained")
    ob = fc.oracle_builds(inp, g, c["cull"])
    old = (rasterizer.TILE_CULL, rasterizer.WAVE_SHAPE, rasterizer.DETERMINISTIC, rasterizer.REF_ARITH)
    try:
        rasterizer.TILE_CULL, rasterizer.WAVE_SHAPE = c["cull"], c["wave_shape"]
        ratios = {}
        for mode, ref_arith, det in (("l-space, deterministic", False, True), ("l-space, float atomics", False, False), ("reference association, deterministic", True, True)):
            rasterizer.REF_ARITH, rasterizer.DETERMINISTIC = ref_arith, det
            col, ist, hip = _hip_grads(inp, g, c)
            ref = ob["plain"][0]
            assert ist["R"] == ref["num_rendered"] and np.array_equal(ist["point_list"], ref["point_list"]), "lists"
            assert l1(col, ref["color"]) < 1e-5 and float((ist["n_contrib"] == ref["n_contrib"]).mean()) > 0.9995
            pairs = fc.arbiter_pairs(hip, ob, c["geo"])
            ratios[mode] = fc.arbiter_ratio(pairs)
            strict = max(p[0] / max(1e-3, max(p[1], p[2])) for p in pairs.values())          # against the oracle's two fp32 builds alone (without the float-sum build)
            print("[fuzz pin] parity %d/%d %-38s ratio %.2f (strict %.2f) | %s" % (seed, index, mode, ratios[mode], strict, ", ".join("%s %.1e|%.1e|%.1e|%.1e" % ((v,) + p) for v, p in pairs.items())))
            if not ref_arith:
                # the default path: no farther from float64 than the fp32 oracle's own builds (measured 0.35 / 0.80, the same in every run) -- bar 2, asserted at 1.1
                assert ratios[mode] <= 1.1 and strict <= 1.25, (mode, ratios[mode], strict)
                assert abs(ratios[mode] - PARITY_PINS[(seed, index)]) <= 0.15, "the measured ratio moved: %s" % ratios
            else:
                # the reference's own arithmetic: one draw from the scatter every fp32 evaluation of backward.cu:405-420 shows on these scenes (0.35 .. 4.1 over six runs)
                assert ratios[mode] <= 8.0, ratios
    finally:
        rasterizer.TILE_CULL, rasterizer.WAVE_SHAPE, rasterizer.DETERMINISTIC, rasterizer.REF_ARITH = old


def test_pinned_generator_states_replay_the_sweep():
    """tests/golden/fuzz_pins.json holds the generator state right before each pinned case: the same case as drawing the sweep up to it (checked on the cheap one)."""
    c1, i1, g1 = fc.parity_case(9103, 9, "trained")
    c2, i2, g2 = fc.parity_case(9103, 9, "trained", replay=True)
    assert c1 == c2 and np.array_equal(g1["color"], g2["color"]) and all(np.array_equal(i1[k], i2[k]) for k in ("means3D", "scales", "rotations", "opacities", "shs"))


# ---- the fused-glue sweep ---------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("index,flips_hip", [(1, 1), (38, 1)])
def test_fused_glue_sweep_cases(index, flips_hip):
    c = fc.fused_case(9105, index)
    v = fc.fused_verdict(c)
    assert v["color_l1"] < 1e-6
    print("[fuzz pin] fused 9105/%d at the kernels' plane map: ratio %.2f; pixels that decide differently from float64: HIP %s, fp32 oracle %s" % (index, v["ratio"], v["flips_hip"], v["flips_oracle"]))
    # flipped pixels are counted, not averaged: no more of them than the fp32 oracle has (+ 1), out of H x W
    assert len(v["flips_hip"]) <= flips_hip and len(v["flips_hip"]) <= len(v["flips_oracle"]) + 1, v
    if v["masked_ratio"] is not None:          # the two fp32 evaluations flipped DIFFERENT pixels: compare what is comparable -- every pixel either side decided differently carries no gradient
        print("[fuzz pin] fused 9105/%d with those pixels' upstream gradients masked: ratio %.2f | %s" % (index, v["masked_ratio"], {k: "%.1e|%.1e|%.1e|%.1e" % x for k, x in v["masked_detail"].items()}))
        assert v["masked_ratio"] <= 2.0, v
    else:
        assert v["ratio"] <= 2.0, v
