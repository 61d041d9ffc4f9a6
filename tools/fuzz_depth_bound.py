"""Random sweep of the depth-bound hint (not part of the suite): random frames, sizes, modes and wave shapes; between frames the scene is left alone, nudged
(a trainer's step) or changed outright (opacities dropped, Gaussians moved) -- so clean frames, repaired frames and frames after a repair all occur.  Every
frame rendered with the hint must equal the frame rendered without it bit for bit: images, radii, final_T / n_contrib, and (deterministic backward) gradients.
python tools/fuzz_depth_bound.py [n_cases] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibgs_amd import rasterizer
from tests import test_gpu_depth_bound as T
from tests.scenes import add_sources, scene

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rasterizer._camera_key = lambda viewmatrix, device, W, H, geo, stream: ("fuzz camera", W, H, bool(geo))
bad = 0
for case in range(n_cases):
    W, H = int(rng.integers(32, 900)), int(rng.integers(32, 600))
    P = int(rng.choice([300, 3000, 20000, 80000])); geo = bool(rng.integers(0, 3) == 0)
    smul = float(rng.choice([1.0, 2.0, 4.0])); deg = int(rng.integers(0, 4))
    rasterizer.WAVE_SHAPE = [None, None, "tile", "quadrant"][int(rng.integers(0, 4))]
    inp = scene(P=P, W=W, H=H, deg=deg, seed=int(rng.integers(0, 10**6)), opacity=str(rng.choice(["init", "trained"])), planes=geo, scale_mul=smul)
    if geo:
        inp = add_sources(inp, n_src=int(rng.integers(1, 5)), L=int(rng.choice([2, 4, 5])), depth=np.full((5, H, W), 4.0, np.float32)[:4])
        inp["src_depths"] = inp["src_depths"][:inp["n_src"]]
    plan = [str(rng.choice(["same", "same", "nudge", "drop", "move"])) for _ in range(6)]
    state = {"cur": inp}
    seeds = rng.integers(0, 10**6, size=6)

    def mutate(i, base):
        if i >= 2:
            cur = dict(state["cur"]); r = np.random.default_rng(int(seeds[i])); what = plan[i]
            if what == "nudge":
                cur["means3D"] = (cur["means3D"] + 1e-3 * r.standard_normal(cur["means3D"].shape)).astype(np.float32)
                cur["opacities"] = np.clip(cur["opacities"] * (1 + 0.02 * r.standard_normal(cur["opacities"].shape)), 0.001, 0.999).astype(np.float32)
            elif what == "drop":
                cur["opacities"] = np.where(r.random(cur["opacities"].shape) < 0.5, 0.02, cur["opacities"]).astype(np.float32)
            elif what == "move":
                cur["means3D"] = (cur["means3D"] + 0.2 * r.standard_normal(cur["means3D"].shape)).astype(np.float32)
            if geo and what in ("nudge", "move"):
                from ibgs_amd import synthetic as syn
                cur["all_map"] = syn.plane_all_map(cur["means3D"], cur["scales"], cur["rotations"], cur["_cam"])
            state["cur"] = cur
        return state["cur"]
    try:
        state["cur"] = inp; ref = T.frames(inp, 6, False, mutate)
        state["cur"] = inp; got = T.frames(inp, 6, True, mutate)
        for i, (a, b) in enumerate(zip(got, ref)):
            T.same_frame(a, b, "frame %d" % i)
        ok = True; msg = ""
    except AssertionError as ex:
        ok = False; msg = str(ex)[:200]
    bad += not ok
    print("%s case %2d: %dx%d P %d geo %d deg %d scale x%.0f shape %s plan %s | repairs %s, listed / R %s %s"
          % ("ok  " if ok else "FAIL", case, W, H, P, geo, deg, smul, rasterizer.WAVE_SHAPE, plan[2:], [int(f["meta"][12]) for f in got[2:]] if ok else "-",
             ["%.2f" % (f["st"]["R"] / max(f["R"], 1)) for f in got[2:]] if ok else "-", msg), flush=True)
rasterizer.WAVE_SHAPE = None
print("failures:", bad)
