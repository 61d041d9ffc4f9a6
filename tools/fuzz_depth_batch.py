"""Random sweep of the batched depth-only pass (not part of the suite): every view of a batch bit-identical to its single pass, and close to the oracle's
depth-only pass of that camera, at random frame sizes, view counts, buffer lengths, both normal modes.  The oracle comparison is on robust statistics: a
ray that grazes its plane gives a depth of 10^4 scene units whose last digits decide a mean (tests/test_gpu_depth_batch.py's fixed frames hold none).
python tools/fuzz_depth_batch.py [n_cases] [seed]"""
import math, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from ibgs_amd import renderer, synthetic as syn
from tests.test_gpu_depth_batch import _setup

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(n_cases):
    W, H = int(rng.integers(40, 700)), int(rng.integers(40, 420))
    n = int(rng.integers(1, 9)); learnt = bool(rng.integers(0, 2)); L = int(rng.choice([1, 2, 3, 4, 5, 8]))
    dev, pc, cams, scene, pipe, args, bg = _setup(3000, W, H, max(n, 3), seed=W + n)
    views = cams[:n]
    if n >= 3:
        views[1].FoVx *= 0.8; views[1].FoVy *= 0.8
    with torch.no_grad():
        singles = torch.stack([renderer.render_depth(c, pc, scene, pipe, args, bg, learnt, 3, L) for c in views])
        batch = renderer.render_depth_batch(views, pc, scene, pipe, args, bg, learnt, 3, L)
    same = bool(torch.equal(batch, singles))
    g = pc._gnp
    worst_frac, worst_mean = 0.0, 0.0
    for v, cam in enumerate(views):
        camd = {"viewmatrix": cam.world_view_transform.cpu().numpy(), "campos": cam.camera_center.cpu().numpy()}
        am = syn.plane_all_map(g["means3D"], g["scales"], g["rotations"], camd, normal=g["normal"] if learnt else None, offset=g["offset"] if learnt else None)
        inp = {"means3D": g["means3D"], "shs": g["shs"], "opacities": g["opacities"], "scales": g["scales"], "rotations": g["rotations"],
               "all_map": am, "W": W, "H": H, "tanfovx": math.tan(cam.FoVx * 0.5), "tanfovy": math.tan(cam.FoVy * 0.5),
               "viewmatrix": camd["viewmatrix"], "projmatrix": cam.full_proj_transform.cpu().numpy(), "campos": camd["campos"],
               "bg": np.zeros(3, np.float32), "sh_degree": 1, "render_depth_only": True, "buffer_length": L}
        ref = oracle.forward(inp)["median_depth"]
        a = batch[v].cpu().numpy()
        d = np.abs(a - ref)
        tame = np.abs(ref) < 10.0 * max(float(np.median(np.abs(ref[ref != 0]))) if (ref != 0).any() else 1.0, 1e-6)
        off = d > 1e-3 * (1 + np.abs(ref))          # pixels where a decision on a rounded float (T > 0.5, depth > 0) fell the other way: counted, not averaged --
        worst_frac = max(worst_frac, float(off.mean()))          # ONE of them on a grazing plane is 10^2 scene units and would be the whole mean
        worst_mean = max(worst_mean, float(d[tame & ~off].mean() / (np.abs(ref[tame]).mean() + 1e-9)))
    ok = same and worst_frac < 2e-3 and worst_mean < 1e-4
    bad += not ok
    print("%s case %2d: %dx%d views %d learnt %d L %d | batch == singles %s, pixels off by > 1e-3: %.2e, mean rel (tame pixels) %.2e"
          % ("ok  " if ok else "FAIL", case, W, H, n, learnt, L, same, worst_frac, worst_mean), flush=True)
print("failures:", bad)
