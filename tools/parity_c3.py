"""Full-size parity datapoint (not in the test suite: the oracle needs ~10 s on the box's host cores): HIP vs oracle on the
C3 bench workload -- image L1 / max / PSNR difference, per-pixel counter agreement, gradient relative L2."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from ibgs_amd import synthetic as syn
from tests import hipref
from tests.metrics import l1, psnr, rel_l2

c = syn.CONFIGS["C3"]
for opacity in ("init", "trained"):
    inp = syn.make_scene(c["P"], c["W"], c["H"], sh_degree=3, seed=c["seed"], opacity=opacity)
    t0 = time.time(); ref = oracle.forward(inp, cull=True); t1 = time.time()
    g = np.random.default_rng(1).standard_normal((3, c["H"], c["W"])).astype(np.float32)
    rb = oracle.backward(inp, ref, g); t2 = time.time()
    outs, lv, _ = hipref.run_forward(inp)
    ist = hipref.internal_state(outs, inp)
    col = outs["color"].detach().cpu().numpy()
    (outs["color"] * torch.as_tensor(g, device="cuda")).sum().backward()
    tgt = np.random.default_rng(2).random(col.shape).astype(np.float32)
    print("C3 opacity=%s: oracle fwd %.1f s bwd %.1f s | R %d == %d: %s | lists equal: %s" % (opacity, t1 - t0, t2 - t1, ist["R"], ref["num_rendered"],
          ist["R"] == ref["num_rendered"], np.array_equal(ist["point_list"], ref["point_list"])))
    print("   image: mean L1 %.3e  max |d| %.3e  PSNR(vs common target) HIP %.4f dB oracle %.4f dB | n_contrib equal on %.5f of pixels"
          % (l1(col, ref["color"]), np.abs(col - ref["color"]).max(), psnr(col, tgt)[0], psnr(ref["color"], tgt)[0],
             (ist["n_contrib"] == ref["n_contrib"]).mean()))
    names = {"dL_dmeans3D": "means3D", "dL_dsh": "shs", "dL_dopacity": "opacities", "dL_dscales": "scales", "dL_drotations": "rotations", "dL_dmeans2D": "means2D"}
    print("   grads rel L2:", {k: float("%.2e" % rel_l2(lv[v].grad.cpu().numpy().reshape(np.asarray(rb[k]).shape), rb[k])) for k, v in names.items()})
