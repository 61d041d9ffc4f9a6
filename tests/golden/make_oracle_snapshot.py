#!/usr/bin/env python3
"""Freezes the oracle's answer for BASELINE config C1 (10 k random-init Gaussians, 400x400, SH 3, forward + backward)
into tests/golden/oracle_c1.npz: a 96x96 crop of the image and of the per-pixel counters, head rows of every
gradient and float64 checksums of the full arrays.  The reference ships no golden vectors for this path,
so this snapshot pins the ORACLE against silent edits and gives the HIP path a committed target that does not move
with the oracle.  Usage (repo root): python tests/golden/make_oracle_snapshot.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from ibgs_amd import synthetic as syn  # noqa: E402

CROP = (slice(152, 248), slice(152, 248))
HEAD = 256


def build():
    c = syn.CONFIGS["C1"]
    inp = syn.make_scene(c["P"], c["W"], c["H"], sh_degree=c["sh_degree"], seed=c["seed"], opacity="trained")
    g = np.random.default_rng(77).standard_normal((3, c["H"], c["W"])).astype(np.float32)
    return inp, g


def snapshot(inp, g):
    f = oracle.forward(inp, cull=True)      # exact tile culling: shorter lists, identical public outputs
    b = oracle.backward(inp, f, g)
    out = {"num_rendered": np.int64(f["num_rendered"]), "num_rendered_aabb": np.int64(oracle.forward(inp)["num_rendered"]), "color_crop": f["color"][(slice(None),) + CROP],
           "n_contrib_crop": f["n_contrib"].reshape(inp["H"], inp["W"])[CROP], "final_T_crop": f["final_T"].reshape(inp["H"], inp["W"])[CROP],
           "radii_head": f["radii"][:HEAD * 8], "color_sum": np.float64(f["color"].astype(np.float64).sum()),
           "color_abs_sum": np.float64(np.abs(f["color"]).astype(np.float64).sum())}
    for k in ("dL_dmeans3D", "dL_dsh", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dmeans2D"):
        a = np.asarray(b[k], np.float32).reshape(inp["means3D"].shape[0], -1)
        out[k + "_head"] = a[:HEAD].copy()
        out[k + "_abs_sum"] = np.float64(np.abs(a).astype(np.float64).sum())
        out[k + "_sum"] = np.float64(a.astype(np.float64).sum())
    return out


if __name__ == "__main__":
    inp, g = build()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "oracle_c1.npz"), **snapshot(inp, g))
    print("wrote tests/golden/oracle_c1.npz")
