"""Generates tests/golden/densify.npz IN THE BUILD CONTAINER by RUNNING the reference's own densification code (scene/gaussian_model.py):

    _prune_optimizer :377-395, prune_points :397-421, cat_tensors_to_optimizer :423-444, densification_postfix :446-469,
    densify_and_split :471-522, densify_and_clone :549-577, densify_and_prune :580-597, training_setup :216-246 (the eight Adam groups)

on a seeded GaussianModel after two optimiser steps (so every group has Adam moments), on torch-CPU (tests/golden/_ref_import.py: allocation device
redirected, absent third-party imports kept off the call path).  Recorded per run: the state before (parameters, exp_avg, exp_avg_sq, per-point
statistics), every surgery call the reference made in order -- `densification_postfix(new tensors)` / `prune_points(mask)` with their arguments --,
every `torch.normal` draw (the only randomness), and the state after.  Three runs: the default thresholds, and two with `max_all_points` small
enough to take the quantile-capped branches of the split and of the clone.  Data only.  Re-run:  python tests/golden/make_densify_fixture.py"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_import as ri  # noqa: E402

ri.install()
from scene.gaussian_model import GaussianModel  # noqa: E402

GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation", "normal", "offset")
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation",
        "normal": "_normal", "offset": "_offset"}
STATS = ("xyz_gradient_accum", "xyz_gradient_accum_abs", "denom", "denom_abs", "max_radii2D", "max_weight")
N, DEG, EXTENT = 320, 1, 3.0
out = {"N": N, "DEG": DEG, "extent": EXTENT, "groups": np.asarray(GROUPS, dtype="U16"), "stats": np.asarray(STATS, dtype="U32")}


def state(gm, prefix):
    for g in gm.optimizer.param_groups:
        p = g["params"][0]
        out[prefix + "param_" + g["name"]] = p.detach().numpy().copy()
        st = gm.optimizer.state.get(p)
        if st is not None:
            out[prefix + "exp_avg_" + g["name"]] = st["exp_avg"].numpy().copy(); out[prefix + "exp_avg_sq_" + g["name"]] = st["exp_avg_sq"].numpy().copy()
            out[prefix + "step_" + g["name"]] = float(st["step"])
        assert getattr(gm, ATTR[g["name"]]) is p
    for s in STATS:
        out[prefix + "stat_" + s] = getattr(gm, s).numpy().copy()


def run(tag, max_all_points):
    rng = np.random.default_rng(77)
    f32 = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))
    gm = GaussianModel(DEG)
    shapes = {"xyz": (3,), "f_dc": (1, 3), "f_rest": ((DEG + 1) ** 2 - 1, 3), "opacity": (1,), "scaling": (3,), "rotation": (4,), "normal": (3,), "offset": (1,)}
    init = {k: rng.normal(size=(N,) + s) for k, s in shapes.items()}
    init["scaling"] = rng.normal(-3.8, 1.2, (N, 3))          # exp(.) straddles percent_dense * extent = 0.03: clones (small) and splits (large)
    init["opacity"] = rng.normal(-2.0, 3.0, (N, 1))          # some below min_opacity after the sigmoid
    for k in GROUPS:
        setattr(gm, ATTR[k], torch.nn.Parameter(f32(init[k])))
    gm.spatial_lr_scale = 1.0
    targs = SimpleNamespace(percent_dense=0.01, abs_split_radii2D_threshold=20, max_abs_split_points=50_000, max_all_points=max_all_points,
                            position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_delay_mult=0.01, position_lr_max_steps=30_000,
                            feature_lr=0.0025, opacity_lr=0.025, scaling_lr=0.005, rotation_lr=0.001, normal_lr=0.001)
    gm.training_setup(targs)
    out[tag + "lr"] = np.asarray([g["lr"] for g in gm.optimizer.param_groups], np.float64)
    out[tag + "eps"] = gm.optimizer.defaults["eps"]
    for _ in range(2):
        for g in gm.optimizer.param_groups:
            g["params"][0].grad = f32(rng.normal(size=tuple(g["params"][0].shape)))
        gm.optimizer.step()
    gm.optimizer.zero_grad(set_to_none=True)
    # per-point statistics as train.py:400-410 leaves them: some points never seen (denom 0 -> NaN -> 0, :583-584)
    denom = rng.integers(0, 6, (N, 1)).astype(np.float32)
    gm.xyz_gradient_accum = f32(rng.gamma(2.0, 0.0002, (N, 1)) * denom); gm.xyz_gradient_accum_abs = f32(rng.gamma(2.0, 0.0008, (N, 1)) * denom)
    gm.denom = f32(denom); gm.denom_abs = f32(denom)
    gm.max_radii2D = f32(rng.uniform(0, 40, N)); gm.max_weight = f32(rng.uniform(0, 1, N))
    state(gm, tag + "before_")

    calls, draws = [], []
    postfix, prune, normal = gm.densification_postfix, gm.prune_points, torch.normal

    def rec_postfix(*new):
        calls.append(("cat", [t.detach().numpy().copy() for t in new]))
        return postfix(*new)

    def rec_prune(mask):
        calls.append(("prune", mask.numpy().copy()))
        return prune(mask)

    def rec_normal(*a, **k):
        s = normal(*a, **k)
        draws.append(s.numpy().copy())
        return s
    gm.densification_postfix, gm.prune_points, torch.normal = rec_postfix, rec_prune, rec_normal
    try:
        torch.manual_seed(123)
        with torch.no_grad():          # train.py:399 calls it under no_grad
            gm.densify_and_prune(0.0002, 0.0008, 0.005, EXTENT, 20)
    finally:
        torch.normal = normal
    kinds = []
    for i, (kind, payload) in enumerate(calls):
        kinds.append(kind)
        if kind == "cat":
            for name, t in zip(GROUPS, payload):          # densification_postfix's argument order IS the group order (:446-455)
                out["%scall%d_new_%s" % (tag, i, name)] = t
        else:
            out["%scall%d_mask" % (tag, i)] = payload
    out[tag + "calls"] = np.asarray(kinds, dtype="U8")
    for i, d in enumerate(draws):
        out["%sdraw%d" % (tag, i)] = d
    out[tag + "ndraws"] = len(draws)
    out[tag + "targs"] = np.asarray([targs.percent_dense, targs.abs_split_radii2D_threshold, targs.max_abs_split_points, targs.max_all_points], np.float64)
    state(gm, tag + "after_")
    n_after = gm.get_xyz.shape[0]
    print("%s %d points -> %d; calls %s, %d normal draws" % (tag, N, n_after, kinds, len(draws)))


run("default_", 5_000_000)
run("capped_", N + 40)
run("capped_clone_", N + 10)          # the clone itself is capped; the split then selects nothing: an EMPTY append + an all-False prune
out["densify_args"] = np.asarray([0.0002, 0.0008, 0.005, EXTENT, 20], np.float64)          # max_grad, abs_max_grad, min_opacity, extent, max_screen_size
ri.save_deduped(os.path.join(HERE, "densify.npz"), out)          # (the three runs share their "before" state)
print("densify.npz written: %.0f KB" % (os.path.getsize(os.path.join(HERE, "densify.npz")) / 1024))
