"""SURVEY 8(f) row 4 against the REFERENCE ITSELF (round 5): tests/golden/densify.npz holds three runs of the reference's own
`GaussianModel.densify_and_prune` (scene/gaussian_model.py:580-597, with `_prune_optimizer`, `cat_tensors_to_optimizer`, `densify_and_clone`,
`densify_and_split` underneath) on torch-CPU.  CPU side of the check (the HIP data movement is checked against the same fixture in
tests/test_gpu_densify.py):

  1. the GPU tests' restatement of the two optimiser routines (`tests/test_gpu_densify.py:_ref_prune / _ref_cat`), replayed over the recorded
     calls, ends in the reference's state bit for bit -- so "HIP == restatement" there means "HIP == reference";
  2. this repository's policy (`ibgs_amd.densify.select_* / *_rows / densify_and_prune`) makes the reference's decisions: same masks, same new
     rows, same final parameters, Adam moments and statistics, fed the reference's recorded normal draws."""
import numpy as np
import pytest
import torch

from ibgs_amd import densify
from tests import golden_densify as gd
from tests.test_gpu_densify import _ref_cat, _ref_prune

F = gd.load()


def cpu_surgery(optimizer, keep_mask=None, extension=None, extra=None):
    """`prune_and_extend_optimizer`'s contract over the restated reference routines: append first, then drop (the reference's order)."""
    n_app = 0
    if extension is not None:
        n_app = int(next(iter(extension.values())).shape[0])
        _ref_cat(optimizer, extension)
    if keep_mask is not None:
        keep = torch.cat((keep_mask, torch.ones(n_app, dtype=torch.bool)))
        _ref_prune(optimizer, keep)
    new_extra = []
    for t in (extra or []):
        t = torch.cat((t, torch.zeros((n_app,) + tuple(t.shape[1:]), dtype=t.dtype))) if n_app else t
        new_extra.append(t[keep] if keep_mask is not None else t)
    return {g["name"]: g["params"][0] for g in optimizer.param_groups}, new_extra


@pytest.mark.parametrize("tag", gd.RUNS)
def test_restated_optimizer_routines_reproduce_the_reference_state(tag):
    opt, stats = gd.build(F, tag, torch.optim.Adam, "cpu")
    groups = [str(g) for g in F["groups"]]
    for i, kind in enumerate(str(c) for c in F[tag + "calls"]):
        if kind == "cat":
            ext = {n: torch.as_tensor(F["%scall%d_new_%s" % (tag, i, n)]) for n in groups}
            _ref_cat(opt, ext)
            n = opt.param_groups[0]["params"][0].shape[0]
            stats = {k: torch.zeros((n, 1) if k in densify.STAT_NAMES[:4] else (n,)) for k in densify.STAT_NAMES}          # densification_postfix :463-468
        else:
            keep = ~torch.as_tensor(F["%scall%d_mask" % (tag, i)])
            _ref_prune(opt, keep)
            stats = {k: v[keep] for k, v in stats.items()}
    gd.check_after(F, tag, opt, stats)


@pytest.mark.parametrize("tag", gd.RUNS)
def test_policy_makes_the_reference_decisions(tag):
    opt, stats = gd.build(F, tag, torch.optim.Adam, "cpu")
    log = []
    a = F["densify_args"]
    p, new_stats = densify.densify_and_prune(opt, stats, float(a[0]), float(a[1]), float(a[2]), float(a[3]), float(a[4]), cfg=gd.config(F, tag),
                                             sampler=gd.replay_sampler(F, tag, "cpu"), surgery=gd.logging_surgery(cpu_surgery, log))
    gd.check_decisions(F, tag, log)
    gd.check_after(F, tag, opt, new_stats, xyz_tol=1e-6)
    assert set(p) == set(str(g) for g in F["groups"]) and all(p[g["name"]] is g["params"][0] for g in opt.param_groups)


def test_fixture_exercises_every_branch():
    n = int(F["N"])
    sizes = {t: [F["%scall%d_new_xyz" % (t, i)].shape[0] for i, k in enumerate(F[t + "calls"]) if str(k) == "cat"] for t in gd.RUNS}
    assert sizes["default_"][0] > 10 and sizes["default_"][1] > 100                    # uncapped clone and split (gradient + abs-gradient selections)
    assert 0 < sizes["capped_"][1] < 20 and sizes["capped_"][0] == sizes["default_"][0]   # split capped by the quantile
    assert 0 < sizes["capped_clone_"][0] <= 10 and sizes["capped_clone_"][1] == 0       # clone capped; the split then appends NOTHING and prunes nothing
    assert all(F[t + "after_param_xyz"].shape[0] != n for t in gd.RUNS)
    with np.errstate(invalid="ignore"):
        assert np.isnan(F["default_before_stat_xyz_gradient_accum"] / F["default_before_stat_denom"]).any()          # never-seen points: the NaN -> 0 path (:583-584)
