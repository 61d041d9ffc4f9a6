"""Reader of tests/golden/densify.npz (made by tests/golden/make_densify_fixture.py from the reference's own `GaussianModel.densify_and_prune`,
scene/gaussian_model.py:377-597) and the replay helpers shared by the CPU test (torch.optim.Adam + the reference's two optimiser routines as the
GPU tests restate them) and the GPU test (FusedAdam + `densify.prune_and_extend_optimizer`)."""
import os

import numpy as np
import torch

from ibgs_amd import densify
from tests.golden_glue import Fixture

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "densify.npz")
RUNS = ("default_", "capped_", "capped_clone_")


def load():
    return Fixture(PATH)


def build(F, tag, opt_cls, device):
    """The optimiser + statistics of the fixture's `before` state: eight named groups, Adam moments after two steps."""
    groups = [str(g) for g in F["groups"]]
    lrs = F[tag + "lr"]
    pg = [{"params": [torch.nn.Parameter(torch.as_tensor(F[tag + "before_param_" + n], device=device))], "lr": float(lr), "name": n} for n, lr in zip(groups, lrs)]
    opt = opt_cls(pg, lr=0.0, eps=float(F[tag + "eps"]))
    for g in opt.param_groups:
        p, n = g["params"][0], g["name"]
        opt.state[p] = {"step": torch.tensor(float(F[tag + "before_step_" + n])), "exp_avg": torch.as_tensor(F[tag + "before_exp_avg_" + n], device=device),
                        "exp_avg_sq": torch.as_tensor(F[tag + "before_exp_avg_sq_" + n], device=device)}
    stats = {s: torch.as_tensor(F[tag + "before_stat_" + s], device=device) for s in densify.STAT_NAMES}
    return opt, stats


def config(F, tag):
    pd, thr, mabs, mall = F[tag + "targs"]
    return densify.DensifyConfig(percent_dense=float(pd), abs_split_radii2D_threshold=float(thr), max_abs_split_points=int(mabs), max_all_points=int(mall))


def check_after(F, tag, opt, stats, xyz_tol=0.0):
    """Parameters, both Adam moments, step counts and the six per-point statistics against the reference's state after densify_and_prune."""
    for g in opt.param_groups:
        p, n = g["params"][0], g["name"]
        want = F[tag + "after_param_" + n]
        got = p.detach().cpu().numpy()
        assert got.shape == want.shape, (n, got.shape, want.shape)
        if n in ("xyz", "scaling") and xyz_tol > 0:          # new positions go through a 3 x 3 product, the children's scales through log(exp(s) / 1.6): rounding
            assert np.allclose(got, want, rtol=xyz_tol, atol=xyz_tol), n          # of another bmm / exp / log (CPU vs HIP); copied rows are exact on either
        else:
            assert np.array_equal(got, want), n
        st = opt.state[p]
        assert float(st["step"]) == float(F[tag + "after_step_" + n])
        assert np.array_equal(st["exp_avg"].cpu().numpy(), F[tag + "after_exp_avg_" + n]), n
        assert np.array_equal(st["exp_avg_sq"].cpu().numpy(), F[tag + "after_exp_avg_sq_" + n]), n
        assert p.requires_grad and p.is_leaf
    for s in densify.STAT_NAMES:
        assert np.array_equal(stats[s].cpu().numpy(), F[tag + "after_stat_" + s]), s


def replay_sampler(F, tag, device):
    """`torch.normal` stand-in handing back the reference's recorded draws in order (the only randomness of densify_and_prune)."""
    it = iter(range(int(F[tag + "ndraws"])))

    def sampler(mean, std):
        d = torch.as_tensor(F["%sdraw%d" % (tag, next(it))], device=device)
        assert d.shape == std.shape, "the reference drew for %s rows, this policy selected %s" % (tuple(d.shape), tuple(std.shape))
        return d
    return sampler


def logging_surgery(surgery, log):
    def wrapped(optimizer, keep_mask=None, extension=None, extra=None):
        log.append((None if keep_mask is None else keep_mask.cpu().numpy().copy(),
                    None if extension is None else {k: v.detach().cpu().numpy().copy() for k, v in extension.items()}))
        return surgery(optimizer, keep_mask, extension, extra=extra)
    return wrapped


def check_decisions(F, tag, log):
    """The passes this repository's policy made against the calls the reference made: `cat` (clone), `cat` + `prune` (split; one fused pass
    here), `prune`.  A clone that selects nothing makes no call on either side."""
    calls = [str(c) for c in F[tag + "calls"]]
    groups = [str(g) for g in F["groups"]]
    i = 0
    for keep, ext in log:
        if ext is not None:
            assert calls[i] == "cat"
            for n in groups:
                want = F["%scall%d_new_%s" % (tag, i, n)]
                assert ext[n].shape == want.shape, (n, ext[n].shape, want.shape)
                assert np.allclose(ext[n], want, rtol=2e-6, atol=2e-6) if n in ("xyz", "scaling") else np.array_equal(ext[n], want), "new rows of %s (call %d)" % (n, i)
            n_app = ext["xyz"].shape[0]
            i += 1
            if keep is not None:          # the split: the reference prunes cat(selected, zeros(children)) right after the append
                assert calls[i] == "prune"
                m = F["%scall%d_mask" % (tag, i)]
                assert np.array_equal(~keep, m[:keep.size]) and not m[keep.size:].any() and m.size == keep.size + n_app
                i += 1
        else:
            assert calls[i] == "prune" and np.array_equal(~keep, F["%scall%d_mask" % (tag, i)])
            i += 1
    assert i == len(calls)
