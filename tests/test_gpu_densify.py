"""the fused "keep the masked rows, then append" pass (ibgs_compact_plan / ibgs_compact_apply) against the
reference's formulation -- boolean indexing + torch.cat per tensor and per optimiser state (scene/gaussian_model.py:377-444).
Pure data movement, so every comparison is bit for bit."""
import numpy as np
import pytest
import torch

from ibgs_amd import densify
from ibgs_amd.optim import FusedAdam

pytestmark = pytest.mark.gpu
SHAPES = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (15, 3), "opacity": (1,), "scaling": (3,), "rotation": (4,), "normal": (3,), "offset": (1,)}
LRS = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 5e-2, "scaling": 5e-3, "rotation": 1e-3, "normal": 1e-3, "offset": 1e-3}


def _groups(N, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return [{"params": [torch.nn.Parameter(torch.randn((N,) + s, device="cuda", generator=g))], "lr": LRS[n], "name": n} for n, s in SHAPES.items()]


# ---- the reference's two routines, restated over a plain optimizer (gaussian_model.py:377-395, 423-444) ----------------------
def _ref_prune(opt, mask):
    out = {}
    for group in opt.param_groups:
        st = opt.state.get(group["params"][0], None)
        if st is not None:
            st["exp_avg"] = st["exp_avg"][mask]; st["exp_avg_sq"] = st["exp_avg_sq"][mask]
            del opt.state[group["params"][0]]
            group["params"][0] = torch.nn.Parameter(group["params"][0][mask].requires_grad_(True))
            opt.state[group["params"][0]] = st
        else:
            group["params"][0] = torch.nn.Parameter(group["params"][0][mask].requires_grad_(True))
        out[group["name"]] = group["params"][0]
    return out


def _ref_cat(opt, ext):
    out = {}
    for group in opt.param_groups:
        e = ext[group["name"]]
        st = opt.state.get(group["params"][0], None)
        if st is not None:
            st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(e)), dim=0)
            st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(e)), dim=0)
            del opt.state[group["params"][0]]
            group["params"][0] = torch.nn.Parameter(torch.cat((group["params"][0], e), dim=0).requires_grad_(True))
            opt.state[group["params"][0]] = st
        else:
            group["params"][0] = torch.nn.Parameter(torch.cat((group["params"][0], e), dim=0).requires_grad_(True))
        out[group["name"]] = group["params"][0]
    return out


@pytest.mark.parametrize("N,n_app,keep_frac", [(5003, 700, 0.8), (4096, 0, 0.5), (777, 300, 1.0), (1000, 64, 0.0), (300001, 50000, 0.93)])
def test_compact_append_is_boolean_indexing_plus_cat(N, n_app, keep_frac):
    g = torch.Generator(device="cuda").manual_seed(N)
    ts = [torch.randn((N,) + s, device="cuda", generator=g) for s in SHAPES.values()] + [torch.randint(0, 1000, (N,), device="cuda", dtype=torch.int32, generator=g)]
    mask = torch.rand(N, device="cuda", generator=g) < keep_frac
    apps = [torch.randn((n_app,) + s, device="cuda", generator=g) for s in SHAPES.values()] + [None]
    got = densify.compact_append(ts, mask, apps if n_app else None)
    for t, a, o in zip(ts, apps, got):
        kept = t[mask]
        want = kept if not n_app else torch.cat((kept, a if a is not None else torch.zeros((n_app,) + t.shape[1:], dtype=t.dtype, device="cuda")), 0)
        assert o.shape == want.shape and o.dtype == t.dtype and torch.equal(o, want)
    # no mask = append only
    got2 = densify.compact_append(ts[:3], None, apps[:3] if n_app else None)
    for t, a, o in zip(ts[:3], apps[:3], got2):
        assert torch.equal(o, torch.cat((t, a), 0) if n_app else t)


def test_optimizer_surgery_equals_the_reference_routines_through_training_steps():
    """clone (append) -> split (append + prune the parents) -> prune, as densify_and_prune does (gaussian_model.py:580-597),
    with optimiser steps in between; the reference routines on torch.optim.Adam vs the fused pass on FusedAdam."""
    N = 6000
    a, b = _groups(N, 3), _groups(N, 3)
    ref = torch.optim.Adam(a, lr=0.0, eps=1e-15)
    fus = FusedAdam(b, lr=0.0, eps=1e-15)
    gen = torch.Generator(device="cuda").manual_seed(5)

    def step():
        for ga, gb in zip(a, b):
            gr = torch.randn(ga["params"][0].shape, device="cuda", generator=gen)
            ga["params"][0].grad = gr; gb["params"][0].grad = gr.clone()
        ref.step(); fus.step()

    def ext(n):
        return {nm: torch.randn((n,) + s, device="cuda", generator=gen) for nm, s in SHAPES.items()}

    step(); step()
    stats_a = torch.rand(N, 1, device="cuda", generator=gen); stats_b = stats_a.clone()
    # 1. clone: append only
    e = ext(500)
    ra = _ref_cat(ref, e)
    rb, _ = densify.prune_and_extend_optimizer(fus, None, {k: v.clone() for k, v in e.items()})
    assert set(ra) == set(rb) == set(SHAPES)
    step()
    # 2. split: append 2 children per selected parent, then prune the parents (prune_filter = cat(selected, zeros))
    n = ra["xyz"].shape[0]
    sel = torch.rand(n, device="cuda", generator=gen) < 0.1
    e = ext(2 * int(sel.sum()))
    _ref_cat(ref, e)
    ra = _ref_prune(ref, ~torch.cat((sel, torch.zeros(e["xyz"].shape[0], device="cuda", dtype=torch.bool))))
    rb, _ = densify.prune_and_extend_optimizer(fus, ~sel, {k: v.clone() for k, v in e.items()})
    step()
    # 3. prune with per-point statistics riding along (prune_points masks them too, :414-420)
    n = ra["xyz"].shape[0]
    stats_a = torch.rand(n, 1, device="cuda", generator=gen); stats_b = stats_a.clone()
    radii_a = torch.randint(0, 50, (n,), device="cuda", dtype=torch.int32, generator=gen); radii_b = radii_a.clone()
    keep = torch.rand(n, device="cuda", generator=gen) > 0.2
    ra = _ref_prune(ref, keep)
    stats_a, radii_a = stats_a[keep], radii_a[keep]
    rb, (stats_b, radii_b) = densify.prune_and_extend_optimizer(fus, keep, None, extra=[stats_b, radii_b])
    assert torch.equal(stats_a, stats_b) and torch.equal(radii_a, radii_b)
    step(); step()
    for ga, gb in zip(a, b):
        pa, pb = ga["params"][0], gb["params"][0]
        assert pa.shape == pb.shape and pb.requires_grad and pb.is_leaf
        assert rb[ga["name"]] is pb
        # the data movement is exact; what differs is FusedAdam vs torch.optim.Adam rounding over the 6 steps (test_gpu_adam.py)
        assert torch.allclose(pa, pb, rtol=2e-5, atol=2e-5), ga["name"]
        sa, sb = ref.state[pa], fus.state[pb]
        assert float(sa["step"]) == float(sb["step"]) == 6.0
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=1e-6 * float(sa["exp_avg"].abs().max()))
        # rows appended in step 2 started from zero moments on both sides
        assert sb["exp_avg"].shape == pb.shape and sb["exp_avg_sq"].shape == pb.shape


def test_surgery_alone_is_bit_exact_on_identical_states():
    """Same optimiser type on both sides (FusedAdam), so nothing but the data movement differs: parameters and both moments
    must be equal bit for bit after prune + extend."""
    N = 5000
    a, b = _groups(N, 11), _groups(N, 11)
    oa, ob = FusedAdam(a, lr=0.0, eps=1e-15), FusedAdam(b, lr=0.0, eps=1e-15)
    gen = torch.Generator(device="cuda").manual_seed(2)
    for _ in range(3):
        for ga, gb in zip(a, b):
            gr = torch.randn(ga["params"][0].shape, device="cuda", generator=gen)
            ga["params"][0].grad = gr; gb["params"][0].grad = gr.clone()
        oa.step(); ob.step()
    keep = torch.rand(N, device="cuda", generator=gen) > 0.3
    e = {nm: torch.randn((400,) + s, device="cuda", generator=gen) for nm, s in SHAPES.items()}
    _ref_prune(oa, keep); _ref_cat(oa, e)
    densify.prune_and_extend_optimizer(ob, keep, e)
    for ga, gb in zip(a, b):
        pa, pb = ga["params"][0], gb["params"][0]
        assert torch.equal(pa, pb)
        for k in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(oa.state[pa][k], ob.state[pb][k]), (ga["name"], k)
        assert float(oa.state[pa]["step"]) == float(ob.state[pb]["step"])


def test_edge_cases_and_errors():
    t = torch.arange(12, device="cuda", dtype=torch.float32).reshape(4, 3)
    none = densify.compact_append([t], torch.zeros(4, dtype=torch.bool, device="cuda"))
    assert none[0].shape == (0, 3)
    all_ = densify.compact_append([t], torch.ones(4, dtype=torch.bool, device="cuda"), [torch.full((2, 3), 7.0, device="cuda")])
    assert torch.equal(all_[0][:4], t) and torch.equal(all_[0][4:], torch.full((2, 3), 7.0, device="cuda"))
    empty = densify.compact_append([t[:0]], None, [torch.ones(2, 3, device="cuda")])
    assert torch.equal(empty[0], torch.ones(2, 3, device="cuda"))
    with pytest.raises(ValueError):
        densify.compact_append([t], torch.ones(3, dtype=torch.bool, device="cuda"))
    with pytest.raises(ValueError):
        densify.compact_append([t], None, [torch.ones(2, 4, device="cuda")])
    with pytest.raises(RuntimeError):
        densify.compact_append([t.cpu()])


# ---- against the reference's own densify_and_prune (tests/golden/densify.npz; round 5) ----------------------------------------------------------
from tests import golden_densify as gd  # noqa: E402

GOLD = gd.load()


@pytest.mark.parametrize("tag", gd.RUNS)
def test_fused_pass_replays_the_reference_surgery_bit_for_bit(tag):
    """Every `densification_postfix` / `prune_points` call the reference made (scene/gaussian_model.py:397-469), replayed through the fused HIP pass
    on FusedAdam: parameters, exp_avg, exp_avg_sq, step counts and the per-point statistics end in the reference's state, bit for bit."""
    opt, stats = gd.build(GOLD, tag, FusedAdam, "cuda")
    groups = [str(g) for g in GOLD["groups"]]
    for i, kind in enumerate(str(c) for c in GOLD[tag + "calls"]):
        if kind == "cat":
            ext = {n: torch.as_tensor(GOLD["%scall%d_new_%s" % (tag, i, n)], device="cuda") for n in groups}
            p, _ = densify.prune_and_extend_optimizer(opt, None, ext)
            n = p["xyz"].shape[0]
            stats = {k: torch.zeros((n, 1) if k in densify.STAT_NAMES[:4] else (n,), device="cuda") for k in densify.STAT_NAMES}
        else:
            keep = ~torch.as_tensor(GOLD["%scall%d_mask" % (tag, i)], device="cuda")
            p, extra = densify.prune_and_extend_optimizer(opt, keep, None, extra=[stats[k] for k in densify.STAT_NAMES])
            stats = dict(zip(densify.STAT_NAMES, extra))
    gd.check_after(GOLD, tag, opt, stats)


@pytest.mark.parametrize("tag", gd.RUNS)
def test_densify_and_prune_on_the_gpu_is_the_reference_s(tag):
    """The whole `densify.densify_and_prune` on the MI355X (selection in torch-HIP, three fused data-movement passes) fed the reference's normal
    draws: the reference's decisions (masks, new rows) and the reference's final state."""
    opt, stats = gd.build(GOLD, tag, FusedAdam, "cuda")
    log = []
    a = GOLD["densify_args"]
    p, new_stats = densify.densify_and_prune(opt, stats, float(a[0]), float(a[1]), float(a[2]), float(a[3]), float(a[4]), cfg=gd.config(GOLD, tag),
                                             sampler=gd.replay_sampler(GOLD, tag, "cuda"), surgery=gd.logging_surgery(densify.prune_and_extend_optimizer, log))
    gd.check_decisions(GOLD, tag, log)
    gd.check_after(GOLD, tag, opt, new_stats, xyz_tol=2e-6)
    for g in opt.param_groups:          # and the optimiser still steps (FusedAdam on the new Parameters)
        g["params"][0].grad = torch.ones_like(g["params"][0])
    opt.step()


def test_densification_statistics_in_one_launch():
    """`densify.add_densification_stats` against the reference's five boolean-indexed updates (train.py:400-405, scene/gaussian_model.py:600-604),
    restated with torch on the same tensors: bit for bit (the norms are formed without contraction, as torch.norm over two values does)."""
    P = 20011
    gen = torch.Generator(device="cuda").manual_seed(4)
    radii = torch.randint(-2, 40, (P,), device="cuda", dtype=torch.int32, generator=gen).clamp_min(0)
    vp = torch.zeros(P, 3, device="cuda", requires_grad=True); vpa = torch.zeros(P, 3, device="cuda", requires_grad=True)
    vp.grad = torch.randn(P, 3, device="cuda", generator=gen); vpa.grad = torch.randn(P, 3, device="cuda", generator=gen).abs()
    a = {k: torch.rand((P, 1) if i < 4 else (P,), device="cuda", generator=gen) * (30.0 if k == "max_radii2D" else 1.0) for i, k in enumerate(densify.STAT_NAMES)}
    b = {k: v.clone() for k, v in a.items()}
    for _ in range(2):
        densify.add_densification_stats(a, vp, vpa, radii)
        mask = radii > 0          # visibility_filter
        b["max_radii2D"][mask] = torch.max(b["max_radii2D"][mask], radii[mask].float())
        b["xyz_gradient_accum"][mask] += torch.norm(vp.grad[mask, :2], dim=-1, keepdim=True)
        b["xyz_gradient_accum_abs"][mask] += torch.norm(vpa.grad[mask, :2], dim=-1, keepdim=True)
        b["denom"][mask] += 1
        b["denom_abs"][mask] += 1
    assert int((radii > 0).sum()) not in (0, P)
    for k in densify.STAT_NAMES:
        assert torch.equal(a[k], b[k]), k
    # without the |.| statistic (renderer.TRACK_ABS_GRAD = False: the abs sink carries no gradient): its two tensors are left alone
    c = {k: v.clone() for k, v in a.items()}
    densify.add_densification_stats(c, vp, None, radii)
    assert torch.equal(c["xyz_gradient_accum_abs"], a["xyz_gradient_accum_abs"]) and torch.equal(c["denom_abs"], a["denom_abs"]) and not torch.equal(c["denom"], a["denom"])
