"""`FusedAdam` (one HIP launch for all parameter tensors) against torch.optim.Adam with the
reference's settings (lr per group, eps = 1e-15; scene/gaussian_model.py:227-241), including the optimiser-state
surgery the reference performs when it densifies."""
import numpy as np
import pytest
import torch

from ibgs_amd.optim import FusedAdam

pytestmark = pytest.mark.gpu
SHAPES = {"xyz": (5003, 3), "f_dc": (5003, 1, 3), "f_rest": (5003, 15, 3), "opacity": (5003, 1), "scaling": (5003, 3),
          "rotation": (5003, 4), "normal": (5003, 3), "offset": (5003, 1), "odd": (1237,)}
LRS = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 5e-2, "scaling": 5e-3, "rotation": 1e-3, "normal": 1e-3, "offset": 1e-3, "odd": 1e-2}


def _groups(seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return [{"params": [torch.nn.Parameter(torch.randn(s, device="cuda", generator=g))], "lr": LRS[n], "name": n} for n, s in SHAPES.items()]


def test_matches_torch_adam_over_many_steps():
    a, b = _groups(1), _groups(1)
    ref = torch.optim.Adam(a, lr=0.0, eps=1e-15)
    fus = FusedAdam(b, lr=0.0, eps=1e-15)
    gen = torch.Generator(device="cuda").manual_seed(9)
    for it in range(30):
        for ga, gb in zip(a, b):
            gr = torch.randn(ga["params"][0].shape, device="cuda", generator=gen) * (10.0 ** ((it % 5) - 3))
            if it == 7 and ga["name"] == "opacity":
                gr = None                                   # a parameter without gradient is skipped, its step does not advance
            ga["params"][0].grad = gr; gb["params"][0].grad = None if gr is None else gr.clone()
        ref.step(); fus.step()
    for ga, gb in zip(a, b):
        pa, pb = ga["params"][0], gb["params"][0]
        # same rule and operation order; single roundings differ (sqrt(v) * rsqrt(bc2) here, a division in torch), which
        # leaves <= 1 ulp per step on O(1) parameters: 5e-6 after 30 steps at the largest learning rate
        assert torch.allclose(pa, pb, rtol=2e-5, atol=2e-5), (ga["name"], float((pa - pb).abs().max()))
        sa, sb = ref.state[pa], fus.state[pb]
        assert float(sa["step"]) == float(sb["step"])
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=1e-5, atol=1e-6 * float(sa["exp_avg"].abs().max()))
        assert torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=1e-5, atol=1e-6 * float(sa["exp_avg_sq"].abs().max()))


def test_survives_densification_style_state_surgery():
    """cat_tensors_to_optimizer / _prune_optimizer of the reference replace parameters and state tensors in place."""
    b = _groups(3)
    fus = FusedAdam(b, lr=0.0, eps=1e-15)
    for g in b:
        g["params"][0].grad = torch.ones_like(g["params"][0])
    fus.step()
    for g in b:                                                  # append 100 rows (zeros state), then prune every third row
        old = g["params"][0]
        st = fus.state.pop(old)
        ext = torch.zeros((100,) + old.shape[1:], device="cuda")
        new = torch.nn.Parameter(torch.cat([old.detach(), ext], 0)[::3].contiguous())
        st["exp_avg"] = torch.cat([st["exp_avg"], torch.zeros_like(ext)], 0)[::3]          # non-contiguous on purpose
        st["exp_avg_sq"] = torch.cat([st["exp_avg_sq"], torch.zeros_like(ext)], 0)[::3]
        g["params"][0] = new; fus.state[new] = st
        new.grad = torch.full_like(new, 0.5)
    before = [g["params"][0].detach().clone() for g in b]
    fus.step()
    for g, p0 in zip(b, before):
        p = g["params"][0]
        assert torch.isfinite(p).all() and (p != p0).any() and float(fus.state[p]["step"]) == 2.0
        assert fus.state[p]["exp_avg"].is_contiguous() and fus.state[p]["exp_avg"].shape == p.shape


def test_rejects_what_it_does_not_cover():
    p = [torch.nn.Parameter(torch.zeros(4, device="cuda"))]
    with pytest.raises(NotImplementedError):
        FusedAdam(p, weight_decay=0.1)
    with pytest.raises(NotImplementedError):
        FusedAdam(p, amsgrad=True)


# ---- the SH coefficients straight from the factors (round 6: C ABI ibgs_adam_step_sh, FusedAdam.step(sh_factors=...)) --------------------------------------
def _sh_setup(P, M, split, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    xyz = torch.nn.Parameter(torch.randn(P, 3, device="cuda", generator=g) * 3.0)
    if split:
        sh = [torch.nn.Parameter(torch.randn(P, 1, 3, device="cuda", generator=g)), torch.nn.Parameter(torch.randn(P, M - 1, 3, device="cuda", generator=g))]
        groups = [{"params": [xyz], "lr": 1.6e-4, "name": "xyz"}, {"params": [sh[0]], "lr": 2.5e-3, "name": "f_dc"}, {"params": [sh[1]], "lr": 1.25e-4, "name": "f_rest"}]
    else:
        sh = [torch.nn.Parameter(torch.randn(P, M, 3, device="cuda", generator=g))]
        groups = [{"params": [xyz], "lr": 1.6e-4, "name": "xyz"}, {"params": [sh[0]], "lr": 2.5e-3, "name": "shs"}]
    return xyz, sh, FusedAdam(groups, lr=0.0, eps=1e-15)


@pytest.mark.parametrize("deg,M,split,V,P", [(3, 16, True, 1, 5003), (2, 16, True, 2, 4097), (3, 16, False, 1, 5003), (1, 4, False, 1, 777), (0, 1, False, 3, 64), (1, 4, True, 1, 130)])
def test_sh_step_from_factors_is_the_dense_step_bit_for_bit(deg, M, split, V, P):
    from ibgs_amd.shgrad import sh_grad_from_views
    xa, sha, oa = _sh_setup(P, M, split, 3)          # dense: ibgs_sh_grad_from_views -> .grad -> ibgs_adam_step
    xb, shb, ob = _sh_setup(P, M, split, 3)          # factored: ibgs_adam_step_sh
    gen = torch.Generator(device="cuda").manual_seed(17)
    for it in range(4):
        dcolor = torch.randn(V, P, 3, device="cuda", generator=gen) * (10.0 ** (it - 2))
        dcolor[:, ::7] = 0.0                          # Gaussians that reached no pixel
        cams = torch.randn(V, 3, device="cuda", generator=gen) * 5.0
        gx = torch.randn(P, 3, device="cuda", generator=gen) * 1e-2
        dense = sh_grad_from_views(xa.detach(), cams, dcolor, deg, M)
        if split:
            sha[0].grad, sha[1].grad = dense[:, :1].contiguous(), dense[:, 1:].contiguous()
        else:
            sha[0].grad = dense
        xa.grad, xb.grad = gx.clone(), gx.clone()
        oa.step()
        items = [{"dcolor": dcolor[v], "campos": cams[v], "degree": deg, "M": M} for v in range(V)]
        ob.step(sh_factors=items, sh_params=tuple(shb), means3D=xb)
        torch.cuda.synchronize()
        for pa, pb in zip([xa] + sha, [xb] + shb):
            assert torch.equal(pa, pb), (it, tuple(pa.shape), float((pa.detach() - pb.detach()).abs().max()))
            sa, sb = oa.state[pa], ob.state[pb]
            assert float(sa["step"]) == float(sb["step"]) == it + 1
            assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
        assert all(p.grad is None for p in shb)
        for p in sha:
            p.grad = None


def test_sh_step_from_the_rasterizers_own_factors():
    """Through the op: one backward with the dense dL/dsh + the plain step against `capture_sh_factors()` + step(sh_factors=...), in the deterministic backward mode."""
    from ibgs_amd import rasterizer
    from tests import hipref
    from tests.scenes import scene
    inp = scene(P=20000, W=320, H=200, deg=3, seed=4, opacity="trained")
    tgt = torch.rand(3, 200, 320, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    old = rasterizer.DETERMINISTIC
    rasterizer.DETERMINISTIC = True
    try:
        res = []
        for factored in (False, True):
            outs, lv, _ = hipref.run_forward(inp)
            params = [torch.nn.Parameter(lv[k].detach().clone()) for k in ("means3D", "shs")]
            opt = FusedAdam([{"params": [params[0]], "lr": 1.6e-4}, {"params": [params[1]], "lr": 2.5e-3}], lr=0.0, eps=1e-15)
            loss = (outs["color"] - tgt).abs().mean()
            if factored:
                with rasterizer.capture_sh_factors() as items:
                    loss.backward()
                assert lv["shs"].grad is None and len(items) == 1
                params[0].grad = lv["means3D"].grad
                opt.step(sh_factors=items, sh_params=(params[1],), means3D=params[0])
            else:
                loss.backward()
                params[0].grad, params[1].grad = lv["means3D"].grad, lv["shs"].grad
                opt.step()
            torch.cuda.synchronize()
            res.append([p.detach().clone() for p in params] + [opt.state[params[1]]["exp_avg"].clone(), opt.state[params[1]]["exp_avg_sq"].clone()])
        assert torch.equal(res[0][0], res[1][0])          # positions: untouched by the factoring
        # the SH tensors: preprocess_bwd's own dL/dsh and the expansion of its factor are the same products of numbers that agree to an ulp (the direction is
        # normalised in two places) -- but Adam's first step is lr * g / |g|, so where a basis polynomial cancels to ~0 an ulp is visible.  Nearly every element
        # is identical, none is farther off than a sign flip could make it, and the moments agree to rounding.
        for a, b, name in zip(res[0][1:], res[1][1:], ("shs", "exp_avg", "exp_avg_sq")):
            same = float((a == b).float().mean())
            print("[sh step through the op] %s: %.4f %% of the elements bit-identical, max |diff| %.2e" % (name, 100 * same, float((a - b).abs().max())))
            assert same > (0.99 if name == "shs" else 0.95), (name, same)
        assert float((res[0][1] - res[1][1]).abs().max()) <= 2.01 * 2.5e-3
        assert torch.allclose(res[0][2], res[1][2], rtol=1e-4, atol=1e-7 * float(res[0][2].abs().max()))
        assert torch.allclose(res[0][3], res[1][3], rtol=1e-4, atol=1e-7 * float(res[0][3].abs().max()))
    finally:
        rasterizer.DETERMINISTIC = old


def test_sh_step_rejects_inconsistent_input():
    xb, shb, ob = _sh_setup(100, 16, True, 1)
    item = {"dcolor": torch.zeros(100, 3, device="cuda"), "campos": torch.zeros(3, device="cuda"), "degree": 3, "M": 16}
    with pytest.raises(ValueError):
        ob.step(sh_factors=[item])                                            # no sh_params / means3D
    with pytest.raises(ValueError):
        ob.step(sh_factors=[item], sh_params=(shb[1],), means3D=xb)           # 15 coefficients against M = 16
    shb[0].grad = torch.zeros_like(shb[0])
    with pytest.raises(RuntimeError, match="dense .grad"):
        ob.step(sh_factors=[item], sh_params=tuple(shb), means3D=xb)
