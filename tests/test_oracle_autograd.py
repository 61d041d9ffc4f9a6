"""Independent check of the oracle's hand-derived backward (B1, B3, B4): a differentiable float64
PyTorch restatement of projection -> EWA -> SH -> per-tile alpha blend (+ un-normalised normal blend),
differentiated by torch.autograd, must reproduce oracle.backward().

Discrete decisions (tile membership, draw order, the alpha >= 1/255 / power <= 0 tests, the T < 1e-4
termination) are not differentiable in the reference either; they are taken from the oracle's forward
state and asserted to agree with the float64 recomputation.  Two reference conventions are mirrored:
alpha = min(0.99, o*G) passes gradient straight through the clamp (backward.cu:786 uses o * dL/dalpha
unconditionally) and the quaternion is used un-normalised (backward.cu:437)."""
import numpy as np
import pytest
import torch

import oracle
from ibgs_amd import synthetic as syn
from tests.metrics import rel_l2

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)


def sh_color(deg, sh, d):
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = C0 * sh[:, 0]
    if deg > 0:
        r = r - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = r + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6] + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8]
    if deg > 2:
        r = (r + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10] + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
             + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12] + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13]
             + C3[5] * z * (xx - yy) * sh[:, 14] + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return r


def torch_render(inp, fwd, leaves, with_normal):
    dd = torch.float64
    W, H = inp["W"], inp["H"]
    vm = torch.tensor(inp["viewmatrix"], dtype=dd).reshape(4, 4)      # transposed: row-vector convention p_h @ vm
    pm = torch.tensor(inp["projmatrix"], dtype=dd).reshape(4, 4)
    campos = torch.tensor(inp["campos"], dtype=dd)
    xyz, shs, opac, scales, quat = leaves["means3D"], leaves["shs"], leaves["opacities"], leaves["scales"], leaves["rotations"]
    P = xyz.shape[0]
    tanx, tany = inp["tanfovx"], inp["tanfovy"]
    fx, fy = W / (2 * tanx), H / (2 * tany)
    ph = torch.cat([xyz, torch.ones(P, 1, dtype=dd)], 1)
    pv = ph @ vm
    hom = ph @ pm
    pw = 1.0 / (hom[:, 3] + 1e-7)
    px = ((hom[:, 0] * pw + 1.0) * W - 1.0) * 0.5
    py = ((hom[:, 1] * pw + 1.0) * H - 1.0) * 0.5
    r, x, y, z = quat[:, 0], quat[:, 1], quat[:, 2], quat[:, 3]
    Rq = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                      2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                      2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).view(P, 3, 3)
    Lm = Rq * scales[:, None, :]
    Sigma = Lm @ Lm.transpose(1, 2)
    tz = pv[:, 2]
    tx = torch.clamp(pv[:, 0] / tz, -1.3 * tanx, 1.3 * tanx) * tz
    ty = torch.clamp(pv[:, 1] / tz, -1.3 * tany, 1.3 * tany) * tz
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * tx / (tz * tz), zero, fy / tz, -fy * ty / (tz * tz)], -1).view(P, 2, 3)
    Rv = vm[:3, :3].t()                                                  # true world->view rotation
    A = J @ Rv
    cov = A @ Sigma @ A.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3; b = cov[:, 0, 1]; c = cov[:, 1, 1] + 0.3
    det = a * c - b * b
    ca, cb, cc = c / det, -b / det, a / det
    d = xyz - campos
    d = d / d.norm(dim=1, keepdim=True)
    rgb = torch.clamp_min(sh_color(inp["sh_degree"], shs, d) + 0.5, 0.0)
    bg = torch.tensor(inp["bg"], dtype=dd)
    nrm = torch.tensor(inp["all_map"][:, :3], dtype=dd) if with_normal else None

    color = torch.zeros(3, H, W, dtype=dd)
    normal = torch.zeros(3, H, W, dtype=dd)
    gxn = (W + 15) // 16
    ncon = fwd["n_contrib"].reshape(H, W)
    for t, (r0, r1) in enumerate(fwd["ranges"]):
        if r1 <= r0:
            ty0, tx0 = (t // gxn) * 16, (t % gxn) * 16
            hh, ww = min(16, H - ty0), min(16, W - tx0)
            color[:, ty0:ty0 + hh, tx0:tx0 + ww] = bg[:, None, None]
            continue
        ids = torch.tensor(fwd["point_list"][r0:r1].astype(np.int64))
        ty0, tx0 = (t // gxn) * 16, (t % gxn) * 16
        hh, ww = min(16, H - ty0), min(16, W - tx0)
        ys, xs = torch.meshgrid(torch.arange(ty0, ty0 + hh, dtype=dd), torch.arange(tx0, tx0 + ww, dtype=dd), indexing="ij")
        dx = px[ids][None, None, :] - xs[..., None]
        dy = py[ids][None, None, :] - ys[..., None]
        power = -0.5 * (ca[ids] * dx * dx + cc[ids] * dy * dy) - cb[ids] * dx * dy
        G = torch.exp(power)
        raw = opac[ids, 0] * G
        alpha = raw + (torch.clamp_max(raw, 0.99) - raw).detach()        # straight-through clamp
        valid = (power <= 0) & (alpha >= 1.0 / 255.0)
        kk = torch.arange(ids.numel())[None, None, :]
        n_c = torch.tensor(ncon[ty0:ty0 + hh, tx0:tx0 + ww].astype(np.int64))[..., None]
        use = valid & (kk < n_c)
        am = torch.where(use, alpha, torch.zeros_like(alpha))
        Tcum = torch.cumprod(1 - am, dim=-1)
        Tbefore = torch.cat([torch.ones_like(Tcum[..., :1]), Tcum[..., :-1]], -1)
        w = am * Tbefore
        color[:, ty0:ty0 + hh, tx0:tx0 + ww] = torch.einsum("hwk,kc->chw", w, rgb[ids]) + Tcum[..., -1][None] * bg[:, None, None]
        if with_normal:
            normal[:, ty0:ty0 + hh, tx0:tx0 + ww] = torch.einsum("hwk,kc->chw", w, nrm[ids])
        # the discrete decisions must agree with the oracle's fp32 run
        Tf = torch.tensor(fwd["final_T"].reshape(H, W)[ty0:ty0 + hh, tx0:tx0 + ww], dtype=dd)
        assert torch.allclose(Tcum[..., -1].detach(), Tf, atol=2e-5), "blend state diverged from the oracle"
    return color, normal


@pytest.mark.parametrize("deg", [0, 3])
def test_oracle_backward_matches_autograd(deg):
    P, W, H = 60, 48, 40
    g = syn.make_gaussians(P, seed=11 + deg, sh_degree=deg, opacity="trained", extent=0.9)
    g["scales"] *= 6.0                                                   # a few tiles per Gaussian at this tiny P
    cam = syn.make_camera(W, H, azimuth_deg=30.0, radius=3.0)
    inp = dict(g)
    inp.update({"W": W, "H": H, "tanfovx": cam["tanfovx"], "tanfovy": cam["tanfovy"], "viewmatrix": cam["viewmatrix"],
                "projmatrix": cam["projmatrix"], "campos": cam["campos"], "bg": np.array([0.1, 0.2, 0.3], np.float32),
                "sh_degree": deg, "scale_modifier": 1.0})
    inp["all_map"] = syn.plane_all_map(g["means3D"], g["scales"], g["rotations"], cam)
    with_normal = (deg == 3)
    if with_normal:
        # render_geo with an always-invalid source (depth 0) exercises the normal blend without the warp path
        inp.update(render_geo=True, n_src=1, buffer_length=4)
    fwd = oracle.forward(inp)
    assert fwd["num_rendered"] > 100 and fwd["n_contrib"].max() >= 5
    rng = np.random.default_rng(5)
    gc = rng.normal(size=(3, H, W)).astype(np.float32)
    gn = rng.normal(size=(3, H, W)).astype(np.float32) if with_normal else None
    ob = oracle.backward(inp, fwd, gc, gn, None, None)

    leaves = {k: torch.tensor(inp[k], dtype=torch.float64, requires_grad=True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    color, normal = torch_render(inp, fwd, leaves, with_normal)
    np.testing.assert_allclose(color.detach().numpy(), fwd["color"], atol=2e-5)
    loss = (color * torch.tensor(gc, dtype=torch.float64)).sum()
    if with_normal:
        np.testing.assert_allclose(normal.detach().numpy(), fwd["normal_map"], atol=2e-5)
        loss = loss + (normal * torch.tensor(gn, dtype=torch.float64)).sum()
    loss.backward()
    pairs = {"means3D": "dL_dmeans3D", "shs": "dL_dsh", "opacities": "dL_dopacity", "scales": "dL_dscales", "rotations": "dL_drotations"}
    for k, ok in pairs.items():
        err = rel_l2(ob[ok], leaves[k].grad.numpy())
        assert err < 2e-4, "%s: relative L2 %.3e vs autograd" % (k, err)
