"""Second, independent restatement of the reference's median / warp backward (row B2 of SURVEY 8(a)):
DPR/cuda_rasterizer/backward.cu:692-771 (the "Backward pass for the median buffer" block of renderCUDA) and
bilinearInterpolateBackward (backward.cu:55-109), with the texture semantics of rasterizer_impl.cu:120-126.

TEST INFRASTRUCTURE ONLY.  It exists because the C oracle's B2 (oracle/ibgs_oracle.c, one (pixel, Gaussian) pair at a time,
fp32) was the only statement of that code path in the repository; every geo-gradient parity number rests on it.  This one is
written from the reference formulas again, in another shape: float64 torch, a whole tile at a time as (pixel x list entry)
arrays, the per-source loop replaced by closed forms over the slot axis (a cumulative product for the -1 terminator, a
cumulative sum for the growing depth gradient of quirk Q2), transmittance by a forward cumulative product instead of the
back-to-front division.  It computes ONLY what B2 adds, so it is compared with the oracle on losses that touch nothing but
the median depth and the warped colours (dL/dcolor = dL/dnormal = 0).

Reference behaviour restated (line numbers of backward.cu):
  * window test (693): `contributor >= min-1 && contributor <= max-1`, contributor unsigned, min/max int: with min == 0 the
    left side compares against 0xFFFFFFFF and the block is skipped (quirk Q4); contributor = 0-based list position.
  * depth of the pair (695-699): d = -dist / (n . (ray.x, ray.y, 1) + 1e-8), only d > 0 continues.
  * median term (703-705): g_d = dL/dmedian * w / sum_w, dL/dalpha += dL/dmedian * (d - median) / sum_w, w = alpha * T.
  * per valid slot m until the -1 terminator (707-709), source s = valid_src_indices[m]: X = (px-cx) d / fx, Y = (py-cy) d / fy,
    Z = d mapped by the row-major ref_to_src[s]; u = X' fx / Z' + cx, v likewise; only 0 <= u <= W-1 and 0 <= v <= H-1 (719)
    continue: colour c = linear fetch at (u + 0.5, v + 0.5) (721-725); g_c = dL/dwarped[m] * w / sw[m];
    dL/dalpha += dL/dwarped[m] . (c - warped[m]) / sw[m] (733-734);  du, dv from FOUR LINEAR-FILTERED fetches at the integer
    coordinates floor(u + 0.5) (+1) -- each returns the mean of a 2x2 texel block (quirk Q3) -- with weights fu = u + 0.5 -
    floor(u + 0.5) (62-100); dp/dd analytic (740-751); g_d += du * dpx + dv * dpy (757);  THEN, still inside the slot loop and
    the in-bounds branch (759-763, quirk Q2): dL/ddist += -g_d / tmp, dL/dn += g_d * dist / tmp^2 * (ray.x, ray.y, 1) with the
    CUMULATIVE g_d.
  * afterwards (773-805): dL/dalpha *= T;  dL/dG = o * dL/dalpha;  mean2D += dL/dG * dG/dd * (0.5 W, 0.5 H);
    conic (x, y, w) += -0.5 G (dx dx, dx dy, dy dy) dL/dG;  opacity += G * dL/dalpha  (Q5: no dependence of later T on alpha).
"""
import numpy as np
import torch

DD = torch.float64


def _tex(img, x, y):
    """Linear-filtered, clamp-addressed, unnormalised fetch of img (C, H, W) at texture coordinates (x, y) (any shape):
    texel centres at i + 0.5.  Returns (C, *x.shape)."""
    C, H, W = img.shape
    xb, yb = x - 0.5, y - 0.5
    i, j = torch.floor(xb), torch.floor(yb)
    a, b = xb - i, yb - j
    i0 = i.long().clamp(0, W - 1); i1 = (i.long() + 1).clamp(0, W - 1)
    j0 = j.long().clamp(0, H - 1); j1 = (j.long() + 1).clamp(0, H - 1)
    flat = img.reshape(C, -1)
    pick = lambda jj, ii: flat[:, (jj * W + ii).reshape(-1)].reshape((C,) + tuple(x.shape))
    return (1 - a) * (1 - b) * pick(j0, i0) + a * (1 - b) * pick(j0, i1) + (1 - a) * b * pick(j1, i0) + a * b * pick(j1, i1)


def b2_gradients(inp, fwd, g_depth, g_warped):
    """-> dict(dL_dall_map (P,5), dL_dmeans2D (P,2), dL_dconic (P,3) [x, y, w], dL_dopacity (P,)) of the B2 terms alone."""
    W, H = int(inp["W"]), int(inp["H"]); HW = W * H
    P = inp["means3D"].shape[0]
    fx, fy = W / (2.0 * float(inp["tanfovx"])), H / (2.0 * float(inp["tanfovy"]))
    fx, fy = float(np.float32(fx)), float(np.float32(fy))            # the op holds them as float (rasterizer_impl.cu:362-363)
    cx, cy = W * 0.5, H * 0.5
    t = lambda a: torch.tensor(np.asarray(a), dtype=DD)
    xy = t(fwd["means2D"]); con = t(fwd["conic_opacity"]); am = t(inp["all_map"])
    r2s = t(np.asarray(inp["ref_to_src"]).reshape(-1, 16))
    img = t(inp["src_images"]).reshape(-1, 3, H, W)
    med = t(fwd["median_depth"]).reshape(HW); sumw = t(fwd["cache_sum_w"]).reshape(HW)
    lo = torch.tensor(fwd["cache_low"].astype(np.int64)); hi = torch.tensor(fwd["cache_high"].astype(np.int64))
    vidx = torch.tensor(fwd["valid_src_idx"].astype(np.int64)); vw = t(fwd["valid_src_w"])          # (5, HW)
    warped = t(fwd["warped_image"]).reshape(5, 3, HW)
    gd = t(g_depth).reshape(HW); gw = t(g_warped).reshape(5, 3, HW)
    ncon = torch.tensor(fwd["n_contrib"].astype(np.int64))
    out_am = torch.zeros(P, 5, dtype=DD); out_m = torch.zeros(P, 2, dtype=DD)
    out_c = torch.zeros(P, 3, dtype=DD); out_o = torch.zeros(P, dtype=DD)
    gxn = (W + 15) // 16
    M = 5
    # slot validity: everything before the first -1 (the terminator is only written when fewer than 5 slots are valid)
    slot_ok = torch.cumprod((vidx != -1).to(DD), dim=0) > 0                                            # (5, HW)
    for tile, (r0, r1) in enumerate(fwd["ranges"]):
        if r1 <= r0:
            continue
        ids = torch.tensor(fwd["point_list"][r0:r1].astype(np.int64)); K = ids.numel()
        ty0, tx0 = (tile // gxn) * 16, (tile % gxn) * 16
        ys, xs = torch.meshgrid(torch.arange(ty0, min(ty0 + 16, H)), torch.arange(tx0, min(tx0 + 16, W)), indexing="ij")
        pix = (ys * W + xs).reshape(-1); n = pix.numel()
        pxf, pyf = xs.reshape(-1, 1).to(DD), ys.reshape(-1, 1).to(DD)
        # ---- blend weights of every (pixel, entry) pair, front to back
        dx = xy[ids, 0][None] - pxf; dy = xy[ids, 1][None] - pyf
        A, B, Cc, op = con[ids, 0][None], con[ids, 1][None], con[ids, 2][None], con[ids, 3][None]
        power = -0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy
        G = torch.exp(power)
        alpha = torch.clamp(op * G, max=0.99)
        k = torch.arange(K)[None]
        use = (power <= 0) & (alpha >= 1.0 / 255.0) & (k < ncon[pix][:, None])
        a_eff = torch.where(use, alpha, torch.zeros_like(alpha))
        Tcum = torch.cumprod(1 - a_eff, dim=1)
        Tfront = torch.cat([torch.ones(n, 1, dtype=DD), Tcum[:, :-1]], dim=1)
        w = a_eff * Tfront
        # ---- which pairs enter the median block
        lo_p, hi_p = lo[pix][:, None], hi[pix][:, None]
        window = use & (lo_p != 0) & (k >= lo_p - 1) & (k <= hi_p - 1)          # lo == 0: unsigned wrap, never true (Q4)
        rayx, rayy = (pxf - cx) / fx, (pyf - cy) / fy
        nx, ny, nz, dist = am[ids, 0][None], am[ids, 1][None], am[ids, 2][None], am[ids, 4][None]
        tmp = nx * rayx + ny * rayy + nz + 1.0e-8
        depth = -dist / tmp
        act = window & (depth > 0)
        if not bool(act.any()):
            continue
        sw_p = sumw[pix][:, None]
        g_base = gd[pix][:, None] * w / sw_p                                        # dL/dmedian * w / sum_w
        dLda = gd[pix][:, None] * (depth - med[pix][:, None]) / sw_p
        # ---- slot axis: (M, n, K)
        s_idx = vidx[:, pix].clamp(min=0)                                           # (M, n)
        ok_m = slot_ok[:, pix][:, :, None] & act[None]
        R = r2s[s_idx]                                                              # (M, n, 16)
        r = lambda q: R[:, :, q][:, :, None]
        X, Y, Z = (pxf - cx) * depth / fx, (pyf - cy) * depth / fy, depth
        tx = r(0) * X + r(1) * Y + r(2) * Z + r(3)
        ty = r(4) * X + r(5) * Y + r(6) * Z + r(7)
        tz = r(8) * X + r(9) * Y + r(10) * Z + r(11)
        tz_safe = torch.where(ok_m, tz, torch.ones_like(tz))
        u = tx * fx / tz_safe + cx; v = ty * fy / tz_safe + cy
        inb = ok_m & (u >= 0) & (u <= W - 1) & (v >= 0) & (v <= H - 1)
        u = torch.where(inb, u, torch.zeros_like(u)); v = torch.where(inb, v, torch.zeros_like(v))
        col = torch.zeros(M, 3, n, K, dtype=DD); dIu = torch.zeros_like(col); dIv = torch.zeros_like(col)
        uu, vv = u + 0.5, v + 0.5
        u0, v0 = torch.floor(uu), torch.floor(vv)
        fu, fv = uu - u0, vv - v0
        for m in range(M):
            for s in torch.unique(s_idx[m]).tolist():
                sel = inb[m] & (s_idx[m] == s)[:, None]
                if not bool(sel.any()):
                    continue
                im = img[s]
                c_here = _tex(im, uu[m][sel], vv[m][sel])                           # forward sample (721-725)
                I00 = _tex(im, u0[m][sel], v0[m][sel]); I01 = _tex(im, u0[m][sel] + 1, v0[m][sel])
                I10 = _tex(im, u0[m][sel], v0[m][sel] + 1); I11 = _tex(im, u0[m][sel] + 1, v0[m][sel] + 1)
                fu_s, fv_s = fu[m][sel], fv[m][sel]
                col[m][:, sel] = c_here
                dIu[m][:, sel] = -(1 - fv_s) * I00 + (1 - fv_s) * I01 - fv_s * I10 + fv_s * I11
                dIv[m][:, sel] = -(1 - fu_s) * I00 - fu_s * I01 + (1 - fu_s) * I10 + fu_s * I11
        sw_m = vw[:, pix][:, None, :, None]                                         # (M, 1, n, 1)
        sw_m = torch.where(sw_m != 0, sw_m, torch.ones_like(sw_m))
        gwp = gw[:, :, pix][:, :, :, None]                                          # (M, 3, n, 1)
        inb_f = inb.to(DD)
        dLda = dLda + (inb_f[:, None] * gwp * (col - warped[:, :, pix][:, :, :, None]) / sw_m).sum(dim=(0, 1))
        g_c = gwp * w[None, None] / sw_m                                            # (M, 3, n, K)
        du = (g_c * dIu).sum(1); dv = (g_c * dIv).sum(1)                            # (M, n, K)
        Av, Bv = rayx, rayy
        U = r(0) * Av + r(1) * Bv + r(2); V = r(4) * Av + r(5) * Bv + r(6); Wc = r(8) * Av + r(9) * Bv + r(10)
        den = torch.where(inb, Wc * depth + r(11), torch.ones_like(tz))
        dpx = fx * (U * r(11) - Wc * r(3)) / (den * den); dpy = fy * (V * r(11) - Wc * r(7)) / (den * den)
        from_col = inb_f * (du * dpx + dv * dpy)
        g_cum = g_base[None] + torch.cumsum(from_col, dim=0)                        # depth gradient after slot m (Q2)
        G_all = (inb_f * g_cum).sum(0)                                              # what the plane parameters receive, summed over in-bounds slots
        tmp2 = dist / (tmp * tmp)
        out_am.index_add_(0, ids, torch.stack([(G_all * tmp2 * rayx).sum(0), (G_all * tmp2 * rayy).sum(0), (G_all * tmp2).sum(0),
                                               torch.zeros(K, dtype=DD), (-G_all / tmp).sum(0)], dim=1))
        # ---- dL/dalpha of the block -> 2D mean, conic, opacity (773-805)
        dLda = torch.where(act, dLda, torch.zeros_like(dLda)) * Tfront
        dLdG = op * dLda
        gdx, gdy = G * dx, G * dy
        out_m.index_add_(0, ids, torch.stack([(dLdG * (-gdx * A - gdy * B) * (0.5 * W)).sum(0), (dLdG * (-gdy * Cc - gdx * B) * (0.5 * H)).sum(0)], dim=1))
        out_c.index_add_(0, ids, torch.stack([(-0.5 * gdx * dx * dLdG).sum(0), (-0.5 * gdx * dy * dLdG).sum(0), (-0.5 * gdy * dy * dLdG).sum(0)], dim=1))
        out_o.index_add_(0, ids, (G * dLda).sum(0))
    return {"dL_dall_map": out_am.numpy(), "dL_dmeans2D": out_m.numpy(), "dL_dconic": out_c.numpy(), "dL_dopacity": out_o.numpy()}
