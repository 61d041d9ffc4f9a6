"""The colour backward's balanced launch order (render_bwd.hip: tile_order_kernel): a performance heuristic that must never change what
is computed.  Checked directly: what the forward reports as walked per tile is the tile's largest n_contrib; the order the backward
built holds every tile exactly once, empty slots are marked, and frames with more tiles than wave slots are launched heaviest first.
(That the gradients are right under it is what every parity test with one wave per tile checks.)"""
import numpy as np
import pytest
import torch

from ibgs_amd import _lib, rasterizer
from tests import hipref
from tests.metrics import rel_l2
from tests.scenes import scene

pytestmark = pytest.mark.gpu


def img_arena(outs):
    """The forward's image arena (a saved tensor of the autograd node): taken BEFORE backward() frees the node's references; the backward writes
    the launch order into the same memory."""
    return outs["color"].grad_fn.saved_tensors[-1]


def order_state(inp, img_t):
    lib = _lib.load()
    W, H = int(inp["W"]), int(inp["H"])
    img = img_t.cpu().numpy()
    gx, gy = (W + 15) // 16, (H + 15) // 16
    nt = gx * gy
    nslots = (nt + 1023) // 1024 * 1024

    def view(name, count):
        off = lib.ibgs_img_offset(W, H, name.encode())
        assert off >= 0, name
        return np.frombuffer(img.tobytes()[off:off + 4 * count], dtype=np.uint32).copy()
    return nt, nslots, view("meta", 32), view("tile_walked", nt), view("tile_order", nslots), view("n_contrib", W * H).reshape(H, W), (gx, gy)


@pytest.mark.parametrize("W,H,P", [(1920, 1088, 30000), (208, 144, 2000), (2560, 1456, 30000), (3840, 2160, 20000)])          # the last: more than 16 K tiles (the order kernel reads its keys twice)
def test_every_tile_once_and_walked_is_the_largest_n_contrib(W, H, P):
    old = rasterizer.WAVE_SHAPE
    rasterizer.WAVE_SHAPE = "tile"
    try:
        inp = scene(P=P, W=W, H=H, deg=1, seed=11, opacity="trained", scale_mul=1.0 if W > 1000 else 3.0)
        outs, lv, _ = hipref.run_forward(inp)
        g = torch.randn(3, H, W, device="cuda")
        img = img_arena(outs)
        (outs["color"] * g).sum().backward()
        torch.cuda.synchronize()
        nt, nslots, meta, walked, order, nc, (gx, gy) = order_state(inp, img)
    finally:
        rasterizer.WAVE_SHAPE = old
    assert meta[10] == 1
    pad = np.zeros((gy * 16, gx * 16), np.uint32); pad[:H, :W] = nc
    assert np.array_equal(walked, pad.reshape(gy, 16, gx, 16).max(axis=(1, 3)).reshape(-1))
    tiles = order[order != 0xFFFFFFFF]
    assert tiles.size == nt and np.array_equal(np.sort(tiles), np.arange(nt, dtype=np.uint32)), "every tile exactly once"
    assert (order == 0xFFFFFFFF).sum() == nslots - nt
    if nslots // 1024 > 8:          # more tiles than wave slots: plain descending order of the sort's key (the top 10 bits of the walked length)
        assert np.array_equal(np.nonzero(order != 0xFFFFFFFF)[0], np.arange(nt))
        sh = max(0, int(walked.max()).bit_length() - 10)
        key = walked[order[:nt]] >> sh
        assert (np.diff(key.astype(np.int64)) <= 0).all()
    else:                          # stratum s (1 024 ranks) sits in round (s + 1) % rounds: the heaviest tiles are not in the first 1 024 workgroups
        r = nslots // 1024
        if r > 1 and walked.max() > 0:
            first = order[:1024]; first = first[first != 0xFFFFFFFF]
            assert walked[first].astype(np.float64).mean() <= walked.astype(np.float64).mean()


def test_quadrant_waves_record_per_wave():
    """Small frames run one wave per 8 x 8 quadrant: four words per tile, meta[10] = 4 (the backward of that shape does not reorder anything)."""
    old = rasterizer.WAVE_SHAPE
    rasterizer.WAVE_SHAPE = "quadrant"
    try:
        W, H = 208, 144
        inp = scene(P=1500, W=W, H=H, deg=1, seed=12, opacity="trained", scale_mul=3.0)
        outs, lv, _ = hipref.run_forward(inp)
        img = img_arena(outs)
        (outs["color"] * torch.randn(3, H, W, device="cuda")).sum().backward()
        torch.cuda.synchronize()
        nt, nslots, meta, _, _, nc, (gx, gy) = order_state(inp, img)
        lib = _lib.load()
        off = lib.ibgs_img_offset(W, H, b"tile_walked")
        walked = np.frombuffer(img.cpu().numpy().tobytes()[off:off + 16 * nt], dtype=np.uint32).reshape(nt, 4)
    finally:
        rasterizer.WAVE_SHAPE = old
    assert meta[10] == 4
    pad = np.zeros((gy * 16, gx * 16), np.uint32); pad[:H, :W] = nc
    q = pad.reshape(gy, 2, 8, gx, 2, 8).max(axis=(2, 5))          # [ty, qy, tx, qx]
    assert np.array_equal(walked, q.transpose(0, 2, 1, 3).reshape(nt, 4))          # wave of the tile = 2 * qy + qx


def _step(inp, lv_from=None):
    outs, lv, _ = hipref.run_forward(inp)
    img = img_arena(outs)
    color = outs["color"].detach().clone()
    (outs["color"] * torch.ones(3, int(inp["H"]), int(inp["W"]), device="cuda")).sum().backward()
    torch.cuda.synchronize()
    return color, img, lv


@pytest.mark.parametrize("W,H", [(1920, 1088), (3840, 2160)])          # all tiles resident (snake order) / four times the slots (heaviest first)
def test_forward_order_hint_changes_nothing_and_bad_hints_are_ignored(monkeypatch, W, H):
    """The backward's launch order comes back as a hint to the same camera's next forward (rasterizer._order_hints): the image, the
    per-pixel state and the gradients are those of a forward without it, bit for bit; a hint that is not a tile order (a duplicated
    tile, a tile out of range) is recognised on the device (meta[11] = 0) and ignored."""
    old = rasterizer.WAVE_SHAPE
    rasterizer.WAVE_SHAPE = "tile"
    rasterizer._order_hints.clear()
    try:
        inp = scene(P=20000, W=W, H=H, deg=1, seed=21, opacity="trained")
        # one persistent device tensor for the view matrix, as the reference keeps per camera: that is what the cache keys on
        inp = dict(inp); vm = torch.as_tensor(np.ascontiguousarray(inp["viewmatrix"]), dtype=torch.float32, device="cuda")
        monkeypatch.setattr(hipref, "settings_from", (lambda f: (lambda i, dev, debug: _with_vm(f(i, dev, debug), vm)))(hipref.settings_from))
        c0, img0, lv0 = _step(inp)
        assert order_state(inp, img0)[2][11] == 0 and len(rasterizer._order_hints) == 1          # first frame: no hint yet; the backward left one
        c1, img1, lv1 = _step(inp)
        st1 = order_state(inp, img1)
        assert st1[2][11] == 1, "the second forward of the camera runs under the first backward's order"
        assert torch.equal(c0, c1)
        assert np.array_equal(order_state(inp, img0)[5], st1[5])          # n_contrib
        for k in ("means3D", "opacities", "scales", "rotations", "shs"):
            assert rel_l2(lv1[k].grad.cpu().numpy(), lv0[k].grad.cpu().numpy()) < 1e-5, k          # (the atomics land in another order on every launch)
        hint = next(iter(rasterizer._order_hints.values())).view(torch.int32)
        good = hint.clone()
        for breakage in ("duplicate", "range"):
            hint.copy_(good)
            idx = int((good != -1).nonzero()[5])
            hint[idx] = good[(good != -1).nonzero()[6]] if breakage == "duplicate" else 10 ** 6
            outs, lv, _ = hipref.run_forward(inp)
            torch.cuda.synchronize()
            assert order_state(inp, img_arena(outs))[2][11] == 0, breakage
            assert torch.equal(outs["color"].detach(), c0), breakage
    finally:
        rasterizer.WAVE_SHAPE = old
        rasterizer._order_hints.clear()


def _with_vm(settings, vm):
    return settings._replace(viewmatrix=vm)


def test_geo_forward_under_an_order_hint(monkeypatch):
    """rasterizer.ORDER_HINT_GEO (off by default): the geo forward launched over the geo backward's order, two workgroups per slot -- every
    output plane bit-identical to the forward without a hint."""
    from tests.scenes import add_sources
    old = (rasterizer.WAVE_SHAPE, rasterizer.ORDER_HINT_GEO)
    rasterizer.WAVE_SHAPE = "tile"; rasterizer.ORDER_HINT_GEO = True
    rasterizer._order_hints.clear()
    try:
        W, H = 1920, 1088
        inp = add_sources(scene(P=15000, W=W, H=H, deg=1, seed=23, opacity="trained", planes=True), n_src=2, seed=3)
        vm = torch.as_tensor(np.ascontiguousarray(inp["viewmatrix"]), dtype=torch.float32, device="cuda")
        monkeypatch.setattr(hipref, "settings_from", (lambda f: (lambda i, dev, debug: _with_vm(f(i, dev, debug), vm)))(hipref.settings_from))
        res = []
        for it in range(2):
            outs, lv, _ = hipref.run_forward(inp)
            img = img_arena(outs)
            keep = {k: v.detach().clone() for k, v in outs.items() if torch.is_tensor(v)}
            loss = sum((v.float() * 0.5).sum() for k, v in outs.items() if torch.is_tensor(v) and v.requires_grad)
            loss.backward()
            torch.cuda.synchronize()
            res.append((keep, order_state(inp, img)[2][11]))
        assert res[0][1] == 0 and res[1][1] == 1
        for k in res[0][0]:
            assert torch.equal(res[0][0][k], res[1][0][k]), k
    finally:
        rasterizer.WAVE_SHAPE, rasterizer.ORDER_HINT_GEO = old
        rasterizer._order_hints.clear()
