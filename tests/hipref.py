"""Test helper: run the HIP operator (through the reference-style Python surface and the C ABI) on an
oracle-style numpy input dict and return numpy results, including the internal arena arrays that the
parity tests compare stage by stage."""
import ctypes

import numpy as np
import torch

from ibgs_amd import _lib
from ibgs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer


def _t(a, dev, dtype=torch.float32):
    return None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=dev)


def settings_from(inp, dev, debug=False):
    n_src = int(inp.get("n_src", 1)); H, W = int(inp["H"]), int(inp["W"])
    r2s = inp.get("ref_to_src"); scp = inp.get("src_cam_pos"); simg = inp.get("src_images"); sdep = inp.get("src_depths")
    r2s = torch.zeros(n_src, 16, device=dev) if r2s is None else _t(r2s, dev)
    scp = torch.zeros(n_src, 3, device=dev) if scp is None else _t(scp, dev)
    simg = torch.zeros(n_src, 3, H * W, device=dev) if simg is None else _t(simg, dev)
    sdep = torch.zeros(n_src, 1, H * W, device=dev) if sdep is None else _t(sdep, dev)
    return GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=float(inp["tanfovx"]), tanfovy=float(inp["tanfovy"]),
        bg=_t(inp["bg"], dev), scale_modifier=float(inp.get("scale_modifier", 1.0)),
        viewmatrix=_t(inp["viewmatrix"], dev), projmatrix=_t(inp["projmatrix"], dev),
        ref_to_src_list=r2s, src_cam_pos=scp, src_images=simg, src_rendered_depths=sdep,
        nb_src_images=n_src, buffer_length=int(inp.get("buffer_length", 4)),
        depth_error_threshold=float(inp.get("depth_thr", 0.01)), sh_degree=int(inp.get("sh_degree", 0)),
        campos=_t(inp["campos"], dev), prefiltered=False, render_geo=bool(inp.get("render_geo", False)),
        render_depth_only=bool(inp.get("render_depth_only", False)), debug=debug)


def leaf_inputs(inp, dev, requires_grad=True):
    P = inp["means3D"].shape[0]
    d = {}
    for k in ("means3D", "shs", "colors_precomp", "opacities", "scales", "rotations", "cov3D_precomp", "all_map"):
        v = inp.get(k)
        if v is None:
            d[k] = None
        else:
            t = _t(v, dev)
            if k == "opacities":
                t = t.reshape(P, 1)
            d[k] = t.requires_grad_(requires_grad)
    d["means2D"] = torch.zeros(P, 3, device=dev, requires_grad=requires_grad)
    d["means2D_abs"] = torch.zeros(P, 3, device=dev, requires_grad=requires_grad)
    return d


def run_forward(inp, dev="cuda", debug=False, requires_grad=True):
    st = settings_from(inp, dev, debug)
    lv = leaf_inputs(inp, dev, requires_grad)
    rast = GaussianRasterizer(st)
    outs = rast(means3D=lv["means3D"], means2D=lv["means2D"], means2D_abs=lv["means2D_abs"], opacities=lv["opacities"],
                shs=lv["shs"], colors_precomp=lv["colors_precomp"], scales=lv["scales"], rotations=lv["rotations"],
                cov3D_precomp=lv["cov3D_precomp"], all_map=lv["all_map"])
    names = ["color", "radii", "normal_map", "median_depth", "cam_feat", "warped_image", "min_depth_diff",
             "camera_ray", "use_first_src_frame_mask"]
    return dict(zip(names, outs)), lv, st


def internal_state(outs, inp):
    """Pull the arena arrays out of the autograd node of `outs['color']` (saved tensors)."""
    lib = _lib.load()
    node = next((outs[k].grad_fn for k in ("color", "median_depth", "normal_map") if outs.get(k) is not None and outs[k].grad_fn is not None), None)          # (a depth-only pass has no colour)
    if node is None:
        raise RuntimeError("internal_state needs outputs of a forward that recorded a graph (run_forward(..., requires_grad=True) outside torch.no_grad())")
    saved = node.saved_tensors
    geom, binning, img = saved[-3], saved[-2], saved[-1]
    P = inp["means3D"].shape[0]; W, H = int(inp["W"]), int(inp["H"]); HW = W * H
    R = int(node.num_rendered) if hasattr(node, "num_rendered") else None
    gb = geom.cpu().numpy(); ib = img.cpu().numpy(); bb = binning.cpu().numpy()

    def view(buf, off, dtype, count):
        return np.frombuffer(buf.tobytes()[off:off + np.dtype(dtype).itemsize * count], dtype=dtype).copy()

    gx, gy = (W + 15) // 16, (H + 15) // 16
    st = {}
    go = lambda n: lib.ibgs_geom_offset(P, n.encode())
    io = lambda n: lib.ibgs_img_offset(W, H, n.encode())
    st["rec"] = view(gb, go("rec"), np.float32, P * 16).reshape(P, 16)
    st["depths"] = view(gb, go("depths"), np.float32, P)
    st["cov3D"] = view(gb, go("cov3D"), np.float32, P * 6).reshape(P, 6)
    st["tiles"] = view(gb, go("tiles"), np.uint32, P)
    st["clamped"] = view(gb, go("clamped"), np.uint8, P)
    st["offsets"] = view(gb, go("offsets"), np.uint32, P + 5)          # [P] = R, [P+1] sort error flag, [P+2] = C, [P+3] = Gaussians with tiles, [P+4] = 1: the order lies in the alternate buffer
    st["order"] = view(gb, go("order_alt" if st["offsets"][P + 4] == 1 else "order"), np.uint32, P)[:int(st["offsets"][P + 3])]
    st["ranges"] = view(ib, io("ranges"), np.uint32, gx * gy * 2).reshape(gx * gy, 2)
    st["final_T"] = view(ib, io("final_T"), np.float32, HW)
    st["n_contrib"] = view(ib, io("n_contrib"), np.uint32, HW)
    st["sum_w"] = view(ib, io("sum_w"), np.float32, HW)
    st["low_high"] = view(ib, io("low_high"), np.uint32, HW * 2).reshape(HW, 2)
    st["valid_idx"] = view(ib, io("valid_idx"), np.int32, HW * 5).reshape(5, HW)
    st["valid_w"] = view(ib, io("valid_w"), np.float32, HW * 5).reshape(5, HW)
    Rr = int(st["offsets"][P])
    st["R"] = Rr
    if Rr > 0:
        # the arena may have been carved for a rendered_hint >= R (include/ibgs_rast.h): recover that size from its length
        lo, hi = Rr, max(2 * Rr + (1 << 20), bb.size // 4 + 1)          # (the hint may come from a much denser scene of the same size: the capacity is bounded by the arena itself)
        while lo < hi:
            mid = (lo + hi) // 2
            if lib.ibgs_required_binning(mid, W, H) >= bb.size:
                hi = mid
            else:
                lo = mid + 1
        assert lib.ibgs_required_binning(lo, W, H) == bb.size, "binning arena size matches no capacity"
        st["binning_capacity"] = lo
        bo = lambda n: lib.ibgs_binning_offset(lo, W, H, n.encode())
        st["point_list"] = view(bb, bo("point_list"), np.uint32, Rr)
        # the two-level binning keeps no key array: the tile of every list entry follows from the ranges (which must tile [0, R))
        rg = st["ranges"].astype(np.int64)
        keys = np.full(Rr, 0xFFFFFFFF, np.uint32)
        for t in np.flatnonzero(rg[:, 1] > rg[:, 0]):
            keys[rg[t, 0]:rg[t, 1]] = t
        st["sorted_tile_keys"] = keys
    else:
        st["point_list"] = np.zeros(0, np.uint32); st["sorted_tile_keys"] = np.zeros(0, np.uint32)
    return st


def to_np(d):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in d.items()}
