"""CPU oracle for the IBGS plane rasterizer -- TEST INFRASTRUCTURE ONLY.

Thin numpy/ctypes front-end over ``ibgs_oracle.c`` (the C restatement of the reference's
CUDA rasterizer; every C function cites the reference file:line it follows).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package; the product path (``ibgs_amd``) never does.

PARITY STATUS: "parity unpinned" -- see the header of ``ibgs_oracle.c`` and DESIGN.md.

The front-end mirrors the stage order of ``CudaRasterizer::Rasterizer::forward/backward``
(cuda_rasterizer/rasterizer_impl.cu:320-515, 519-666) and the tensor allocation of
``RasterizeGaussiansCUDA`` / ``RasterizeGaussiansBackwardCUDA`` (rasterize_points.cu:37-271).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libibgs_oracle.so")
_SO_FMA = os.path.join(_HERE, "_build", "libibgs_oracle_fma.so")
_SO_F64 = os.path.join(_HERE, "_build", "libibgs_oracle_f64.so")
_SO_ACC32 = os.path.join(_HERE, "_build", "libibgs_oracle_acc32.so")
_SO_LFORM = os.path.join(_HERE, "_build", "libibgs_oracle_lform.so")
_RT = np.float32          # element type of the float arrays at the C interface: float32, float64 inside `variant("f64")`
_SRC = os.path.join(_HERE, "ibgs_oracle.c")
_lib = None
_libs = {}

MAX_SRC = 5


def build(force=False):
    """Compile the C oracle with gcc (a few seconds).  Two builds of the same source: the oracle proper (no fused multiply-adds, IEEE
    operation by operation) and a variant in which gcc contracts a*b+c into fma wherever it likes (`variant("fma")`).  nvcc contracts too
    (-fmad=true is its default) in a pattern that cannot be known here, so the difference between the two builds is the size of what
    the reference's own arithmetic leaves undetermined -- tests use it as the noise floor of ill-conditioned quantities.
    A third build, `variant("f64")`, turns every float of the source into a double (-DORC_F64): the arbiter for ill-conditioned quantities --
    a float build is as good as its distance from this one.  A fourth, `variant("acc32")` (-DORC_ACC_FLOAT), is the oracle proper with its per-Gaussian gradient
    sums kept in float like the reference's atomicAdd (single-threaded, pixel order): a diagnostic for what float accumulation alone costs."""
    os.makedirs(os.path.dirname(_SO), exist_ok=True)
    for so, flags in ((_SO, ["-ffp-contract=off"]), (_SO_FMA, ["-ffp-contract=fast", "-mfma"]), (_SO_F64, ["-ffp-contract=off", "-DORC_F64"]),
                      (_SO_ACC32, ["-ffp-contract=off", "-DORC_ACC_FLOAT"]), (_SO_LFORM, ["-ffp-contract=off", "-DORC_ACC_FLOAT", "-DORC_LFORM"])):
        if (not force) and os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(_SRC):
            continue
        subprocess.check_call(["gcc", "-O2"] + flags + ["-fno-fast-math", "-fopenmp", "-shared", "-fPIC", "-o", so, _SRC, "-lm"])
    return _SO


def _load(path):
    if path not in _libs:
        build()
        L = ctypes.CDLL(path)
        L.orc_bin_count.restype = ctypes.c_int64
        L.orc_higher_msb.restype = ctypes.c_uint32
        _libs[path] = L
    return _libs[path]


def lib():
    global _lib
    if _lib is None:
        _lib = _load(_SO)
    return _lib


class variant:
    """`with oracle.variant("fma"):` runs the calls inside on the fma-contracted build of the same C source, `variant("f64")` on the
    build in which every float is a double (inputs are converted, results come back as float64 arrays) -- see build()."""

    def __init__(self, name):
        assert name in ("fma", "plain", "f64", "acc32", "lform")
        self.path = {"fma": _SO_FMA, "plain": _SO, "f64": _SO_F64, "acc32": _SO_ACC32, "lform": _SO_LFORM}[name]
        self.rt = np.float64 if name == "f64" else np.float32

    def __enter__(self):
        global _lib, _RT
        self.prev = (_lib, _RT)
        _lib = _load(self.path)
        _RT = self.rt
        return self

    def __exit__(self, *exc):
        global _lib, _RT
        _lib, _RT = self.prev
        return False


def _p(a):
    """numpy array (or None) -> void*"""
    if a is None:
        return ctypes.c_void_p(0)
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be contiguous"
    return ctypes.c_void_p(a.ctypes.data)


def _f32(a):
    """contiguous array of the build's float type (float32; float64 inside variant("f64")), or None"""
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=_RT))
    return a if a.size else None


def _cf(x):
    return ctypes.c_double(float(x)) if _RT is np.float64 else ctypes.c_float(float(x))


def _ci(x):
    return ctypes.c_int(int(x))


def tile_grid(W, H):
    return (W + 15) // 16, (H + 15) // 16


def mark_visible(means3D, viewmatrix):
    means3D = _f32(means3D)
    P = means3D.shape[0]
    out = np.zeros(P, dtype=np.uint8)
    lib().orc_mark_visible(_ci(P), _p(means3D), _p(_f32(viewmatrix).reshape(-1)), _p(out))
    return out.astype(bool)


def forward(inp, tex_quant=False, cull=False):
    """Run the full forward.  ``inp`` is a dict of numpy arrays / scalars:

    means3D (P,3), opacities (P,) or (P,1), one of {shs (P,M,3) | colors_precomp (P,3)},
    one of {scales (P,3) + rotations (P,4) | cov3D_precomp (P,6)}, all_map (P,5) or None,
    bg (3,), viewmatrix (16,) / (4,4) [transposed convention], projmatrix, campos (3,),
    W, H, tanfovx, tanfovy, sh_degree, scale_modifier,
    n_src, ref_to_src (n,16), src_cam_pos (n,3), src_images (n,3,H,W), src_depths (n,1,H,W),
    buffer_length, depth_thr, render_geo, render_depth_only.

    Returns a dict with the 9 public outputs plus every internal state array.
    """
    L = lib()
    means3D = np.ascontiguousarray(np.asarray(inp["means3D"], dtype=_RT)).reshape(-1, 3); P = means3D.shape[0]
    W, H = int(inp["W"]), int(inp["H"]); HW = W * H
    gx, gy = tile_grid(W, H)
    shs = _f32(inp.get("shs")); colors_precomp = _f32(inp.get("colors_precomp"))
    scales = _f32(inp.get("scales")); rotations = _f32(inp.get("rotations"))
    cov3D_precomp = _f32(inp.get("cov3D_precomp")); all_map = _f32(inp.get("all_map"))
    opac = np.ascontiguousarray(np.asarray(inp["opacities"], dtype=_RT)).reshape(-1)
    vm = _f32(inp["viewmatrix"]).reshape(-1); pm = _f32(inp["projmatrix"]).reshape(-1)
    campos = _f32(inp["campos"]).reshape(-1); bg = _f32(inp["bg"]).reshape(-1)
    D = int(inp.get("sh_degree", 0)); M = 0 if shs is None else int(shs.shape[1])
    mod = float(inp.get("scale_modifier", 1.0))
    tanx, tany = float(inp["tanfovx"]), float(inp["tanfovy"])
    render_geo = bool(inp.get("render_geo", False)); depth_only = bool(inp.get("render_depth_only", False))
    n_src = int(inp.get("n_src", 1)); Lbuf = int(inp.get("buffer_length", 4)); thr = float(inp.get("depth_thr", 0.01))
    ref_to_src = _f32(inp.get("ref_to_src")); src_cam_pos = _f32(inp.get("src_cam_pos"))
    src_images = _f32(inp.get("src_images")); src_depths = _f32(inp.get("src_depths"))
    if ref_to_src is None: ref_to_src = np.zeros((n_src, 16), _RT)
    if src_cam_pos is None: src_cam_pos = np.zeros((n_src, 3), _RT)
    if src_images is None: src_images = np.zeros((n_src, 3, H, W), _RT)
    if src_depths is None: src_depths = np.zeros((n_src, 1, H, W), _RT)

    st = {}
    st["radii"] = np.zeros(P, np.int32); st["means2D"] = np.zeros((P, 2), _RT)
    st["depths"] = np.zeros(P, _RT); st["cov3D"] = np.zeros((P, 6), _RT)
    st["rgb"] = np.zeros((P, 3), _RT); st["conic_opacity"] = np.zeros((P, 4), _RT)
    st["tiles_touched"] = np.zeros(P, np.uint32); st["clamped"] = np.zeros((P, 3), np.uint8)
    st["rect4"] = np.zeros((P, 4), np.int32); st["tmask"] = np.zeros((P, 4), np.uint64)      # CULL_WORDS mask words per Gaussian
    out = {
        "color": np.zeros((3, H, W), _RT), "normal_map": np.zeros((3, H, W), _RT),
        "median_depth": np.zeros((1, H, W), _RT), "cam_feat": np.zeros((4 * MAX_SRC, H, W), _RT),
        "warped_image": np.zeros((3 * MAX_SRC, H, W), _RT), "min_depth_diff": np.zeros((1, H, W), _RT),
        "camera_ray": np.zeros((3, H, W), _RT), "use_first_src_frame_mask": np.zeros((1, H, W), np.int32),
    }
    st["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    st["final_T"] = np.zeros(HW, _RT); st["n_contrib"] = np.zeros(HW, np.uint32)
    st["cache_sum_w"] = np.zeros(HW, _RT); st["cache_low"] = np.zeros(HW, np.uint32)
    st["cache_high"] = np.zeros(HW, np.uint32)
    st["valid_src_idx"] = np.full((MAX_SRC, HW), -1, np.int32); st["valid_src_w"] = np.zeros((MAX_SRC, HW), _RT)
    if P == 0:   # rasterize_points.cu:101-102
        st["point_list"] = np.zeros(0, np.uint32); st["keys"] = np.zeros(0, np.uint64)
        out.update(st); out["num_rendered"] = 0
        return out

    L.orc_preprocess(_ci(P), _ci(D), _ci(M), _p(means3D), _p(scales), _cf(mod), _p(rotations), _p(opac), _p(shs),
                     _p(cov3D_precomp), _p(colors_precomp), _p(vm), _p(pm), _p(campos), _ci(W), _ci(H),
                     _cf(tanx), _cf(tany), _ci(depth_only), _ci(bool(cull)),
                     _p(st["radii"]), _p(st["means2D"]), _p(st["depths"]), _p(st["cov3D"]), _p(st["rgb"]),
                     _p(st["conic_opacity"]), _p(st["tiles_touched"]), _p(st["clamped"]), _p(st["rect4"]), _p(st["tmask"]))
    R = int(L.orc_bin_count(_ci(P), _p(st["tiles_touched"])))
    st["keys"] = np.zeros(R, np.uint64); st["point_list"] = np.zeros(R, np.uint32)
    rc = L.orc_bin(_ci(P), ctypes.c_int64(R), _p(st["radii"]), _p(st["rect4"]), _p(st["tmask"]), _p(st["depths"]),
                   _p(st["means2D"]), _p(st["conic_opacity"]), _ci(W), _ci(H),
                   _p(st["keys"]), _p(st["point_list"]), _p(st["ranges"]))
    assert rc == 0, "oracle binning failed (%d)" % rc
    feats = colors_precomp if colors_precomp is not None else st["rgb"]
    L.orc_render_forward(_ci(W), _ci(H), _p(st["ranges"]), _p(st["point_list"]), _p(st["means2D"]), _p(feats),
                         _p(all_map), _p(st["conic_opacity"]), _p(vm), _p(campos), _p(bg), _cf(tanx), _cf(tany),
                         _ci(n_src), _p(ref_to_src.reshape(-1)), _p(src_cam_pos.reshape(-1)),
                         _p(src_images.reshape(-1)), _p(src_depths.reshape(-1)), _ci(Lbuf), _cf(thr),
                         _ci(render_geo), _ci(depth_only), _ci(bool(tex_quant)),
                         _p(st["final_T"]), _p(st["n_contrib"]), _p(st["cache_sum_w"]), _p(st["cache_low"]),
                         _p(st["cache_high"]), _p(st["valid_src_idx"]), _p(st["valid_src_w"]),
                         _p(out["color"]), _p(out["normal_map"]), _p(out["median_depth"]), _p(out["cam_feat"]),
                         _p(out["warped_image"]), _p(out["min_depth_diff"]), _p(out["camera_ray"]),
                         _p(out["use_first_src_frame_mask"]))
    out.update(st)
    out["num_rendered"] = R
    return out


def backward(inp, fwd, dL_dcolor, dL_dnormal=None, dL_ddepth=None, dL_dwarped=None, tex_quant=False):
    """Full backward given ``forward``'s result.  Returns the reference's 10 gradients
    (rasterize_points.cu:209-219, 270) plus dL_dconic."""
    L = lib()
    means3D = np.ascontiguousarray(np.asarray(inp["means3D"], dtype=_RT)).reshape(-1, 3); P = means3D.shape[0]
    W, H = int(inp["W"]), int(inp["H"])
    shs = _f32(inp.get("shs")); colors_precomp = _f32(inp.get("colors_precomp"))
    scales = _f32(inp.get("scales")); rotations = _f32(inp.get("rotations"))
    cov3D_precomp = _f32(inp.get("cov3D_precomp")); all_map = _f32(inp.get("all_map"))
    vm = _f32(inp["viewmatrix"]).reshape(-1); pm = _f32(inp["projmatrix"]).reshape(-1)
    campos = _f32(inp["campos"]).reshape(-1); bg = _f32(inp["bg"]).reshape(-1)
    D = int(inp.get("sh_degree", 0)); M = 0 if shs is None else int(shs.shape[1])
    mod = float(inp.get("scale_modifier", 1.0))
    tanx, tany = float(inp["tanfovx"]), float(inp["tanfovy"])
    render_geo = bool(inp.get("render_geo", False))
    n_src = int(inp.get("n_src", 1))
    ref_to_src = _f32(inp.get("ref_to_src")); src_images = _f32(inp.get("src_images"))
    if ref_to_src is None: ref_to_src = np.zeros((n_src, 16), _RT)
    if src_images is None: src_images = np.zeros((n_src, 3, H, W), _RT)
    g_c = _f32(dL_dcolor)
    g_n = _f32(dL_dnormal) if dL_dnormal is not None else np.zeros((3, H, W), _RT)
    g_d = _f32(dL_ddepth) if dL_ddepth is not None else np.zeros((1, H, W), _RT)
    g_w = _f32(dL_dwarped) if dL_dwarped is not None else np.zeros((3 * MAX_SRC, H, W), _RT)
    if g_n is None: g_n = np.zeros((3, H, W), _RT)
    if g_d is None: g_d = np.zeros((1, H, W), _RT)
    if g_w is None: g_w = np.zeros((3 * MAX_SRC, H, W), _RT)

    res = {
        "dL_dmeans3D": np.zeros((P, 3), _RT), "dL_dmeans2D": np.zeros((P, 3), _RT),
        "dL_dmeans2D_abs": np.zeros((P, 3), _RT), "dL_dcolors": np.zeros((P, 3), _RT),
        "dL_dall_map": np.zeros((P, 5), _RT), "dL_dconic": np.zeros((P, 4), _RT),
        "dL_dopacity": np.zeros((P, 1), _RT), "dL_dcov3D": np.zeros((P, 6), _RT),
        "dL_dsh": np.zeros((P, M, 3), _RT), "dL_dscales": np.zeros((P, 3), _RT),
        "dL_drotations": np.zeros((P, 4), _RT),
    }
    if P == 0:
        return res
    acc_m = np.zeros((P, 2), np.float64); acc_ma = np.zeros((P, 2), np.float64)
    acc_c = np.zeros((P, 3), np.float64); acc_o = np.zeros(P, np.float64)
    acc_col = np.zeros((P, 3), np.float64); acc_am = np.zeros((P, 5), np.float64)
    feats = colors_precomp if colors_precomp is not None else fwd["rgb"]
    L.orc_render_backward(_ci(W), _ci(H), _p(fwd["ranges"]), _p(fwd["point_list"]), _p(fwd["means2D"]),
                          _p(fwd["conic_opacity"]), _p(feats), _p(all_map), _p(bg), _cf(tanx), _cf(tany),
                          _ci(n_src), _p(ref_to_src.reshape(-1)), _p(src_images.reshape(-1)),
                          _p(fwd["median_depth"]), _p(fwd["warped_image"]),
                          _p(fwd["final_T"]), _p(fwd["n_contrib"]), _p(fwd["cache_sum_w"]), _p(fwd["cache_low"]),
                          _p(fwd["cache_high"]), _p(fwd["valid_src_idx"]), _p(fwd["valid_src_w"]),
                          _p(g_c), _p(g_n), _p(g_d), _p(g_w), _ci(render_geo), _ci(bool(tex_quant)),
                          _p(acc_m), _p(acc_ma), _p(acc_c), _p(acc_o), _p(acc_col), _p(acc_am))
    res["dL_dmeans2D"][:, :2] = acc_m; res["dL_dmeans2D_abs"][:, :2] = acc_ma
    res["dL_dconic"][:, 0] = acc_c[:, 0]; res["dL_dconic"][:, 1] = acc_c[:, 1]; res["dL_dconic"][:, 3] = acc_c[:, 2]
    res["dL_dopacity"][:, 0] = acc_o; res["dL_dcolors"][:] = acc_col; res["dL_dall_map"][:] = acc_am
    cov = cov3D_precomp if cov3D_precomp is not None else fwd["cov3D"]
    L.orc_preprocess_backward(_ci(P), _ci(D), _ci(M), _p(means3D), _p(fwd["radii"]), _p(shs), _p(fwd["clamped"]),
                              _p(scales), _p(rotations), _cf(mod), _p(cov), _p(vm), _p(pm), _p(campos),
                              _ci(W), _ci(H), _cf(tanx), _cf(tany),
                              _p(res["dL_dmeans2D"]), _p(res["dL_dconic"]), _p(res["dL_dcolors"]),
                              _p(res["dL_dmeans3D"]), _p(res["dL_dcov3D"]), _p(res["dL_dsh"]),
                              _p(res["dL_dscales"]), _p(res["dL_drotations"]))
    return res


def power_skips():
    """(forward, backward) counts of (pixel, Gaussian) pairs that the LAST forward / backward call dropped through the
    reference's `power > 0` test (forward.cu:420, backward.cu:645) -- it can only fire through rounding, on needle-shaped conics."""
    out = (ctypes.c_longlong * 2)()
    lib().orc_power_skips(out)
    return int(out[0]), int(out[1])


def eval_sh(deg, shs, dirs):
    """shs (N,M,3), dirs (N,3) unit -> (N,3) SH colour before the +0.5 / clamp."""
    shs = _f32(shs); dirs = _f32(dirs)
    N, M = shs.shape[0], shs.shape[1]
    out = np.zeros((N, 3), _RT)
    lib().orc_eval_sh(_ci(N), _ci(deg), _ci(M), _p(dirs), _p(shs), _p(out))
    return out


def knn_mean_dist2(points):
    """(P,3) -> (P,) mean squared distance to the 3 nearest other points (brute force)."""
    pts = np.ascontiguousarray(np.asarray(points, dtype=_RT)).reshape(-1, 3)
    out = np.zeros(pts.shape[0], _RT)
    lib().orc_knn_mean_dist2(_ci(pts.shape[0]), _p(pts), _p(out))
    return out
