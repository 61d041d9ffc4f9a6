"""ctypes binding of libibgs_rast.so (C ABI declared in include/ibgs_rast.h).

The library is the product path: if it is missing or fails to load this module raises -- there is
no Python / PyTorch / oracle fallback (a silent fallback would void every parity claim).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# IBGS_LIB: another build of the same ABI (A/B runs of two builds on ONE box: tools/ab_lib.sh); never a fallback -- a path that does not load raises
LIB_PATH = os.environ.get("IBGS_LIB") or os.path.join(_HERE, "libibgs_rast.so")

MAX_SRC = 5
MAX_BUFFER_LENGTH = 8
FLAG_DEBUG = 1
FLAG_TEX_QUANT = 2
FLAG_NO_TILE_CULL = 4
FLAG_CLEAR_GRAD_ACC = 8
FLAG_SH_FACTORED = 16
FLAG_TILE_WAVES = 32
FLAG_QUADRANT_WAVES = 64
FLAG_DETERMINISTIC = 128
FLAG_TEX_PACKED = 256
FLAG_NO_REF_POWER_SKIP = 512
FLAG_NO_ABS_GRAD = 1024
FLAG_REF_ARITH = 4096
FLAG_SRC_DEPTH_SLOTS = 8192
PLANE_NONE, PLANE_LEARNT, PLANE_SMALLEST_AXIS = 0, 1, 2
MAX_VIEWS = 8

c_float_p = ctypes.c_void_p  # raw device pointers travel as integers

ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)


class ForwardArgs(ctypes.Structure):
    _fields_ = [
        ("stream", ctypes.c_void_p),
        ("P", ctypes.c_int32), ("D", ctypes.c_int32), ("M", ctypes.c_int32),
        ("W", ctypes.c_int32), ("H", ctypes.c_int32),
        ("means3D", c_float_p), ("shs", c_float_p), ("colors_precomp", c_float_p), ("opacities", c_float_p),
        ("scales", c_float_p), ("rotations", c_float_p), ("cov3D_precomp", c_float_p), ("all_map", c_float_p),
        ("scale_modifier", ctypes.c_float),
        ("bg", c_float_p), ("viewmatrix", c_float_p), ("projmatrix", c_float_p), ("campos", c_float_p),
        ("tanfovx", ctypes.c_float), ("tanfovy", ctypes.c_float),
        ("n_src", ctypes.c_int32),
        ("ref_to_src", c_float_p), ("src_cam_pos", c_float_p), ("src_images", c_float_p), ("src_depths", c_float_p),
        ("buffer_length", ctypes.c_int32), ("depth_error_threshold", ctypes.c_float),
        ("prefiltered", ctypes.c_int32), ("render_geo", ctypes.c_int32), ("render_depth_only", ctypes.c_int32),
        ("flags", ctypes.c_uint32),
        ("geom", ctypes.c_void_p), ("geom_bytes", ctypes.c_size_t),
        ("img", ctypes.c_void_p), ("img_bytes", ctypes.c_size_t),
        ("binning_alloc", ALLOC_FN), ("binning_user", ctypes.c_void_p),
        ("tex", ctypes.c_void_p), ("tex_bytes", ctypes.c_size_t),
        ("out_color", c_float_p), ("radii", ctypes.c_void_p), ("out_normal", c_float_p), ("out_depth", c_float_p),
        ("out_cam_feat", c_float_p), ("out_warped", c_float_p), ("out_min_depth_diff", c_float_p),
        ("out_camera_ray", c_float_p), ("out_mask", ctypes.c_void_p),
        ("rendered_hint", ctypes.c_int64),
        ("plane_normal", c_float_p), ("plane_offset", c_float_p), ("plane_mode", ctypes.c_int32),
        ("n_views", ctypes.c_int32), ("view_tanfovx", ctypes.c_float * 8), ("view_tanfovy", ctypes.c_float * 8),
        ("tile_order_hint", ctypes.c_void_p),
        ("binning", ctypes.c_void_p), ("binning_bytes", ctypes.c_size_t),
        ("shs_rest", c_float_p),
        ("src_depth_slot", ctypes.c_int32 * 5),
    ]


class AdamTensor(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p), ("exp_avg_sq", ctypes.c_void_p),
                ("numel", ctypes.c_int64), ("lr", ctypes.c_double), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double),
                ("eps", ctypes.c_double), ("bias_correction1", ctypes.c_double), ("bias_correction2", ctypes.c_double)]


class CompactTensor(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p), ("append", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("width", ctypes.c_int32), ("reserved", ctypes.c_int32)]


COMPACT_MAX_TENSORS = 40


class BackwardArgs(ctypes.Structure):
    _fields_ = [
        ("stream", ctypes.c_void_p),
        ("P", ctypes.c_int32), ("D", ctypes.c_int32), ("M", ctypes.c_int32),
        ("W", ctypes.c_int32), ("H", ctypes.c_int32),
        ("R", ctypes.c_int64),
        ("means3D", c_float_p), ("shs", c_float_p), ("colors_precomp", c_float_p),
        ("scales", c_float_p), ("rotations", c_float_p), ("cov3D_precomp", c_float_p), ("all_map", c_float_p),
        ("scale_modifier", ctypes.c_float),
        ("bg", c_float_p), ("viewmatrix", c_float_p), ("projmatrix", c_float_p), ("campos", c_float_p),
        ("tanfovx", ctypes.c_float), ("tanfovy", ctypes.c_float),
        ("n_src", ctypes.c_int32),
        ("ref_to_src", c_float_p), ("src_cam_pos", c_float_p), ("src_images", c_float_p), ("src_depths", c_float_p),
        ("radii", ctypes.c_void_p),
        ("out_depth", c_float_p), ("out_warped", c_float_p),
        ("geom", ctypes.c_void_p), ("binning", ctypes.c_void_p), ("img", ctypes.c_void_p),
        ("tex", ctypes.c_void_p), ("tex_bytes", ctypes.c_size_t),
        ("dL_dcolor", c_float_p), ("dL_dnormal", c_float_p), ("dL_ddepth", c_float_p), ("dL_dwarped", c_float_p),
        ("grad_acc", c_float_p),
        ("dL_dmean2D", c_float_p), ("dL_dmean2D_abs", c_float_p), ("dL_dconic", c_float_p), ("dL_dopacity", c_float_p),
        ("dL_dcolors", c_float_p), ("dL_dmean3D", c_float_p), ("dL_dcov3D", c_float_p), ("dL_dsh", c_float_p),
        ("dL_dscale", c_float_p), ("dL_drot", c_float_p), ("dL_dall_map", c_float_p),
        ("render_geo", ctypes.c_int32), ("flags", ctypes.c_uint32),
        ("plane_normal", c_float_p), ("plane_offset", c_float_p), ("plane_mode", ctypes.c_int32),
        ("dL_dplane_normal", c_float_p), ("dL_dplane_offset", c_float_p),
        ("geo_table", ctypes.c_void_p), ("geo_table_bytes", ctypes.c_size_t),
        ("det_scratch", ctypes.c_void_p), ("det_scratch_bytes", ctypes.c_size_t),
        ("buffer_length", ctypes.c_int32),
        ("tile_order_out", ctypes.c_void_p),
        ("shs_rest", c_float_p), ("dL_dsh_rest", c_float_p),
    ]


# every symbol include/ibgs_rast.h declares
EXPORTS = ["ibgs_required_geom", "ibgs_required_img", "ibgs_required_binning", "ibgs_required_tex",
           "ibgs_forward", "ibgs_backward", "ibgs_mark_visible", "ibgs_tile_order_slots",
           "ibgs_geom_offset", "ibgs_img_offset", "ibgs_binning_offset",
           "ibgs_sizeof_forward_args", "ibgs_sizeof_backward_args", "ibgs_timing_enable", "ibgs_timing_collect",
           "ibgs_required_knn", "ibgs_knn_mean_dist2", "ibgs_sh_grad_from_views", "ibgs_adam_step", "ibgs_adam_step_sh",
           "ibgs_required_compact", "ibgs_compact_plan", "ibgs_compact_apply", "ibgs_densify_stats", "ibgs_required_deterministic", "ibgs_required_geo_table", "ibgs_required_deterministic_for", "ibgs_required_geo_table_for", "ibgs_last_forward_stats", "ibgs_check_async",
           "ibgs_required_l1", "ibgs_l1_loss", "ibgs_l1_grad", "ibgs_l1_rescale",
           "ibgs_depth_normal_forward", "ibgs_depth_normal_backward", "ibgs_activate_forward", "ibgs_activate_backward",
           "ibgs_last_error", "ibgs_version"]

_lib = None


class RasterizerLibraryError(RuntimeError):
    pass


def load():
    """Load libibgs_rast.so (after torch, so that the HIP runtime torch ships is the one in use)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RasterizerLibraryError(
            "libibgs_rast.so is not built (%s). Run `python -m ibgs_amd._build` or __graft_entry__.build(); "
            "there is no fallback path." % LIB_PATH)
    import torch  # noqa: F401  (loads libamdhip64 first; both sides then share one HIP runtime)
    lib = ctypes.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise RasterizerLibraryError("libibgs_rast.so lacks symbol %s" % name)
    lib.ibgs_required_geom.restype = ctypes.c_size_t
    lib.ibgs_required_geom.argtypes = [ctypes.c_int32]
    lib.ibgs_required_img.restype = ctypes.c_size_t
    lib.ibgs_required_img.argtypes = [ctypes.c_int32, ctypes.c_int32]
    lib.ibgs_required_binning.restype = ctypes.c_size_t
    lib.ibgs_required_binning.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
    lib.ibgs_required_tex.restype = ctypes.c_size_t
    lib.ibgs_required_tex.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    lib.ibgs_forward.restype = ctypes.c_int64
    lib.ibgs_forward.argtypes = [ctypes.POINTER(ForwardArgs)]
    lib.ibgs_backward.restype = ctypes.c_int32
    lib.ibgs_backward.argtypes = [ctypes.POINTER(BackwardArgs)]
    lib.ibgs_mark_visible.restype = ctypes.c_int32
    lib.ibgs_mark_visible.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p]
    for f in (lib.ibgs_geom_offset,):
        f.restype = ctypes.c_int64
        f.argtypes = [ctypes.c_int32, ctypes.c_char_p]
    lib.ibgs_img_offset.restype = ctypes.c_int64
    lib.ibgs_img_offset.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_char_p]
    lib.ibgs_binning_offset.restype = ctypes.c_int64
    lib.ibgs_binning_offset.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_char_p]
    lib.ibgs_last_error.restype = ctypes.c_char_p
    lib.ibgs_version.restype = ctypes.c_char_p
    lib.ibgs_timing_enable.restype = None
    lib.ibgs_timing_enable.argtypes = [ctypes.c_uint32]
    lib.ibgs_timing_collect.restype = ctypes.c_int32
    lib.ibgs_timing_collect.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.ibgs_required_knn.restype = ctypes.c_size_t
    lib.ibgs_required_knn.argtypes = [ctypes.c_int32]
    lib.ibgs_sh_grad_from_views.restype = ctypes.c_int32
    lib.ibgs_sh_grad_from_views.argtypes = [ctypes.c_void_p] + [ctypes.c_int32] * 4 + [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_void_p]
    lib.ibgs_adam_step.restype = ctypes.c_int32
    lib.ibgs_adam_step.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    lib.ibgs_adam_step_sh.restype = ctypes.c_int32
    lib.ibgs_adam_step_sh.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.ibgs_tile_order_slots.restype = ctypes.c_size_t
    lib.ibgs_tile_order_slots.argtypes = [ctypes.c_int32, ctypes.c_int32]
    lib.ibgs_required_l1.restype = ctypes.c_size_t
    lib.ibgs_required_l1.argtypes = []
    lib.ibgs_l1_loss.restype = ctypes.c_int32
    lib.ibgs_l1_loss.argtypes = [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 5 + [ctypes.c_size_t]
    lib.ibgs_l1_grad.restype = ctypes.c_int32
    lib.ibgs_l1_grad.argtypes = [ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 4
    lib.ibgs_densify_stats.restype = ctypes.c_int32
    lib.ibgs_densify_stats.argtypes = [ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_void_p] * 8
    lib.ibgs_depth_normal_forward.restype = ctypes.c_int32
    lib.ibgs_depth_normal_forward.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32] + [ctypes.c_float] * 4 + [ctypes.c_void_p] * 2
    lib.ibgs_depth_normal_backward.restype = ctypes.c_int32
    lib.ibgs_depth_normal_backward.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32] + [ctypes.c_float] * 4 + [ctypes.c_void_p] * 3
    lib.ibgs_activate_forward.restype = ctypes.c_int32
    lib.ibgs_activate_forward.argtypes = [ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_void_p] * 6
    lib.ibgs_activate_backward.restype = ctypes.c_int32
    lib.ibgs_activate_backward.argtypes = [ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_void_p] * 9
    lib.ibgs_l1_rescale.restype = ctypes.c_int32
    lib.ibgs_l1_rescale.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    lib.ibgs_required_geo_table.restype = ctypes.c_size_t
    lib.ibgs_required_geo_table.argtypes = [ctypes.c_int32, ctypes.c_int32]
    lib.ibgs_last_forward_stats.restype = None
    lib.ibgs_last_forward_stats.argtypes = [ctypes.POINTER(ctypes.c_int64)]
    lib.ibgs_check_async.restype = ctypes.c_int32
    lib.ibgs_check_async.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    lib.ibgs_required_geo_table_for.restype = ctypes.c_size_t
    lib.ibgs_required_geo_table_for.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    lib.ibgs_required_deterministic_for.restype = ctypes.c_size_t
    lib.ibgs_required_deterministic_for.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_uint32]
    lib.ibgs_required_deterministic.restype = ctypes.c_size_t
    lib.ibgs_required_deterministic.argtypes = [ctypes.c_int64, ctypes.c_int32]
    lib.ibgs_required_compact.restype = ctypes.c_size_t
    lib.ibgs_required_compact.argtypes = [ctypes.c_int32]
    lib.ibgs_compact_plan.restype = ctypes.c_int64
    lib.ibgs_compact_plan.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.ibgs_compact_apply.restype = ctypes.c_int32
    lib.ibgs_compact_apply.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    lib.ibgs_knn_mean_dist2.restype = ctypes.c_int32
    lib.ibgs_knn_mean_dist2.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.ibgs_sizeof_forward_args.restype = ctypes.c_size_t
    lib.ibgs_sizeof_backward_args.restype = ctypes.c_size_t
    if (lib.ibgs_sizeof_forward_args() != ctypes.sizeof(ForwardArgs)
            or lib.ibgs_sizeof_backward_args() != ctypes.sizeof(BackwardArgs)):
        raise RasterizerLibraryError("ctypes struct layout does not match libibgs_rast.so (stale build?)")
    _lib = lib
    return lib


# "binning" = coarse entries placed per cell + per-tile counts + ranges, "list_scatter" = the ids to their slots (two-level binning,
# csrc/binning.hip); "ranges" has no kernel of its own any more (always 0)
STAGES = ["preprocess", "depth_sort", "scan", "binning", "list_scatter", "ranges", "render_fwd", "render_bwd",
          "preprocess_bwd", "geo_window", "tile_order"]


def timing_enable(stages):
    """Bracket the named stages with hipEvents on the op's stream (bench.py roofline)."""
    mask = 0
    for s in stages:
        mask |= 1 << STAGES.index(s)
    load().ibgs_timing_enable(mask)


def timing_collect():
    """-> {stage: (total_ms, launches)} since the previous collect."""
    ms = (ctypes.c_float * len(STAGES))()
    n = (ctypes.c_int32 * len(STAGES))()
    rc = load().ibgs_timing_collect(ms, n)
    if rc < 0:
        raise RuntimeError("ibgs_timing_collect failed: %s" % last_error())
    return {STAGES[i]: (float(ms[i]), int(n[i])) for i in range(len(STAGES))}


def last_error():
    return load().ibgs_last_error().decode("utf-8", "replace")


def last_forward_stats():
    """(R, coarse binning entries or -1, hint missed?) of this thread's last ibgs_forward."""
    out = (ctypes.c_int64 * 3)()
    load().ibgs_last_forward_stats(out)
    return int(out[0]), int(out[1]), bool(out[2])
