"""Render glue: this repository's counterpart of the reference's ``gaussian_renderer/__init__.py``.

``render()``, ``render_depth()`` and ``render_normal()`` keep the reference's signatures, argument
meaning and returned dictionary keys (reference gaussian_renderer/__init__.py:143-147, 349-363, 41-43,
16-26) so that a ``train.py`` / ``render.py`` written against the reference can call them unchanged.
What they do around the rasterizer:

  (i)   zero ``viewspace_points`` / ``viewspace_points_abs`` sinks whose ``.grad`` receive dL/dmean2D
        (reference :153-159),
  (ii)  tan(FoV/2), camera matrices, SH or pre-computed colours, scale+rotation or 3D covariance,
  (iii) source-frame selection, ``ref_to_src`` = W2C_src @ C2W_ref, source camera centres (:228-267),
        optional fresh depth-only renders of the sources (:245-253),
  (iv)  the per-Gaussian plane map ``[n_cam, 1, |d_cam|]`` (:304-316),
  (v)   depth -> normal by finite differences (:338-342; utils/graphics_utils.py:25-83),
  (vi)  the appearance affine ``exp(a) * img + b`` (:344-347).

Unlike the reference nothing here hard-codes ``device="cuda"``: tensors follow ``pc.get_xyz.device``
(one process per GPU in view-parallel mode must not assume device 0).
"""
import math
import random
import weakref
from typing import Optional

import numpy as np
import torch

from .activations import fused_activations, standard_model
from . import rasterizer as _rz
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer

_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435)


def eval_sh(deg, sh, dirs):
    """SH -> colour (before the +0.5 / clamp), sh: (..., 3, K), dirs: (..., 3) unit vectors.
    Same polynomial as the kernels (reference utils/sh_utils.py:57-112, degrees 0..3)."""
    res = _C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        res = res - _C1 * y * sh[..., 1] + _C1 * z * sh[..., 2] - _C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + _C2[0] * xy * sh[..., 4] + _C2[1] * yz * sh[..., 5] + _C2[2] * (2.0 * zz - xx - yy) * sh[..., 6]
                   + _C2[3] * xz * sh[..., 7] + _C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                res = (res + _C3[0] * y * (3 * xx - yy) * sh[..., 9] + _C3[1] * xy * z * sh[..., 10]
                       + _C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + _C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                       + _C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + _C3[5] * z * (xx - yy) * sh[..., 14]
                       + _C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return res


def normal_from_depth_image(depth, intrinsic, extrinsic=None):
    """(H,W) depth -> (H,W,3) normals from the cross product of central differences of the
    back-projected points, border = 0 (reference utils/graphics_utils.py:25-83, offset=None branch)."""
    H, W = depth.shape
    dev = depth.device
    xs = torch.arange(W, dtype=torch.float32, device=dev) / (W - 1)
    ys = torch.arange(H, dtype=torch.float32, device=dev) / (H - 1)
    gx, gy = torch.meshgrid(xs, ys, indexing="xy")
    ndc = torch.stack([gx, gy], dim=-1) * torch.tensor([[W - 1, H - 1]], dtype=torch.float32, device=dev)
    cam = torch.cat([ndc * depth[..., None], depth[..., None]], dim=-1)
    xyz = cam @ torch.inverse(intrinsic.to(dev).t())
    bottom = xyz[2:H, 1:W - 1]; top = xyz[0:H - 2, 1:W - 1]
    right = xyz[1:H - 1, 2:W]; left = xyz[1:H - 1, 0:W - 2]
    n = torch.cross(right - left, top - bottom, dim=-1)
    n = torch.nn.functional.normalize(n, p=2, dim=-1)
    n = torch.nn.functional.pad(n.permute(2, 0, 1), (1, 1, 1, 1), mode="constant").permute(1, 2, 0)
    return n


# render(return_depth_normal=True): the depth -> normal map and its normalisation as one HIP kernel each way (ibgs_amd/depthnormal.py).  False = the
# reference's torch formulation below, kept as the behavioural definition (tests compare the two).
DEPTH_TABLE = True          # render(): hand the rasterizer the depth cache itself + the sources' plane numbers instead of a stack of their planes (round 6)
FUSED_DEPTH_NORMAL = True


def render_normal(viewpoint_cam, depth, offset=None, normal=None, scale=1):
    intrinsic_matrix, extrinsic_matrix = viewpoint_cam.get_calib_matrix_nerf(scale=scale)
    st = max(int(scale / 2) - 1, 0)
    normal_ref = normal_from_depth_image(depth[st::scale, st::scale], intrinsic_matrix.to(depth.device),
                                         extrinsic_matrix.to(depth.device))
    return normal_ref.permute(2, 0, 1)


# build [n_cam, 1, |d_cam|] inside the preprocess kernels (and its backward inside
# preprocess_bwd) instead of the ~10 torch kernels + (P, 5) round trip of `_plane_map` below.  False = the
# reference's glue, kept as the behavioural definition (tests compare the two).
FUSED_PLANE_MAP = True


def _plane_inputs(pc, viewpoint_camera, learnt_normal, means3D, scales, rotations):
    """kwargs for GaussianRasterizer.forward: either the fused raw plane parameters or the reference's all_map."""
    if FUSED_PLANE_MAP:      # means3D is pc.get_xyz in both callers, as in the reference (the flip test uses pc._xyz)
        if learnt_normal and hasattr(pc, "_normal"):
            return dict(plane_normal=pc._normal, plane_offset=getattr(pc, "_offset", None), plane_mode=1)
        if not learnt_normal and scales is not None and rotations is not None:
            return dict(plane_mode=2)
    return dict(all_map=_plane_map(pc, viewpoint_camera, learnt_normal, means3D))


def _plane_map(pc, viewpoint_camera, learnt_normal, means3D):
    """all_map (P,5) = [normal in camera frame, 1, |plane offset in camera frame|]."""
    V = viewpoint_camera.world_view_transform.to(means3D.device)
    if learnt_normal:
        global_normal, offset_global = pc.get_normal(viewpoint_camera)
    else:
        global_normal = pc.get_normal_w_smallest_axis(viewpoint_camera)
    local_normal = global_normal @ V[:3, :3]
    global_distance = -(global_normal * means3D).sum(-1)
    if learnt_normal:
        global_distance = global_distance + offset_global.squeeze()
    local_distance = (global_distance - torch.sum(local_normal * V[[3], :3], dim=1)).abs()
    all_map = torch.zeros((means3D.shape[0], 5), device=means3D.device, dtype=torch.float32)
    all_map[:, :3] = local_normal
    all_map[:, 3] = 1.0
    all_map[:, 4] = local_distance
    return all_map


# SH coefficients handed to the rasterizer as the model's two arrays (`_features_dc`, `_features_rest`) instead of `pc.get_features` = their torch.cat
# (scene/gaussian_model.py:140-143): no 192-byte-per-Gaussian copy per call, and none for the gradient on the way back.  Bit-identical results
# (tests/test_gpu_sh_split.py).  False = the reference's expression.
SPLIT_SH = True


# The three activations of the model -- get_scaling = exp, get_rotation = F.normalize, get_opacity = sigmoid (scene/gaussian_model.py:44-52, 128-147), evaluated
# by every render() call -- in ONE kernel each way (ibgs_amd/activations.py) instead of ~17 torch launches (exp, sigmoid, norm, clamp, expand, div and their
# backward: ~0.09 ms of a 2.0 ms trainer iteration at 1 M Gaussians).  Only for a model that keeps the reference's raw parameters and activation functions
# (`activations.standard_model`); anything else, and FUSED_ACTIVATIONS = False, goes through the model's own properties.
FUSED_ACTIVATIONS = True


def _activated(pc):
    """(scales, rotations, opacities) of the model."""
    if FUSED_ACTIVATIONS and standard_model(pc):
        return fused_activations(pc._scaling, pc._rotation, pc._opacity)
    return pc.get_scaling, pc.get_rotation, pc.get_opacity


def _appearance(pc, pipe, viewpoint_camera, scaling_modifier, override_color):
    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
        opacities = pc.get_opacity
    else:
        scales, rotations, opacities = _activated(pc)
    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            feats = pc.get_features
            shs_view = feats.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = pc.get_xyz - viewpoint_camera.camera_center.to(feats.device).repeat(feats.shape[0], 1)
            dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dir_pp) + 0.5, 0.0)
        else:
            dc, rest = getattr(pc, "_features_dc", None), getattr(pc, "_features_rest", None)
            if (SPLIT_SH and torch.is_tensor(dc) and torch.is_tensor(rest) and dc.is_cuda and dc.dim() == 3 and rest.dim() == 3 and dc.shape[1] == 1
                    and rest.shape[1] >= 1 and dc.is_contiguous() and rest.is_contiguous() and dc.dtype == torch.float32 and rest.dtype == torch.float32):
                shs = (dc, rest)          # (render / render_depth unpack it into shs= / shs_rest=)
            else:
                shs = pc.get_features
    else:
        colors_precomp = override_color
    return scales, rotations, cov3D_precomp, shs, colors_precomp, opacities


def _sh_kw(shs):
    return dict(shs=shs[0], shs_rest=shs[1]) if isinstance(shs, tuple) else dict(shs=shs)


# The |dL/dmean2D| statistic (viewspace_points_abs.grad) has one reader, densification (train.py:400-410, until densify_until_iter).  A trainer
# that is past it -- or a caller that never densifies -- sets this False: the abs sink is then created without requires_grad, the rasterizer's
# backward runs with IBGS_FLAG_NO_ABS_GRAD (the colour blend skips the two |.| moments: -8 % of that kernel) and `viewspace_points_abs.grad`
# stays None.  Default: the reference's behaviour.
TRACK_ABS_GRAD = True


# The reference builds its two gradient sinks as `torch.zeros_like(xyz, requires_grad=True) + 0` per call (gaussian_renderer/__init__.py:153-159): two
# fills and two adds of a (P, 3) tensor whose VALUES nobody reads (the rasterizer uses neither; train.py:404-405 reads `.grad` only).  SHARED_SINKS: the
# zeros are ONE cached read-only buffer per (device, P), and each call gets two fresh non-leaf views of it through a no-op autograd node -- same
# properties (zeros, requires_grad, not a leaf, retain_grad() works, `.grad` receives dL/dmean2D), no kernel.  False = the reference's expression.
# LEAF_SINKS (opt-in, round 6): the two aliases are leaves instead -- everything a reader of `.grad` sees is the same, `is_leaf` is not (tests/test_glue_golden.py pins
# the reference's value of it, hence not the default); saves the two (P, 3) clones per iteration that a non-leaf's retain_grad() hook makes.
SHARED_SINKS = True
LEAF_SINKS = False         # (round 6, opt-in) True: the two sinks are LEAF aliases of the shared zeros; False: non-leaf views with retain_grad(), like the reference's `zeros + 0`
_sink_zeros = {}


class _SinkView(torch.autograd.Function):
    """zeros (shared, never written) -> a non-leaf alias that requires grad; nothing flows further back."""

    @staticmethod
    def forward(ctx, zeros, anchor):
        return zeros.view_as(zeros)

    @staticmethod
    def backward(ctx, grad):
        return None, None


def _sinks(pc):
    xyz = pc.get_xyz
    if SHARED_SINKS and xyz.is_cuda:
        key = (xyz.device, tuple(xyz.shape), xyz.dtype)
        ent = _sink_zeros.get(key)
        if ent is None:
            if len(_sink_zeros) > 8:
                _sink_zeros.clear()
            # (the third member: what a caller gets as `viewspace_points_abs` when nobody tracks it -- a zero-stride expansion of ONE row of zeros, which torch
            # refuses to write into; handing out the shared buffer itself let an in-place write corrupt every later call's zeros: ADVICE r5)
            ent = _sink_zeros[key] = (torch.zeros_like(xyz, requires_grad=False), torch.zeros(1, device=xyz.device, requires_grad=True),
                                      torch.zeros((1,) + tuple(xyz.shape[1:]), dtype=xyz.dtype, device=xyz.device).expand(tuple(xyz.shape)))
        z, anchor, z_ro = ent
        if LEAF_SINKS:
            # a fresh LEAF alias of the shared zeros per call: `.grad` is then filled by autograd's AccumulateGrad, which takes the backward's tensor as it is; the
            # retain_grad() hook of a non-leaf (the reference's `zeros + 0`, and _SinkView below) CLONES it -- two (P, 3) copies per iteration that nobody needs
            a = z.detach().requires_grad_(True)
            b = z.detach().requires_grad_(True) if TRACK_ABS_GRAD else z_ro
            return a, b
        a = _SinkView.apply(z, anchor)
        b = _SinkView.apply(z, anchor) if TRACK_ABS_GRAD else z_ro
        if a.requires_grad:          # (not under torch.no_grad(): render.py:297)
            a.retain_grad()
            if TRACK_ABS_GRAD:
                b.retain_grad()
        return a, b
    a = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True) + 0
    b = (torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True) + 0) if TRACK_ABS_GRAD else torch.zeros_like(xyz, dtype=xyz.dtype)
    try:
        a.retain_grad()
        if TRACK_ABS_GRAD:
            b.retain_grad()
    except Exception:
        pass
    return a, b


def _no_sources(viewpoint_camera, dev):
    hw = int(viewpoint_camera.image_height) * int(viewpoint_camera.image_width)
    return (1, torch.zeros((1, 16), device=dev), torch.zeros((1, 3, hw), device=dev),
            torch.zeros((1, 1, hw), device=dev), torch.zeros((1, 3), device=dev))


def render_depth(viewpoint_camera, pc, scene, pipe, args, bg_color: torch.Tensor, learnt_normal: bool,
                 nb_src_frames: int, buffer_length: int, depth_error_threshold: Optional[float] = None,
                 scaling_modifier=1.0, override_color=None):
    """Depth-only pass (median ray/plane depth), reference gaussian_renderer/__init__.py:41-140."""
    dev = pc.get_xyz.device
    means2D, means2D_abs = _sinks(pc)
    means3D = pc.get_xyz
    scales, rotations, cov3D_precomp, shs, colors_precomp, opacities = _appearance(pc, pipe, viewpoint_camera, scaling_modifier, override_color)
    if depth_error_threshold is None:
        depth_error_threshold = getattr(args, "depth_error_threshold", 0.01)
    n, ref_to_src_list, src_images, src_rendered_depths, src_cam_pos = _no_sources(viewpoint_camera, dev)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        ref_to_src_list=ref_to_src_list, src_cam_pos=src_cam_pos, src_images=src_images,
        src_rendered_depths=src_rendered_depths, nb_src_images=n, buffer_length=buffer_length,
        depth_error_threshold=float(depth_error_threshold), sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, render_geo=False, render_depth_only=True,
        debug=pipe.debug)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    outs = rasterizer(means3D=means3D, means2D=means2D, means2D_abs=means2D_abs, **_sh_kw(shs),
                      colors_precomp=colors_precomp, opacities=opacities, scales=scales, rotations=rotations,
                      cov3D_precomp=cov3D_precomp, **_plane_inputs(pc, viewpoint_camera, learnt_normal, means3D, scales, rotations))
    return outs[3]


def render_depth_batch(viewpoint_cameras, pc, scene, pipe, args, bg_color, learnt_normal: bool, nb_src_frames: int,
                       buffer_length: int, depth_error_threshold: Optional[float] = None, scaling_modifier=1.0,
                       override_color=None, _activated_by_caller=None):
    """Depth maps of several cameras in ONE rasterizer pass: same results as
    `[render_depth(c, ...) for c in viewpoint_cameras]` stacked to (n, 1, H, W).  Falls back to that loop when the
    cameras differ in size or the fused plane-map inputs are unavailable."""
    from .rasterizer import rasterize_depth_batch, _lib as _rl
    cams = list(viewpoint_cameras)
    H, W = int(cams[0].image_height), int(cams[0].image_width)
    same = all(int(c.image_height) == H and int(c.image_width) == W for c in cams)
    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
        opacities = pc.get_opacity
    else:          # (render() hands over what it has just computed for its own pass: the same model, the same call)
        scales, rotations, opacities = _activated_by_caller if _activated_by_caller is not None else _activated(pc)
    plane = _plane_inputs(pc, cams[0], learnt_normal, pc.get_xyz, scales, rotations) if FUSED_PLANE_MAP else {}
    if not same or "plane_mode" not in plane or len(cams) > _rl.MAX_VIEWS:
        return torch.stack([render_depth(c, pc, scene, pipe, args, bg_color, learnt_normal, nb_src_frames, buffer_length,
                                         depth_error_threshold, scaling_modifier, override_color) for c in cams], dim=0)
    dev = pc.get_xyz.device
    vms = torch.stack([c.world_view_transform.to(dev) for c in cams])
    pms = torch.stack([c.full_proj_transform.to(dev) for c in cams])
    cps = torch.stack([c.camera_center.to(dev) for c in cams])
    depths, _ = rasterize_depth_batch(pc.get_xyz, opacities, scales, rotations, cov3D_precomp, scaling_modifier, vms, pms, cps,
                                      [math.tan(c.FoVx * 0.5) for c in cams], [math.tan(c.FoVy * 0.5) for c in cams], H, W,
                                      buffer_length, plane.get("plane_normal"), plane.get("plane_offset"), plane["plane_mode"],
                                      debug=pipe.debug)
    return depths


def find_closest_frames(viewpoint_camera, scene, args):
    """Neighbour search of the test-time path (reference :200-227): sort training views by distance
    then angle, keep those inside the angle / distance window, optionally move the most similar pose
    to the front."""
    dev = scene.camera_centers.device
    camera_center = viewpoint_camera.camera_center.to(dev)
    R = torch.as_tensor(viewpoint_camera.R, dtype=torch.float32, device=dev)
    center_ray = torch.tensor([0.0, 0.0, 1.0], device=dev) @ R.transpose(-1, -2)
    dist = torch.norm(camera_center.unsqueeze(0) - scene.camera_centers, dim=-1).detach().cpu().numpy()
    ang = (torch.arccos(torch.sum(center_ray.unsqueeze(0) * scene.center_rays, dim=-1)) * 180 / torch.pi).detach().cpu().numpy()
    order = np.lexsort((ang, dist))
    keep = (ang[order] < args.multi_view_max_angle) & (dist[order] > args.multi_view_min_dis) & (dist[order] < args.multi_view_max_dis)
    order = order[keep]
    order = order[:min(args.multi_view_num, len(order))].tolist()
    if getattr(args, "enable_exposure_correction", False) and len(order) > 0:
        w2c = viewpoint_camera.world_view_transform.T.to(dev)
        rel = torch.matmul(w2c.unsqueeze(0), torch.inverse(scene.world_view_transforms))
        diff = torch.mean(torch.abs(rel - torch.eye(4, device=dev).unsqueeze(0)), dim=[1, 2]).detach().cpu().numpy()
        best = order[int(np.argmin(diff[order]))]
        order.remove(best)
        order = [best] + order
    return np.array(order)


# What render() derives from the scene's per-view tables for a (reference camera, chosen sources) pair does not change while the tables do not: the
# stack of source images (4 x 3 x H x W: a 100 MB copy per call at 1080p, and a NEW tensor object each time -- which also defeats the rasterizer's
# one-pack-per-source-stack cache, rasterizer.TEX_CACHE: +1 pack kernel) and the pose algebra (two LU inversions through rocSOLVER: ~15 small kernels).  Both
# are kept per (tables' identity + version counters, camera matrix identity + version, chosen indices); an in-place write to a table moves its version
# counter and drops the entry.  Same tensors, same arithmetic -- computed once.  SOURCE_CACHE_BYTES bounds the image stacks kept (least recently used first out):
# None = min(8 GiB, 2.5 % of the device memory free at first use).  Round 6: the scene's tables are remembered by WEAK reference (a discarded Scene is
# not kept alive; its entries go with the next insertion), and sources drawn with random.sample (`args.shuffle_source_frame`) are not cached at all -- their key
# space is combinatorial, every call would miss and leave a dead stack (and a dead RGBA pack in rasterizer.TEX_CACHE) behind.  The cached stacks and the sinks are
# READ-ONLY for callers; clear_caches() drops everything.
SOURCE_CACHE_BYTES = None
_src_cache = {}          # key -> [versions, weak references to the tables, stacked images, ref_to_src_list, src_cam_pos]


def clear_caches():
    """Drop everything this module and the rasterizer shim keep between calls (source stacks, pose algebra, RGBA packs, scratch, launch orders, sinks)."""
    _src_cache.clear(); _sink_zeros.clear()
    _rz.clear_caches()


def _version(t):
    try:
        return t._version
    except (RuntimeError, AttributeError):
        return None


def _cached_sources(scene, viewpoint_camera, chosen, dev):
    imgs, w2s_all, V = scene.original_image_list, scene.world_view_transforms, viewpoint_camera.world_view_transform
    vers = (_version(imgs), _version(w2s_all), _version(V))
    if None in vers:          # inference tensors: nothing to key on
        return None, None
    key = (id(imgs), id(w2s_all), id(V), tuple(int(i) for i in chosen), str(dev))
    ent = _src_cache.get(key)
    if ent is not None and ent[0] == vers and ent[1][0]() is imgs and ent[1][1]() is w2s_all and ent[1][2]() is V:
        _src_cache[key] = _src_cache.pop(key)          # most recently used last
        return ent, key
    return None, key


def _remember_sources(key, scene, viewpoint_camera, src_images, ref_to_src_list, src_cam_pos):
    if key is None:
        return
    imgs, w2s_all, V = scene.original_image_list, scene.world_view_transforms, viewpoint_camera.world_view_transform
    _src_cache.pop(key, None)
    for k in [k for k, e in _src_cache.items() if any(r() is None for r in e[1])]:          # entries of tables that no longer exist
        del _src_cache[k]
    _src_cache[key] = [(_version(imgs), _version(w2s_all), _version(V)), (weakref.ref(imgs), weakref.ref(w2s_all), weakref.ref(V)), src_images, ref_to_src_list, src_cam_pos]
    total = sum(e[2].numel() * e[2].element_size() for e in _src_cache.values())
    cap = _rz._cache_cap(src_images.device, SOURCE_CACHE_BYTES, 8 << 30, 0.025)
    while total > cap and len(_src_cache) > 1:
        k0 = next(iter(_src_cache))
        e = _src_cache.pop(k0)
        total -= e[2].numel() * e[2].element_size()


def _rows(t, idx):
    """t[idx] for a short Python list of indices WITHOUT letting torch build the index tensor on the host: that is a pageable host-to-device
    copy, i.e. a wait for everything queued on the stream (the previous pass's render), after which the glue's small kernels run one by one on
    an idle GPU -- 0.45 ms of idle time per test-time frame.  Views + one stack kernel instead."""
    return torch.stack([t[int(i)] for i in idx])


def render(viewpoint_camera, pc, scene, pipe, args, bg_color: torch.Tensor, learnt_normal: bool,
           nb_src_frames: int, buffer_length: int, depth_error_threshold: Optional[float] = None,
           scaling_modifier=1.0, override_color=None, app_model=None, render_geo=True, return_depth_normal=True,
           do_find_closest_frame=False, do_render_src_depth=False, render_depth_only=False):
    """Render one view; returns the reference's dictionary (gaussian_renderer/__init__.py:349-363)."""
    dev = pc.get_xyz.device
    screenspace_points, screenspace_points_abs = _sinks(pc)
    if depth_error_threshold is None:
        depth_error_threshold = getattr(args, "depth_error_threshold", 0.01)
    depth_error_threshold = float(depth_error_threshold)
    means3D = pc.get_xyz
    scales, rotations, cov3D_precomp, shs, colors_precomp, opacities = _appearance(pc, pipe, viewpoint_camera, scaling_modifier, override_color)

    depth_slots = None
    if render_geo:
        nearest = find_closest_frames(viewpoint_camera, scene, args) if do_find_closest_frame else viewpoint_camera.nearest_id
        if len(nearest) == 0:
            nb_src_frames, ref_to_src_list, src_images, src_rendered_depths, src_cam_pos = _no_sources(viewpoint_camera, dev)
        else:
            nb_src_frames = min(nb_src_frames, len(nearest))
            shuffled = bool(getattr(args, "shuffle_source_frame", False))
            if shuffled:
                chosen = random.sample(list(nearest), nb_src_frames)
            else:
                chosen = list(nearest[:nb_src_frames])
            cached, ckey = (None, None) if shuffled else _cached_sources(scene, viewpoint_camera, chosen, dev)          # (a random draw: nothing to cache, see SOURCE_CACHE_BYTES)
            src_images = cached[2] if cached is not None else _rows(scene.original_image_list, chosen)
            if do_render_src_depth:
                # the reference loops render_depth over the sources (:245-253); here they share ONE rasterizer pass
                src_rendered_depths = render_depth_batch([scene.getTrainCameras()[i] for i in chosen], pc, scene, pipe, args, bg_color,
                                                         learnt_normal, nb_src_frames, buffer_length, depth_error_threshold,
                                                         scaling_modifier, override_color,
                                                         _activated_by_caller=(scales, rotations, opacities) if scales is not None else None)
            else:
                table = scene.rendered_depth_list
                if (DEPTH_TABLE and torch.is_tensor(table) and table.is_cuda and table.device == torch.device(dev) and table.dtype == torch.float32 and table.is_contiguous()
                        and table.dim() in (3, 4) and int(table.shape[-1]) == int(viewpoint_camera.image_width) and int(table.shape[-2]) == int(viewpoint_camera.image_height)
                        and (table.dim() == 3 or int(table.shape[1]) == 1)):
                    # the depth cache as it is + the sources' plane numbers: the rasterizer reads plane chosen[m] for source m (IBGS_FLAG_SRC_DEPTH_SLOTS) --
                    # no stack of n_src planes per call (33 MB read + written at 1080p, 4 sources: ~20 us of every iteration)
                    src_rendered_depths, depth_slots = table, tuple(int(i) for i in chosen)
                else:
                    src_rendered_depths = _rows(table, chosen)
            if cached is not None:
                ref_to_src_list, src_cam_pos = cached[3], cached[4]
            else:
                world_to_src = _rows(scene.world_view_transforms, chosen).to(dev)
                # (torch.inverse looks at its `info` result on the host, i.e. waits for the whole queue -- here the batched depth pass -- in the middle of
                # the frame, and the GPU then idles while the main pass is being queued; inv_ex is the same LU without that look)
                src_to_world = torch.linalg.inv_ex(world_to_src).inverse
                ref_to_world = torch.linalg.inv_ex(viewpoint_camera.world_view_transform.T.to(dev)).inverse
                ref_to_src_list = world_to_src @ ref_to_world.unsqueeze(0)
                src_cam_pos = src_to_world[:, :3, 3].contiguous()
                src_images = src_images.to(dev)
                if shuffled:
                    src_images._ibgs_transient = True          # one-off stack: the rasterizer packs its RGBA into the shared scratch slot (rasterizer.TEX_CACHE)
                _remember_sources(ckey, scene, viewpoint_camera, src_images, ref_to_src_list, src_cam_pos)
            src_rendered_depths = src_rendered_depths.to(dev)
    else:
        nb_src_frames, ref_to_src_list, src_images, src_rendered_depths, src_cam_pos = _no_sources(viewpoint_camera, dev)

    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        ref_to_src_list=ref_to_src_list, src_cam_pos=src_cam_pos, src_images=src_images,
        src_rendered_depths=src_rendered_depths, nb_src_images=nb_src_frames, buffer_length=buffer_length,
        depth_error_threshold=depth_error_threshold, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, render_geo=render_geo,
        render_depth_only=render_depth_only, debug=pipe.debug, src_depth_slots=depth_slots)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    plane_kw = _plane_inputs(pc, viewpoint_camera, learnt_normal, means3D, scales, rotations) if (render_geo or render_depth_only) else {}

    (rendered_image, radii, out_normal_map, out_median_intersected_depth, out_cam_feat, out_warped_image,
     out_min_depth_diff, out_camera_ray, use_first_src_frame_mask) = rasterizer(
        means3D=means3D, means2D=screenspace_points, means2D_abs=screenspace_points_abs, **_sh_kw(shs),
        colors_precomp=colors_precomp, opacities=opacities, scales=scales, rotations=rotations,
        cov3D_precomp=cov3D_precomp, **plane_kw)

    rendered_normal = out_normal_map[0:3] if render_geo else None
    if return_depth_normal:
        if FUSED_DEPTH_NORMAL and out_median_intersected_depth.is_cuda and all(hasattr(viewpoint_camera, k) for k in ("Fx", "Fy", "Cx", "Cy")):
            from .depthnormal import depth_normal
            dn = depth_normal(viewpoint_camera, out_median_intersected_depth.squeeze())          # one kernel (and one in the backward) instead of ~25 each way
        else:
            dn = render_normal(viewpoint_camera, out_median_intersected_depth.squeeze())
            dn = dn / (torch.norm(dn, dim=0, keepdim=True) + 1e-8)
    else:
        dn = None
    if app_model is not None and getattr(pc, "use_app", False):
        ab = app_model.appear_ab[torch.tensor(viewpoint_camera.uid, device=dev)]
        app_image = torch.exp(ab[0]) * rendered_image + ab[1]
    else:
        app_image = None

    return {"render": rendered_image, "app_image": app_image,
            "viewspace_points": screenspace_points, "viewspace_points_abs": screenspace_points_abs,
            "visibility_filter": radii > 0, "radii": radii, "rendered_normal": rendered_normal,
            "median_intersected_depth": out_median_intersected_depth, "median_intersected_depth_normal": dn,
            "cam_feat": out_cam_feat, "warped_image": out_warped_image, "min_depth_diff": out_min_depth_diff,
            "camera_ray": out_camera_ray, "use_first_src_frame_mask": use_first_src_frame_mask}
