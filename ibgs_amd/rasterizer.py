"""Host-side mirror of the reference operator for the plane rasterizer.

Same names, argument order, return values and error behaviour as
``submodules/diff-plane-rasterization/diff_plane_rasterization/__init__.py`` (reference), so that
``gaussian_renderer.render()`` / ``train.py`` / ``render.py`` can import it unchanged:

* ``GaussianRasterizationSettings``  (reference :252-276, 21 fields, same order)
* ``GaussianRasterizer``             (reference :278-331; ``forward`` and ``markVisible``)
* ``rasterize_gaussians`` / ``_RasterizeGaussians`` (reference :21-250; 11 inputs, 9 outputs, 11 grads)
* ``_C``  -- an object with ``rasterize_gaussians`` (29 args), ``rasterize_gaussians_backward`` (34 args)
  and ``mark_visible`` like the reference's pybind module (ext.cpp:15-19, rasterize_points.cu:37-292),
  implemented over the C ABI of ``libibgs_rast.so`` (include/ibgs_rast.h).

PyTorch is used for device memory, streams and autograd plumbing only; all arithmetic happens in the
HIP library.  If the library is missing the import of ``_lib`` raises -- there is no fallback.
"""
import collections
import weakref
import ctypes
import os
import threading
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib

NUM_CHANNELS = 3
NUM_NORMAL_CHANNELS = 3
NUM_PLANE_PARAMS = 5
M_SRC = _lib.MAX_SRC

# Tests may flip this to emulate the CUDA texture unit's 8-bit filter weights.
TEX_QUANT = False
# Exact tile culling (shorter per-tile lists, identical outputs).  False reproduces the reference's AABB lists.
TILE_CULL = True

# The reference's `if (power > 0.0f) continue;` (forward.cu:420, backward.cu:645): reproduced for the Gaussians whose conic is close enough
# to singular for it to fire (csrc/common.h); False = IBGS_FLAG_NO_REF_POWER_SKIP, every Gaussian takes the fast path.
REF_POWER_SKIP = True
# The backward sums the pairs of those Gaussians in a well-conditioned form of the reference's chain (dL/dcov2D = 0.5 sum q l l^T: csrc/render_bwd.hip).  True =
# IBGS_FLAG_REF_ARITH: the reference's own arithmetic for them instead -- its eight per-pair quantities in its association and the chain of backward.cu:405-420 on
# their sums (SURVEY Q1 as a switch; farther from a float64 evaluation, like every fp32 evaluation of those expressions).
REF_ARITH = False

KEEP_DET_SCRATCH = False          # diagnostics: keep the last deterministic backward's scratch alive as _CModule.last_det
# Deterministic backward (IBGS_FLAG_DETERMINISTIC): no float atomics, gradients bit-identical from run to run (CI mode, slower).
DETERMINISTIC = False

# means2D_abs without requires_grad (nobody will read the |dL/dmean2D| statistic: after densify_until_iter, at test time) = IBGS_FLAG_NO_ABS_GRAD:
# the colour blend skips the two |.| moments.  False: always compute them (the reference does).
NO_ABS_GRAD_WHEN_UNUSED = True

# Work decomposition of the colour blend kernels: None = by frame size (one wave per 16x16 tile from 4096 tiles on,
# one wave per 8x8 quadrant below), "tile" / "quadrant" force one of them (tests run both against the oracle).
WAVE_SHAPE = None


def _shape_flag():
    return ({None: 0, "tile": _lib.FLAG_TILE_WAVES, "quadrant": _lib.FLAG_QUADRANT_WAVES}[WAVE_SHAPE]
            | (0 if REF_POWER_SKIP else _lib.FLAG_NO_REF_POWER_SKIP) | (_lib.FLAG_REF_ARITH if REF_ARITH else 0))


# View-parallel training (ibgs_amd/dist.py): while a `capture_sh_factors()` block is active the backward leaves
# dL/dsh unwritten (returns None for it) and records the per-view factors -- clamp-masked dL/dRGB (P, 3), camera
# centre, active degree -- so that ranks exchange 3 floats per Gaussian instead of 3 M (include/ibgs_rast.h,
# IBGS_FLAG_SH_FACTORED / ibgs_sh_grad_from_views).
_sh_factor_sink = None


# Zero-copy gradient hand-over for the view-parallel exchange (ibgs_amd/dist.py): while set, the NEXT backward writes dL/dmeans3D,
# dL/dopacity, dL/dscales, dL/drotations straight into these caller-owned buffers (views of the flat all-reduce bucket; ibgs_backward
# writes to caller-supplied pointers anyway) and returns them as its gradients -- when the leaves are fed to the rasterizer as they are,
# autograd adopts them as `.grad` and the exchange needs no pack copy.  {"means3D": (P,3), "opacities": (P,1), "scales": (P,3),
# "rotations": (P,4)} -> tensors; consumed (emptied) by the first backward that uses it.
_grad_out_sink = None


def _sink_or_new(name, shape, new, opts):
    sink = _grad_out_sink
    if sink:
        t = sink.pop(name, None)
        if t is not None and tuple(t.shape) == tuple(shape) and t.is_contiguous() and t.dtype == torch.float32 and t.device == opts["device"]:
            return t
    return new(*shape, **opts)


class capture_sh_factors:
    def __enter__(self):
        global _sh_factor_sink
        self._prev, self.items = _sh_factor_sink, []
        _sh_factor_sink = self.items
        return self.items

    def __exit__(self, *exc):
        global _sh_factor_sink
        _sh_factor_sink = self._prev
        return False


# R (the number of (Gaussian, tile) pairs) of recent forwards with the same (device, P, W, H, mode): the next call passes
# 1.25 x their maximum as ibgs_forward_args.rendered_hint so that the host does not stall the GPU while R travels back
# (include/ibgs_rast.h).  The maximum over the last RENDERED_WINDOW calls covers a trainer that hops between cameras; a
# view that still exceeds the hint costs one repeated binning + render pass, never a wrong result.
# RENDERED_HINT = False restores the reference's synchronous sizing.
RENDERED_HINT = True
RENDERED_WINDOW = 16
HINT_MISSES = 0          # forwards whose hint was too small (binning + render ran twice); a trainer hopping between cameras pays these only while the window fills
_last_rendered = {}           # key -> R of the last call, or a list of recent R values
LAST_NUM_RENDERED = 0         # diagnostic: R of the most recent forward
LAST_BINNING_CAPACITY = 0     # diagnostic: the size (in pairs) the most recent forward carved its binning arena for

# Launch order hints for the colour forward (include/ibgs_rast.h: ibgs_forward_args.tile_order_hint).  The backward of a colour pass leaves the
# balanced order it launched its tiles in inside the image arena; a copy is kept per camera and handed to that camera's next forward, whose
# lists saturate where they did before.  The camera is recognised by its view matrix: the tensor's address when it lives on the device (the
# reference keeps `world_view_transform` per camera), its 64 bytes otherwise.  A wrong or stale hint costs performance only (the library
# checks the words and ignores anything that is not a tile order).  ORDER_HINT = False: never pass one.
ORDER_HINT = True
# ... for geo passes as well: their forward runs two waves per tile behind a queue and would take the heaviest tile first.  Off by default: that
# forward lives on the 8 x 8 block map's L2 locality (source texels), and giving it up costs more than the balance brings on even scenes
# (C3-geo forward 0.668 -> 0.716 ms, trained 0.375 -> 0.398 ms; half of the Gaussians in one blob: 0.751 -> 0.677 ms)
ORDER_HINT_GEO = os.environ.get("IBGS_ORDER_HINT_GEO", "0") == "1"          # (the environment switch is for A/B runs: tools/ab_env.sh)
ORDER_HINT_MAX = 512          # cameras remembered (32 KB each at 1080p); the least recently used one goes first
_order_hints = collections.OrderedDict()


def _camera_key(viewmatrix, device, W, H, geo, stream):
    """(a colour pass and a geo pass of one camera keep separate orders: their kernels hold different numbers of waves per SIMD; and one buffer
    per stream, like the other cached scratch: the backward that writes it and the forward that reads it are then ordered by the stream)"""
    if not torch.is_tensor(viewmatrix):
        return None
    if viewmatrix.is_cuda:
        return (device.index, stream, W, H, bool(geo), "p", viewmatrix.data_ptr())
    return (device.index, stream, W, H, bool(geo), "b", viewmatrix.detach().float().contiguous().numpy().tobytes())


_tex_scratch = {}
_gacc_scratch = {}   # (device index, stream, P) -> [zeroed P x 16 tensor, dirty flag]; ibgs_backward re-zeroes what it consumed


def cpu_deep_copy_tuple(input_tuple):
    copied_tensors = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied_tensors)


def _dev_f32(t, device):
    """contiguous fp32 tensor on `device`, or None for the reference's 'empty tensor = not provided'."""
    if t is None or t.numel() == 0:
        return None
    # the common case first -- an fp32, contiguous, 16-byte-aligned tensor that already lives on the device -- in as few Python-level calls as possible
    # (this runs ~30 times per step; docs/EXPERIMENTS.md section 7, host time)
    if t.dtype is torch.float32 and t.device == device and t.is_contiguous() and not (t.data_ptr() & 15):
        return t
    if t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    t = t.contiguous()
    if t.data_ptr() & 15:          # a view that starts in the middle of an allocation: the kernels fetch rows with 16-byte loads
        t = t.clone()
    return t


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream_key(device):
    """Scratch is reused across calls, so it is owned by ONE stream: two streams (or threads with their own streams) that run
    rasterizer calls side by side get separate buffers instead of racing on a shared one."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return (idx, torch.cuda.current_stream(device).cuda_stream)


def _tex(device, nbytes, what="tex"):
    key = _stream_key(device) + (what,)
    if len(_tex_scratch) > 8 and key not in _tex_scratch:
        _tex_scratch.clear()
    buf = _tex_scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _tex_scratch[key] = buf
    return buf


_tex_writes = [0]

# T1 once per SOURCE SET instead of once per call ("build replaces with a cached packed buffer"; the reference packs,
# allocates and synchronises per forward AND per backward, rasterizer_impl.cu:366, 582).  When a forward is handed the very tensor object
# whose pack is still held -- same object, same version counter (no in-place write since), same n / W / H -- the pack kernel is skipped
# (IBGS_FLAG_TEX_PACKED).  Round 5: the packs are kept PER SOURCE STACK (least recently used first out, TEX_CACHE_BYTES in total per stream), not
# only the last one: a trainer hops between cameras, each with its own stack (renderer.render keeps those per camera), and with one slot every
# step packed again (40 us per geo forward at 1080p).  A slot remembers its tensor by WEAK reference (round 6: a strong one kept discarded scenes alive): a stack
# that has been freed -- and whose id() a new tensor may have inherited -- no longer matches.  A stack marked `_ibgs_transient` (renderer.render: sources drawn with
# random.sample, a fresh stack per call that would never hit) packs into the shared scratch slot instead of filling the pool with dead entries.
# TEX_CACHE = False: always pack.  TEX_CACHE_BYTES = None: min(16 GiB, 5 % of the device memory that is free at first use) per stream; clear_caches() drops everything.
TEX_CACHE = True
TEX_CACHE_BYTES = None
_tex_pool = {}          # stream key -> OrderedDict[slot -> buffer]; slot = ("src", id(source tensor)) or "scratch"


def _cache_cap(device, configured, ceiling, fraction):
    """A cache's byte cap: the configured number, or -- None -- a fraction of what the device has free right now, under a ceiling (decided once per device)."""
    if configured is not None:
        return int(configured)
    if device.type != "cuda":          # (the glue's CPU tests: nothing to ask)
        return int(ceiling)
    key = (device.index if device.index is not None else torch.cuda.current_device(), ceiling, fraction)
    cap = _cache_caps.get(key)
    if cap is None:
        free, _total = torch.cuda.mem_get_info(device)
        cap = _cache_caps[key] = int(min(ceiling, fraction * free))
    return cap


_cache_caps = {}


def _tex_slot(device, nbytes, slot):
    pool = _tex_pool.get(_stream_key(device))
    if pool is None:
        if len(_tex_pool) > 8:
            _tex_pool.clear()
        pool = _tex_pool[_stream_key(device)] = collections.OrderedDict()
    buf = pool.get(slot)
    if buf is None or buf.numel() < nbytes:
        buf = pool[slot] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    pool.move_to_end(slot)
    total = sum(x.numel() for x in pool.values())
    cap = _cache_cap(device, TEX_CACHE_BYTES, 16 << 30, 0.05)
    while total > cap and len(pool) > 1:
        k0 = next(iter(pool))
        if k0 == slot:
            break
        total -= pool.pop(k0).numel()
    return buf


def _tex_packed(device, nbytes, source=None):
    """The buffer that receives the packed source RGBA (T1), plus a ticket that names this pack: a later call that holds the ticket and finds
    it still current (`_tex_still(ticket)`) knows that nothing overwrote the buffer in between.
    `source` = (tensor, its version counter, n, W, H): what is being packed (see _tex_cached); None / unversioned: the shared scratch slot."""
    cacheable = TEX_CACHE and source is not None and source[1] is not None and not getattr(source[0], "_ibgs_transient", False)
    slot = ("src", id(source[0])) if cacheable else "scratch"
    buf = _tex_slot(device, nbytes, slot)
    _tex_writes[0] += 1
    buf._ibgs_ticket = _tex_writes[0]
    buf._ibgs_src = (weakref.ref(source[0]),) + tuple(source[1:]) if cacheable else None
    return buf, (_stream_key(device), slot, _tex_writes[0], nbytes)


def _tensor_version(t):
    """The autograd version counter, or None when the tensor has none to read: inference tensors raise RuntimeError ("Inference tensors do not
    track version counter"), which `getattr(t, "_version", None)` does not swallow.  None = uncacheable: such a stack is packed on every call."""
    try:
        return t._version
    except (RuntimeError, AttributeError):
        return None


def clear_caches():
    """Drop every buffer the shim keeps between calls: the source-RGBA packs (TEX_CACHE), the per-stream scratch (texture, geo table, moment rows), the
    per-camera launch orders and the R history.  The next calls allocate what they need again.  (renderer.clear_caches() calls this one too.)"""
    _tex_pool.clear(); _tex_scratch.clear(); _gacc_scratch.clear(); _order_hints.clear(); _last_rendered.clear(); _cache_caps.clear()


def invalidate_tex_cache():
    """Forget every cached source-RGBA pack.  The cache notices writes through the version counter only; a write that does not bump it
    (`src_images.data.copy_(...)`, a raw kernel filling a persistent buffer) must be followed by this call."""
    for pool in _tex_pool.values():
        for buf in pool.values():
            buf._ibgs_src = None


def _tex_cached(device, nbytes, source):
    if not TEX_CACHE or source[1] is None:
        return None
    pool = _tex_pool.get(_stream_key(device))
    slot = ("src", id(source[0]))
    buf = pool.get(slot) if pool is not None else None
    had = getattr(buf, "_ibgs_src", None) if buf is not None else None
    if had is None or buf.numel() < nbytes or had[0]() is not source[0] or had[1:] != tuple(source[1:]):
        return None
    pool.move_to_end(slot)
    return buf, (_stream_key(device), slot, buf._ibgs_ticket, nbytes)


def _tex_still(ticket, device, nbytes):
    if ticket is None:
        return None
    skey, slot, serial, had = ticket
    pool = _tex_pool.get(skey)
    buf = pool.get(slot) if pool is not None else None
    if buf is None or skey != _stream_key(device) or had < nbytes or getattr(buf, "_ibgs_ticket", None) != serial:
        return None
    return buf


def _gacc(device, P):
    key = _stream_key(device) + (P,)
    ent = _gacc_scratch.get(key)
    if ent is None:
        if len(_gacc_scratch) > 8:
            _gacc_scratch.clear()
        ent = [torch.zeros(P, 16, dtype=torch.float32, device=device), False]
        _gacc_scratch[key] = ent
    elif ent[1]:            # a previous backward died between accumulation and clean-up
        ent[0].zero_()
    ent[1] = True
    return ent


_zero_scalar = {}


class _CallState(threading.local):
    """Per-thread scratch of the Python layer: ONE ForwardArgs / BackwardArgs structure that is cleared and refilled per call (allocating a
    ~70-field ctypes structure costs ~15 us), and ONE arena callback whose closure hands the binning tensor of the current call back through
    `holder` (creating a ctypes callback per call: ~10 us)."""

    def __init__(self):
        self.fwd = _lib.ForwardArgs()
        self.bwd = _lib.BackwardArgs()
        self.holder = {}
        self.alloc_device = None

        def _alloc(nbytes, _user):
            try:
                self.holder["t"] = torch.empty(int(nbytes), dtype=torch.uint8, device=self.alloc_device)
                return self.holder["t"].data_ptr()
            except Exception as ex:  # surfaces as IBGS_ERR_ALLOC
                self.holder["err"] = ex
                return 0

        self.cb = _lib.ALLOC_FN(_alloc)

    def forward_args(self, device):
        ctypes.memset(ctypes.byref(self.fwd), 0, ctypes.sizeof(self.fwd))
        self.holder.clear()
        self.alloc_device = device
        return self.fwd, self.holder, self.cb

    def backward_args(self):
        ctypes.memset(ctypes.byref(self.bwd), 0, ctypes.sizeof(self.bwd))
        return self.bwd


_call_state = _CallState()


class _on_device:
    """`with torch.cuda.device(d)` only when d is not already the current device (the context manager costs ~8 us per entry)."""

    def __init__(self, device):
        idx = device.index
        self.ctx = None if (idx is None or idx == torch.cuda.current_device()) else torch.cuda.device(device)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


def _zeros_view(shape, device, dtype=torch.float32):
    """Zeros of the given shape as a zero-stride view of ONE cached element per (device, dtype): no allocation,
    no fill kernel.  Read-only by construction (PyTorch refuses in-place writes through overlapping views)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), dtype)
    z = _zero_scalar.get(key)
    if z is None:
        z = torch.zeros(1, dtype=dtype, device=device)
        _zero_scalar[key] = z
    return z.expand(*shape)


def _zero_plane(c, H, W, device, dtype=torch.float32):
    """Outputs that the selected mode never writes: zeros of the reference's shape without the
    reference's per-call memset (rasterize_points.cu:80-90 fills 47 planes every forward)."""
    return _zeros_view((c, H, W), device, dtype)


class _CModule:
    """Stand-in for the reference's pybind module ``diff_plane_rasterization._C``."""
    last_tex = None        # ticket of the most recent pack of source RGBA (picked up by the autograd node)

    @staticmethod
    def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                            cov3D_precomp, all_map, viewmatrix, projmatrix, ref_to_src_list, src_cam_pos,
                            src_images, src_rendered_depths, nb_src_images, buffer_length,
                            depth_error_threshold, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                            campos, prefiltered, render_geo, render_depth_only, debug, plane=None, sh_rest=None, depth_slots=None):
        """The reference's 29 positional arguments; `plane` = (raw_normal or None, raw_offset or None, mode) is this
        library's extension (fused plane-map glue, include/ibgs_rast.h) and replaces `all_map`; `sh_rest`: `sh` holds the DC coefficient only
        (P, 1, 3) and `sh_rest` the others (P, M - 1, 3) -- the model's two arrays instead of their torch.cat (ibgs_forward_args.shs_rest);
        `depth_slots`: `src_rendered_depths` is a table of depth planes (the trainer's depth cache) and source m reads plane depth_slots[m] of it
        (IBGS_FLAG_SRC_DEPTH_SLOTS: no stack of the n_src planes is built per call)."""
        lib = _lib.load()
        if means3D.ndimension() != 2 or means3D.size(1) != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")   # rasterize_points.cu:69-71
        if not means3D.is_cuda:
            raise RuntimeError("means3D must live on a HIP device (libibgs_rast.so has no CPU path)")
        device = means3D.device
        P = int(means3D.size(0)); H = int(image_height); W = int(image_width)
        render_geo = bool(render_geo); render_depth_only = bool(render_depth_only)

        with _on_device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            means3D_c = _dev_f32(means3D, device)
            sh_c = _dev_f32(sh, device); colors_c = _dev_f32(colors, device)
            sh_rest_c = _dev_f32(sh_rest, device) if sh_rest is not None else None
            opacity_c = _dev_f32(opacity, device)
            scales_c = _dev_f32(scales, device); rot_c = _dev_f32(rotations, device)
            cov_c = _dev_f32(cov3D_precomp, device); all_map_c = _dev_f32(all_map, device)
            bg_c = _dev_f32(background, device); vm_c = _dev_f32(viewmatrix, device); pm_c = _dev_f32(projmatrix, device)
            campos_c = _dev_f32(campos, device)
            r2s_c = _dev_f32(ref_to_src_list, device); scp_c = _dev_f32(src_cam_pos, device)
            simg_c = _dev_f32(src_images, device); sdep_c = _dev_f32(src_rendered_depths, device)

            # radii and the colour planes are written for every Gaussian / pixel by the kernels: no memset needed
            radii = torch.empty(P, dtype=torch.int32, device=device)
            write_color = not render_depth_only
            if write_color:
                out_color = (torch.empty if P != 0 else torch.zeros)(NUM_CHANNELS, H, W, dtype=torch.float32, device=device)
            else:
                out_color = _zero_plane(NUM_CHANNELS, H, W, device)
            if render_geo:
                # every element of the geo planes is written by the render kernel (unused source slots as zeros)
                mk = (lambda *shp, **kw: torch.empty(*shp, device=device, **kw)) if P != 0 else (lambda *shp, **kw: torch.zeros(*shp, device=device, **kw))
                out_normal = mk(NUM_NORMAL_CHANNELS, H, W)
                out_depth = mk(1, H, W)
                out_cam_feat = mk(4 * M_SRC, H, W)
                out_warped = mk(3 * M_SRC, H, W)
                out_min_depth_diff = mk(1, H, W)
                out_camera_ray = mk(3, H, W)
                out_mask = mk(1, H, W, dtype=torch.int32)
            else:
                out_normal = _zero_plane(NUM_NORMAL_CHANNELS, H, W, device)
                if render_depth_only:
                    out_depth = (torch.empty if P != 0 else torch.zeros)(1, H, W, dtype=torch.float32, device=device)
                else:
                    out_depth = _zero_plane(1, H, W, device)
                out_cam_feat = _zero_plane(4 * M_SRC, H, W, device)
                out_warped = _zero_plane(3 * M_SRC, H, W, device)
                out_min_depth_diff = _zero_plane(1, H, W, device)
                out_camera_ray = _zero_plane(3, H, W, device)
                out_mask = _zero_plane(1, H, W, device, torch.int32)

            geomBuffer = torch.empty(0, dtype=torch.uint8, device=device)
            binningBuffer = torch.empty(0, dtype=torch.uint8, device=device)
            imgBuffer = torch.empty(0, dtype=torch.uint8, device=device)
            rendered = 0
            if P != 0:
                M = 0 if sh_c is None else int(sh_c.size(1)) + (int(sh_rest_c.size(1)) if sh_rest_c is not None else 0)
                geomBuffer = torch.empty(lib.ibgs_required_geom(P), dtype=torch.uint8, device=device)
                imgBuffer = torch.empty(lib.ibgs_required_img(W, H), dtype=torch.uint8, device=device)
                a, holder, cb = _call_state.forward_args(device)
                a.stream = stream
                a.P, a.D, a.M, a.W, a.H = P, int(degree), M, W, H
                a.means3D = _ptr(means3D_c); a.shs = _ptr(sh_c); a.colors_precomp = _ptr(colors_c); a.shs_rest = _ptr(sh_rest_c)
                a.opacities = _ptr(opacity_c); a.scales = _ptr(scales_c); a.rotations = _ptr(rot_c)
                a.cov3D_precomp = _ptr(cov_c); a.all_map = _ptr(all_map_c)
                if plane is not None and plane[2]:
                    pn_c, po_c = _dev_f32(plane[0], device), _dev_f32(plane[1], device)
                    a.plane_normal = _ptr(pn_c); a.plane_offset = _ptr(po_c); a.plane_mode = int(plane[2]); a.all_map = None
                a.scale_modifier = float(scale_modifier)
                a.bg = _ptr(bg_c); a.viewmatrix = _ptr(vm_c); a.projmatrix = _ptr(pm_c); a.campos = _ptr(campos_c)
                a.tanfovx = float(tan_fovx); a.tanfovy = float(tan_fovy)
                a.n_src = int(nb_src_images)
                a.ref_to_src = _ptr(r2s_c); a.src_cam_pos = _ptr(scp_c); a.src_images = _ptr(simg_c); a.src_depths = _ptr(sdep_c)
                a.buffer_length = int(buffer_length); a.depth_error_threshold = float(depth_error_threshold)
                a.prefiltered = int(bool(prefiltered)); a.render_geo = int(render_geo); a.render_depth_only = int(render_depth_only)
                tex_flag = 0
                a.geom = geomBuffer.data_ptr(); a.geom_bytes = geomBuffer.numel()
                a.img = imgBuffer.data_ptr(); a.img_bytes = imgBuffer.numel()
                a.binning_alloc = cb; a.binning_user = None
                if render_geo:
                    if simg_c is None or simg_c.numel() < int(nb_src_images) * 3 * H * W:
                        raise RuntimeError("src_images must hold nb_src_images x 3 x H x W values")
                    if depth_slots is not None:
                        slots = [int(x) for x in depth_slots]
                        if len(slots) < int(nb_src_images) or min(slots[:int(nb_src_images)]) < 0:
                            raise RuntimeError("depth_slots must name one non-negative plane per source")
                        if sdep_c is None or sdep_c.numel() < (max(slots[:int(nb_src_images)]) + 1) * H * W:
                            raise RuntimeError("src_rendered_depths (a table of planes) does not hold plane %d" % max(slots[:int(nb_src_images)]))
                        for m_ in range(M_SRC):
                            a.src_depth_slot[m_] = slots[m_] if m_ < int(nb_src_images) else 0
                    elif sdep_c is None or sdep_c.numel() < int(nb_src_images) * H * W:
                        raise RuntimeError("src_rendered_depths must hold nb_src_images x 1 x H x W values")
                    # the packed RGBA of the sources goes to the per-stream scratch; the ticket lets the backward of this call
                    # skip its own pack when no other geo call used the scratch in between (the training loop's normal case)
                    nbytes = lib.ibgs_required_tex(int(nb_src_images), W, H)
                    source = (src_images, _tensor_version(src_images), int(nb_src_images), W, H)
                    hit = _tex_cached(device, nbytes, source)
                    if hit is not None:
                        tex, _CModule.last_tex = hit
                        tex_flag = _lib.FLAG_TEX_PACKED          # this stream's scratch still holds the pack of this very image stack
                    else:
                        tex, _CModule.last_tex = _tex_packed(device, nbytes, source)
                    a.tex = tex.data_ptr(); a.tex_bytes = tex.numel()
                a.flags = ((_lib.FLAG_DEBUG if debug else 0) | (_lib.FLAG_TEX_QUANT if TEX_QUANT else 0)
                           | (0 if TILE_CULL else _lib.FLAG_NO_TILE_CULL) | _shape_flag() | tex_flag
                           | (_lib.FLAG_SRC_DEPTH_SLOTS if (render_geo and depth_slots is not None) else 0))
                a.out_color = out_color.data_ptr() if write_color else None
                a.radii = radii.data_ptr()
                if render_geo:
                    a.out_normal = out_normal.data_ptr(); a.out_depth = out_depth.data_ptr()
                    a.out_cam_feat = out_cam_feat.data_ptr(); a.out_warped = out_warped.data_ptr()
                    a.out_min_depth_diff = out_min_depth_diff.data_ptr(); a.out_camera_ray = out_camera_ray.data_ptr()
                    a.out_mask = out_mask.data_ptr()
                elif render_depth_only:
                    a.out_depth = out_depth.data_ptr()
                if ORDER_HINT and (ORDER_HINT_GEO or not render_geo) and not render_depth_only and not debug:
                    ckey = _camera_key(viewmatrix, device, W, H, render_geo, stream)
                    oh = _order_hints.get(ckey)
                    if oh is not None:
                        _order_hints.move_to_end(ckey)
                        a.tile_order_hint = oh.data_ptr()
                hkey = (device.index, P, W, H, render_geo, render_depth_only)
                hist = _last_rendered.get(hkey) if (RENDERED_HINT and not debug) else None
                prev = (max(hist) if isinstance(hist, list) else int(hist)) if hist else 0
                a.rendered_hint = (prev + prev // 4 + 4096) if prev > 0 else 0
                pre = None
                if prev > 0:          # the arena for the hinted pass, allocated HERE: no call back from the library into Python (~10 us through ctypes)
                    pre = torch.empty(lib.ibgs_required_binning(int(a.rendered_hint), W, H), dtype=torch.uint8, device=device)
                    a.binning = pre.data_ptr(); a.binning_bytes = pre.numel()
                rc = lib.ibgs_forward(ctypes.byref(a))
                if rc < 0:
                    if "err" in holder:
                        raise holder["err"]
                    raise RuntimeError("ibgs_forward failed (%d): %s" % (rc, _lib.last_error()))
                rendered = int(rc)
                if pre is not None and "t" not in holder:
                    binningBuffer = pre          # (a too small hint: the repeated pass asked the callback, whose tensor holds the lists)
                if len(_last_rendered) > 64:
                    _last_rendered.clear()
                hist = _last_rendered.get(hkey)
                hist = (hist if isinstance(hist, list) else ([int(hist)] if hist else [])) + [rendered]
                _last_rendered[hkey] = hist[-RENDERED_WINDOW:]
                global LAST_BINNING_CAPACITY, LAST_NUM_RENDERED, HINT_MISSES
                if a.rendered_hint and rendered > int(a.rendered_hint):
                    HINT_MISSES += 1
                LAST_NUM_RENDERED = rendered
                LAST_BINNING_CAPACITY = max(rendered, int(a.rendered_hint)) if a.rendered_hint else rendered
                binningBuffer = holder.get("t", binningBuffer)
        return (rendered, out_color, radii, out_normal, out_depth, out_cam_feat, out_warped, out_min_depth_diff,
                out_camera_ray, out_mask, geomBuffer, binningBuffer, imgBuffer)

    @staticmethod
    def rasterize_gaussians_backward(background, normal_map_pixels, intersected_depth_pixels, warped_image_pixels,
                                     means3D, radii, colors, all_maps, scales, rotations, scale_modifier,
                                     cov3D_precomp, viewmatrix, projmatrix, ref_to_src_list, src_cam_pos,
                                     src_images, src_rendered_depths, nb_src_images, tan_fovx, tan_fovy,
                                     dL_dout_color, dL_dout_normal_map, dL_dout_median_intersected_depth,
                                     dL_dout_warped_image, sh, degree, campos, geomBuffer, R, binningBuffer,
                                     imageBuffer, render_geo, debug, plane=None, packed_tex=None, skip_unused=False, buffer_length=0, want_abs=True, sh_rest=None):
        """The reference's 34 positional arguments and 10 results; with `plane` (see rasterize_gaussians) two more
        results follow: dL/d raw normal (P, 3) and dL/d raw offset (P, 1).  `buffer_length` (the forward's; not among the reference's
        arguments, 0 = unknown) only sizes the geo backward's scratch table."""
        lib = _lib.load()
        device = means3D.device
        P = int(means3D.size(0))
        H = int(normal_map_pixels.size(1)); W = int(normal_map_pixels.size(2))
        render_geo = bool(render_geo)
        with _on_device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            sh_c = _dev_f32(sh, device)
            sh_rest_c = _dev_f32(sh_rest, device) if sh_rest is not None else None
            M_dc = int(sh.size(1)) if sh.dim() == 3 else 0     # keeps (0, M, 3) for P == 0 so autograd accepts the shape
            M = M_dc + (int(sh_rest.size(1)) if sh_rest is not None else 0)          # (sh_rest: `sh` is the DC coefficient alone, see rasterize_gaussians)
            opts = dict(dtype=torch.float32, device=device)
            # ibgs_backward overwrites every element of its outputs (zeros for invisible Gaussians): no memsets
            have_sr = scales is not None and scales.numel() != 0
            new = torch.empty if P != 0 else torch.zeros
            dL_dmeans3D = _sink_or_new("means3D", (P, 3), new, opts) if P != 0 else new(P, 3, **opts); dL_dmeans2D = new(P, 3, **opts)
            # want_abs = False (the autograd node: means2D_abs needs no gradient -- after densify_until_iter, at test time): IBGS_FLAG_NO_ABS_GRAD
            dL_dmeans2D_abs = new(P, 3, **opts) if want_abs else None
            # gradients of inputs the mode does not use: zeros without a fill (the reference memsets them, rasterize_points.cu:196-206)
            fused = plane is not None and bool(plane[2])
            dL_dall_map = new(P, NUM_PLANE_PARAMS, **opts) if (render_geo and all_maps.numel() != 0 and P != 0 and not fused) else _zeros_view((P, NUM_PLANE_PARAMS), device)
            dL_dplane_normal = dL_dplane_offset = None
            if fused:
                learnt = int(plane[2]) == _lib.PLANE_LEARNT
                want = render_geo and P != 0
                dL_dplane_normal = (new(P, 3, **opts) if want else _zeros_view((P, 3), device)) if (learnt and plane[0] is not None) else None
                dL_dplane_offset = (new(P, 1, **opts) if want else _zeros_view((P, 1), device)) if (learnt and plane[1] is not None) else None
            dL_dopacity = _sink_or_new("opacities", (P, 1), new, opts) if P != 0 else new(P, 1, **opts)
            factored = _sh_factor_sink is not None and M != 0 and P != 0
            # skip_unused (the autograd node sets it): dL/dcolors and dL/dcov3D are the gradients of colors_precomp / cov3D_precomp -- with
            # SH coefficients and scales + rotations as inputs nobody reads them (the reference writes them regardless, 36 B per Gaussian)
            want_colors = not skip_unused or factored or M == 0 or (colors is not None and colors.numel() != 0)
            want_cov = not skip_unused or not have_sr
            dL_dcolors = new(P, NUM_CHANNELS, **opts) if want_colors else _zeros_view((P, NUM_CHANNELS), device)
            dL_dcov3D = new(P, 6, **opts) if want_cov else _zeros_view((P, 6), device)
            dL_dsh = None if factored else new(P, M_dc, 3, **opts)
            dL_dsh_rest = None if (factored or sh_rest is None) else new(P, M - M_dc, 3, **opts)
            dL_dscales = _sink_or_new("scales", (P, 3), new, opts) if (have_sr and P != 0) else _zeros_view((P, 3), device)
            dL_drotations = _sink_or_new("rotations", (P, 4), new, opts) if (have_sr and P != 0) else _zeros_view((P, 4), device)
            if P != 0:
                means3D_c = _dev_f32(means3D, device); colors_c = _dev_f32(colors, device)
                scales_c = _dev_f32(scales, device); rot_c = _dev_f32(rotations, device)
                cov_c = _dev_f32(cov3D_precomp, device); all_map_c = _dev_f32(all_maps, device)
                bg_c = _dev_f32(background, device); vm_c = _dev_f32(viewmatrix, device); pm_c = _dev_f32(projmatrix, device)
                campos_c = _dev_f32(campos, device)
                r2s_c = _dev_f32(ref_to_src_list, device); scp_c = _dev_f32(src_cam_pos, device)
                simg_c = _dev_f32(src_images, device); sdep_c = _dev_f32(src_rendered_depths, device)
                g_color = _dev_f32(dL_dout_color, device)
                g_normal = _dev_f32(dL_dout_normal_map, device) if render_geo else None
                g_depth = _dev_f32(dL_dout_median_intersected_depth, device) if render_geo else None
                g_warp = _dev_f32(dL_dout_warped_image, device) if render_geo else None
                depth_c = _dev_f32(intersected_depth_pixels, device) if render_geo else None
                warped_c = _dev_f32(warped_image_pixels, device) if render_geo else None
                gacc_ent = _gacc(device, P)
                grad_acc = gacc_ent[0]
                radii_c = radii.contiguous()
                a = _call_state.backward_args()
                a.stream = stream
                a.P, a.D, a.M, a.W, a.H = P, int(degree), M, W, H
                a.R = int(R)
                a.means3D = _ptr(means3D_c); a.shs = _ptr(sh_c); a.colors_precomp = _ptr(colors_c); a.shs_rest = _ptr(sh_rest_c)
                a.scales = _ptr(scales_c); a.rotations = _ptr(rot_c); a.cov3D_precomp = _ptr(cov_c); a.all_map = _ptr(all_map_c)
                a.scale_modifier = float(scale_modifier)
                a.bg = _ptr(bg_c); a.viewmatrix = _ptr(vm_c); a.projmatrix = _ptr(pm_c); a.campos = _ptr(campos_c)
                a.tanfovx = float(tan_fovx); a.tanfovy = float(tan_fovy)
                a.n_src = int(nb_src_images)
                a.ref_to_src = _ptr(r2s_c); a.src_cam_pos = _ptr(scp_c); a.src_images = _ptr(simg_c); a.src_depths = _ptr(sdep_c)
                a.radii = radii_c.data_ptr()
                a.out_depth = _ptr(depth_c); a.out_warped = _ptr(warped_c)
                a.geom = geomBuffer.data_ptr(); a.binning = binningBuffer.data_ptr() if binningBuffer.numel() else None
                a.img = imageBuffer.data_ptr()
                tex_flag = 0
                if render_geo:
                    nbytes = lib.ibgs_required_tex(int(nb_src_images), W, H)
                    tex = _tex_still(packed_tex, device, nbytes)
                    if tex is not None:
                        tex_flag = _lib.FLAG_TEX_PACKED          # still the forward's pack of the same images: no second pack (T1)
                    else:
                        tex, _ = _tex_packed(device, nbytes)
                    a.tex = tex.data_ptr(); a.tex_bytes = tex.numel()
                    a.buffer_length = int(buffer_length)
                    tab = _tex(device, lib.ibgs_required_geo_table_for(W, H, int(buffer_length)), "geo_table")
                    a.geo_table = tab.data_ptr(); a.geo_table_bytes = tab.numel()
                a.dL_dcolor = _ptr(g_color); a.dL_dnormal = _ptr(g_normal); a.dL_ddepth = _ptr(g_depth); a.dL_dwarped = _ptr(g_warp)
                a.grad_acc = grad_acc.data_ptr()
                a.dL_dmean2D = dL_dmeans2D.data_ptr(); a.dL_dmean2D_abs = dL_dmeans2D_abs.data_ptr() if want_abs else None
                a.dL_dconic = None
                a.dL_dopacity = dL_dopacity.data_ptr(); a.dL_dcolors = dL_dcolors.data_ptr() if want_colors else None
                a.dL_dmean3D = dL_dmeans3D.data_ptr(); a.dL_dcov3D = dL_dcov3D.data_ptr() if want_cov else None
                a.dL_dsh = dL_dsh.data_ptr() if (M and not factored) else None
                a.dL_dsh_rest = dL_dsh_rest.data_ptr() if dL_dsh_rest is not None else None
                a.dL_dscale = dL_dscales.data_ptr() if have_sr else None; a.dL_drot = dL_drotations.data_ptr() if have_sr else None
                a.dL_dall_map = dL_dall_map.data_ptr() if (render_geo and all_maps.numel() != 0 and not fused) else None
                if fused and render_geo:
                    pn_c, po_c = _dev_f32(plane[0], device), _dev_f32(plane[1], device)
                    a.plane_normal = _ptr(pn_c); a.plane_offset = _ptr(po_c); a.plane_mode = int(plane[2]); a.all_map = None
                    a.dL_dplane_normal = _ptr(dL_dplane_normal); a.dL_dplane_offset = _ptr(dL_dplane_offset)
                a.render_geo = int(render_geo)
                a.flags = ((_lib.FLAG_DEBUG if debug else 0) | (_lib.FLAG_TEX_QUANT if TEX_QUANT else 0)
                           | _lib.FLAG_CLEAR_GRAD_ACC | (_lib.FLAG_SH_FACTORED if factored else 0) | _shape_flag() | tex_flag
                           | (0 if want_abs else _lib.FLAG_NO_ABS_GRAD))
                if DETERMINISTIC and int(R) > 0:
                    det = torch.empty(lib.ibgs_required_deterministic_for(int(R), P, W, H, int(render_geo), int(a.flags)), dtype=torch.uint8, device=device)
                    if KEEP_DET_SCRATCH:
                        _CModule.last_det = det          # tools/pairing_stats.py reads the slab: which (list entry, wave) rows the backward wrote
                    a.det_scratch = det.data_ptr(); a.det_scratch_bytes = det.numel()
                    a.flags |= _lib.FLAG_DETERMINISTIC
                if ORDER_HINT and (ORDER_HINT_GEO or not render_geo) and not debug:
                    # the order this backward launches its tiles in (render_bwd.hip) = the hint for this camera's next forward, written straight into
                    # the camera's buffer.  (Small frames build no order: the buffer keeps what it had, which the forward checks before use.)
                    ckey = _camera_key(viewmatrix, device, W, H, render_geo, stream)
                    if ckey is not None:
                        oh = _order_hints.get(ckey)
                        if oh is None:
                            while len(_order_hints) >= ORDER_HINT_MAX:
                                _order_hints.popitem(last=False)          # (a buffer still referenced by an enqueued kernel stays alive in the stream's allocator until it ran)
                            oh = _order_hints[ckey] = torch.full((int(lib.ibgs_tile_order_slots(W, H)),), -1, dtype=torch.int32, device=device)
                        else:
                            _order_hints.move_to_end(ckey)
                        a.tile_order_out = oh.data_ptr()
                rc = lib.ibgs_backward(ctypes.byref(a))
                if rc < 0:
                    raise RuntimeError("ibgs_backward failed (%d): %s" % (rc, _lib.last_error()))
                gacc_ent[1] = False
                if factored:
                    _sh_factor_sink.append({"dcolor": dL_dcolors, "campos": campos_c.reshape(3), "degree": int(degree), "M": M})
        res = (dL_dmeans2D, dL_dmeans2D_abs, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh,
               dL_dscales, dL_drotations, dL_dall_map)
        if plane is not None or sh_rest is not None:          # the extensions' results follow the reference's ten: plane normal, plane offset, SH rest
            res = res + (dL_dplane_normal, dL_dplane_offset)
            if sh_rest is not None:
                res = res + (dL_dsh_rest,)
        return res

    @staticmethod
    def mark_visible(means3D, viewmatrix, projmatrix):
        lib = _lib.load()
        device = means3D.device
        P = int(means3D.size(0))
        present = torch.zeros(P, dtype=torch.bool, device=device)
        if P != 0:
            with torch.cuda.device(device):
                m = _dev_f32(means3D, device); vm = _dev_f32(viewmatrix, device); pm = _dev_f32(projmatrix, device)
                rc = lib.ibgs_mark_visible(torch.cuda.current_stream(device).cuda_stream, P, m.data_ptr(),
                                           vm.data_ptr(), _ptr(pm), present.data_ptr())
                if rc < 0:
                    raise RuntimeError("ibgs_mark_visible failed (%d): %s" % (rc, _lib.last_error()))
        return present


_C = _CModule()


def check_async_errors(device=None, wait=True):
    """The library's one asynchronous error -- a depth sort whose look-back timed out in a forward that did not wait for it (include/ibgs_rast.h:
    ibgs_check_async) -- for callers no backward follows: evaluation renders, the last forward of a run.  (A training step needs nothing: its own backward
    reports the error of its forward.)  wait=True drains the device's current stream first; raises RuntimeError when a forward on it produced mis-ordered lists."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(device):
        rc = _lib.load().ibgs_check_async(torch.cuda.current_stream(device).cuda_stream, 1 if wait else 0)
    if rc < 0:
        raise RuntimeError("ibgs_check_async failed (%d): %s" % (rc, _lib.last_error()))


def rasterize_gaussians(means3D, means2D, means2D_abs, sh, colors_precomp, opacities, scales, rotations,
                        cov3Ds_precomp, all_map, raster_settings, plane_normal=None, plane_offset=None, plane_mode=0, sh_rest=None):
    return _RasterizeGaussians.apply(means3D, means2D, means2D_abs, sh, colors_precomp, opacities, scales,
                                     rotations, cov3Ds_precomp, all_map, raster_settings, plane_normal, plane_offset, plane_mode, sh_rest)


def rasterize_depth_batch(means3D, opacities, scales, rotations, cov3D_precomp, scale_modifier, viewmatrices, projmatrices,
                          camposes, tanfovxs, tanfovys, image_height, image_width, buffer_length, plane_normal=None,
                          plane_offset=None, plane_mode=2, debug=False):
    """median ray/plane depth of n cameras of equal size in ONE rasterizer pass (C ABI
    `ibgs_forward_args.n_views`).  Replaces the reference's loop of `render_depth` calls over the source views
    (gaussian_renderer/__init__.py:245-253); every view's map is bit-identical to its single pass.
    viewmatrices / projmatrices: (n, 4, 4) as `world_view_transform` / `full_proj_transform`; camposes (n, 3).
    Returns (depths (n, 1, H, W), radii (n, P)).  Forward only (the reference runs these passes without gradients)."""
    lib = _lib.load()
    device = means3D.device
    if not means3D.is_cuda:
        raise RuntimeError("means3D must live on a HIP device (libibgs_rast.so has no CPU path)")
    n = int(viewmatrices.shape[0])
    if not (1 <= n <= _lib.MAX_VIEWS):
        raise RuntimeError("rasterize_depth_batch: 1..%d views per call" % _lib.MAX_VIEWS)
    if not plane_mode:
        raise RuntimeError("rasterize_depth_batch needs plane_mode 1 (learnt normal) or 2 (smallest axis)")
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    with torch.cuda.device(device), torch.no_grad():
        depths = (torch.empty if P else torch.zeros)(n, 1, H, W, dtype=torch.float32, device=device)
        radii = (torch.empty if P else torch.zeros)(n, P, dtype=torch.int32, device=device)
        if P == 0:
            return depths, radii
        m_c = _dev_f32(means3D, device); o_c = _dev_f32(opacities, device)
        s_c = _dev_f32(scales, device); r_c = _dev_f32(rotations, device); cov_c = _dev_f32(cov3D_precomp, device)
        vm_c = _dev_f32(viewmatrices.reshape(n, 16), device); pm_c = _dev_f32(projmatrices.reshape(n, 16), device)
        cp_c = _dev_f32(camposes.reshape(n, 3), device)
        pn_c = _dev_f32(plane_normal, device); po_c = _dev_f32(plane_offset, device)
        bg_c = _zeros_view((3,), device).contiguous()
        gy = (H + 15) // 16
        geom = torch.empty(lib.ibgs_required_geom(n * P), dtype=torch.uint8, device=device)
        img = torch.empty(lib.ibgs_required_img(W, n * 16 * gy if n > 1 else H), dtype=torch.uint8, device=device)
        holder = {}

        def _alloc(nbytes, _user):
            try:
                holder["t"] = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
                return holder["t"].data_ptr()
            except Exception as ex:
                holder["err"] = ex
                return 0

        cb = _lib.ALLOC_FN(_alloc)
        a = _lib.ForwardArgs()
        a.stream = torch.cuda.current_stream(device).cuda_stream
        a.P, a.D, a.M, a.W, a.H = P, 0, 0, W, H
        a.means3D = _ptr(m_c); a.opacities = _ptr(o_c); a.scales = _ptr(s_c); a.rotations = _ptr(r_c); a.cov3D_precomp = _ptr(cov_c)
        a.scale_modifier = float(scale_modifier)
        a.bg = _ptr(bg_c); a.viewmatrix = _ptr(vm_c); a.projmatrix = _ptr(pm_c); a.campos = _ptr(cp_c)
        a.tanfovx = float(tanfovxs[0]); a.tanfovy = float(tanfovys[0])
        a.n_src = 1; a.buffer_length = int(buffer_length); a.depth_error_threshold = 0.0
        a.render_geo = 0; a.render_depth_only = 1
        a.flags = (_lib.FLAG_DEBUG if debug else 0) | (0 if TILE_CULL else _lib.FLAG_NO_TILE_CULL) | (0 if REF_POWER_SKIP else _lib.FLAG_NO_REF_POWER_SKIP)
        a.geom = geom.data_ptr(); a.geom_bytes = geom.numel(); a.img = img.data_ptr(); a.img_bytes = img.numel()
        a.binning_alloc = cb; a.binning_user = None
        a.radii = radii.data_ptr(); a.out_depth = depths.data_ptr()
        a.plane_normal = _ptr(pn_c); a.plane_offset = _ptr(po_c); a.plane_mode = int(plane_mode)
        a.n_views = n if n > 1 else 0
        for v in range(n):
            a.view_tanfovx[v] = float(tanfovxs[v]); a.view_tanfovy[v] = float(tanfovys[v])
        hkey = (device.index, P, W, H, "depth_batch", n)
        hist = _last_rendered.get(hkey) if (RENDERED_HINT and not debug) else None
        prev = max(hist) if hist else 0
        a.rendered_hint = (prev + prev // 4 + 4096) if prev > 0 else 0
        rc = lib.ibgs_forward(ctypes.byref(a))
        if rc < 0:
            if "err" in holder:
                raise holder["err"]
            raise RuntimeError("ibgs_forward (depth batch) failed (%d): %s" % (rc, _lib.last_error()))
        _last_rendered[hkey] = ((hist or []) + [int(rc)])[-RENDERED_WINDOW:]
    return depths, radii


_EMPTY = torch.Tensor([])          # never written, never returned: the stand-in for "not provided"


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, means2D_abs, sh, colors_precomp, opacities, scales, rotations,
                cov3Ds_precomp, all_maps, raster_settings, plane_normal=None, plane_offset=None, plane_mode=0, sh_rest=None):
        plane = (plane_normal, plane_offset, int(plane_mode)) if plane_mode else None
        kw = {"plane": plane} if plane is not None else {}
        if sh_rest is not None:
            kw["sh_rest"] = sh_rest
        if raster_settings.render_geo and getattr(raster_settings, "src_depth_slots", None) is not None:
            kw["depth_slots"] = raster_settings.src_depth_slots
        # argument order of the reference's _C.rasterize_gaussians (reference __init__.py:66-98)
        args = (
            raster_settings.bg, means3D, colors_precomp, opacities, scales, rotations,
            raster_settings.scale_modifier, cov3Ds_precomp, all_maps,
            raster_settings.viewmatrix, raster_settings.projmatrix,
            raster_settings.ref_to_src_list, raster_settings.src_cam_pos, raster_settings.src_images,
            raster_settings.src_rendered_depths, raster_settings.nb_src_images,
            raster_settings.buffer_length, raster_settings.depth_error_threshold,
            raster_settings.tanfovx, raster_settings.tanfovy,
            raster_settings.image_height, raster_settings.image_width,
            sh, raster_settings.sh_degree, raster_settings.campos, raster_settings.prefiltered,
            raster_settings.render_geo, raster_settings.render_depth_only, raster_settings.debug,
        )
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)  # copy them before they can be corrupted
            try:
                res = _C.rasterize_gaussians(*args, **kw)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            res = _C.rasterize_gaussians(*args, **kw)
        (num_rendered, color, radii, out_normal_map, out_median_intersected_depth, out_cam_feat, out_warped_image,
         out_min_depth_diff, out_camera_ray, out_use_first_src_frame, geomBuffer, binningBuffer, imgBuffer) = res

        ctx.raster_settings = raster_settings
        ctx.packed_tex, _CModule.last_tex = _CModule.last_tex, None
        ctx.plane_mode = int(plane_mode) if plane is not None else 0
        ctx.num_rendered = num_rendered
        # outputs that the loss does not touch arrive as None instead of freshly zero-filled planes
        ctx.set_materialize_grads(False)
        none = _EMPTY
        ctx.save_for_backward(out_normal_map, out_median_intersected_depth, out_warped_image, colors_precomp,
                              all_maps, means3D, scales, rotations, cov3Ds_precomp, radii, sh,
                              plane_normal if plane_normal is not None else none, plane_offset if plane_offset is not None else none,
                              sh_rest if sh_rest is not None else none,
                              geomBuffer, binningBuffer, imgBuffer)
        ctx.has_sh_rest = sh_rest is not None
        ctx.mark_non_differentiable(radii, out_use_first_src_frame)
        return (color, radii, out_normal_map, out_median_intersected_depth, out_cam_feat, out_warped_image,
                out_min_depth_diff, out_camera_ray, out_use_first_src_frame)

    @staticmethod
    def backward(ctx, grad_out_color, grad_radii, grad_out_normal_map, grad_out_median_intersected_depth,
                 grad_out_cam_feat, grad_out_warped_image, grad_out_min_depth_diff, grad_out_camera_ray,
                 grad_out_use_first_src_frame):
        num_rendered = ctx.num_rendered
        raster_settings = ctx.raster_settings
        (normal_map_pixels, median_intersected_depth_pixels, warped_image_pixels, colors_precomp, all_maps, means3D,
         scales, rotations, cov3Ds_precomp, radii, sh, plane_normal, plane_offset, sh_rest, geomBuffer, binningBuffer, imgBuffer) = ctx.saved_tensors
        kw = {}
        if ctx.has_sh_rest:
            kw["sh_rest"] = sh_rest
        if ctx.plane_mode:
            kw["plane"] = (plane_normal if plane_normal.numel() else None, plane_offset if plane_offset.numel() else None, ctx.plane_mode)
        if getattr(ctx, "packed_tex", None) is not None:
            kw["packed_tex"] = ctx.packed_tex
        kw["skip_unused"] = True
        kw["buffer_length"] = int(raster_settings.buffer_length)
        kw["want_abs"] = bool(ctx.needs_input_grad[2]) or not NO_ABS_GRAD_WHEN_UNUSED

        # argument order of the reference's _C.rasterize_gaussians_backward (reference __init__.py:182-221)
        args = (raster_settings.bg, normal_map_pixels, median_intersected_depth_pixels, warped_image_pixels,
                means3D, radii, colors_precomp, all_maps, scales, rotations, raster_settings.scale_modifier,
                cov3Ds_precomp, raster_settings.viewmatrix, raster_settings.projmatrix,
                raster_settings.ref_to_src_list, raster_settings.src_cam_pos, raster_settings.src_images,
                raster_settings.src_rendered_depths, raster_settings.nb_src_images,
                raster_settings.tanfovx, raster_settings.tanfovy,
                grad_out_color, grad_out_normal_map, grad_out_median_intersected_depth, grad_out_warped_image,
                sh, raster_settings.sh_degree, raster_settings.campos, geomBuffer, num_rendered, binningBuffer,
                imgBuffer, raster_settings.render_geo, raster_settings.debug)
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                res = _C.rasterize_gaussians_backward(*args, **kw)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            res = _C.rasterize_gaussians_backward(*args, **kw)
        (grad_means2D, grad_means2D_abs, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp,
         grad_sh, grad_scales, grad_rotations, grad_all_map) = res[:10]
        grad_plane_normal, grad_plane_offset = (res[10], res[11]) if ctx.plane_mode else (None, None)
        grad_sh_rest = res[12] if ctx.has_sh_rest else None
        return (grad_means3D, grad_means2D, grad_means2D_abs, grad_sh, grad_colors_precomp, grad_opacities,
                grad_scales, grad_rotations, grad_cov3Ds_precomp, grad_all_map, None, grad_plane_normal, grad_plane_offset, None, grad_sh_rest)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    ref_to_src_list: torch.Tensor
    src_cam_pos: torch.Tensor
    src_images: torch.Tensor
    src_rendered_depths: torch.Tensor
    nb_src_images: int
    buffer_length: int
    depth_error_threshold: float
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    render_geo: bool
    render_depth_only: bool
    debug: bool
    # this library's extension (the reference's settings end with `debug`): `src_rendered_depths` is a TABLE of depth planes -- e.g. the trainer's depth cache
    # scene.rendered_depth_list, one plane per training camera -- and source m reads plane src_depth_slots[m] of it; what the reference obtains by indexing the
    # cache (a copy of n_src planes per call, gaussian_renderer/__init__.py:255).  None: src_rendered_depths holds exactly the n_src planes, in order.
    src_depth_slots: Optional[tuple] = None


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        # boolean mask of points passing the near-plane test of the camera (reference :283-292)
        with torch.no_grad():
            raster_settings = self.raster_settings
            visible = _C.mark_visible(positions, raster_settings.viewmatrix, raster_settings.projmatrix)
        return visible

    def forward(self, means3D, means2D, means2D_abs, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None, all_map=None, plane_normal=None, plane_offset=None, plane_mode=0, shs_rest=None):
        """Reference signature plus the fused plane-map extension: instead of `all_map` pass the raw `_normal` /
        `_offset` parameters with plane_mode=1 (learnt normals) or plane_mode=2 (normal = smallest-scale axis) -- and `shs_rest`: the SH
        coefficients as the model's two arrays, `shs` = `_features_dc` (P, 1, 3) and `shs_rest` = `_features_rest` (P, M - 1, 3), instead of their torch.cat."""
        if shs_rest is not None and (shs is None or shs.dim() != 3 or int(shs.shape[1]) != 1 or shs_rest.dim() != 3):
            raise Exception('shs_rest needs shs = the DC coefficients (P, 1, 3)')
        raster_settings = self.raster_settings
        if plane_mode and all_map is not None:
            raise Exception('Please provide either all_map or plane_mode, not both!')

        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')

        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')

        # the reference's "empty CPU tensor = not provided" (DPR/.../__init__.py:304-316): ONE shared empty tensor instead of up to six fresh ones per call
        if shs is None:
            shs = _EMPTY
        if colors_precomp is None:
            colors_precomp = _EMPTY
        if scales is None:
            scales = _EMPTY
        if rotations is None:
            rotations = _EMPTY
        if cov3D_precomp is None:
            cov3D_precomp = _EMPTY
        if all_map is None:
            all_map = _EMPTY

        return rasterize_gaussians(means3D, means2D, means2D_abs, shs, colors_precomp, opacities, scales,
                                   rotations, cov3D_precomp, all_map, raster_settings, plane_normal, plane_offset, plane_mode, shs_rest)
