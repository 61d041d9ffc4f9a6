"""Densification surgery of the reference trainer in one data-movement pass (C ABI `ibgs_compact_plan` /
`ibgs_compact_apply`, ibgs_amd/csrc/compact.hip).

The reference changes its point set with boolean indexing + `torch.cat`, tensor by tensor and optimiser state by optimiser state
(scene/gaussian_model.py: `_prune_optimizer` :377-395, `prune_points` :397-421, `cat_tensors_to_optimizer` :423-444,
`densification_postfix` :446-469).  `compact_append` does "keep the masked rows, then append" for ANY number of tensors that
share the leading dimension in one launch, and `prune_and_extend_optimizer` is the drop-in for the two optimiser routines: same
resulting Parameters / Adam state (bit for bit), same `{group name: new Parameter}` return value.

HIP only: raises when the library or a GPU tensor is missing (no torch fallback in the product path)."""
import ctypes

import torch

from . import _lib


def compact_append(tensors, keep_mask=None, appends=None):
    """tensors: list of 4-byte-element tensors (N, ...) on one HIP device; keep_mask: bool (N,) or None (keep all);
    appends: None (nothing appended), or a list with one entry per tensor: a tensor (n_app, ...) of matching trailing shape and
    dtype (rows are copied), an int n_app (that many ZERO rows: the Adam moments of new points), or None (= zero rows, count
    taken from the other entries).  Returns the list of new tensors (n_keep + n_app, ...): kept rows in their original order,
    then the appended rows."""
    if not tensors:
        return []
    lib = _lib.load()
    dev = tensors[0].device
    if not tensors[0].is_cuda:
        raise RuntimeError("compact_append runs on the MI355X only (no CPU path)")
    N = int(tensors[0].shape[0])
    appends = list(appends) if appends is not None else [None] * len(tensors)
    if len(appends) != len(tensors):
        raise ValueError("appends must have one entry per tensor")
    n_app = 0
    for a in appends:
        if isinstance(a, torch.Tensor):
            n_app = int(a.shape[0])
            break
        if isinstance(a, int):
            n_app = a
    srcs, apps = [], []
    for t, a in zip(tensors, appends):
        if t.device != dev or int(t.shape[0]) != N or t.element_size() != 4:
            raise ValueError("compact_append: tensors must share device and leading size and have 4-byte elements")
        srcs.append(t.detach().contiguous())
        if isinstance(a, torch.Tensor):
            if tuple(a.shape[1:]) != tuple(t.shape[1:]) or int(a.shape[0]) != n_app or a.dtype != t.dtype:
                raise ValueError("compact_append: appended rows must match the tensor's trailing shape, dtype and a common count")
            apps.append(a.detach().to(dev).contiguous())
        else:
            apps.append(None)
    if keep_mask is not None:
        if keep_mask.dtype != torch.bool or keep_mask.shape != (N,):
            raise ValueError("keep_mask must be a bool tensor of shape (N,)")
        keep_mask = keep_mask.to(dev).contiguous()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        if N == 0:
            return [(a.clone() if a is not None else torch.zeros((n_app,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)) for t, a in zip(srcs, apps)]
        scratch = torch.empty(lib.ibgs_required_compact(N), dtype=torch.uint8, device=dev)
        n_keep = lib.ibgs_compact_plan(stream, N, keep_mask.data_ptr() if keep_mask is not None else None, scratch.data_ptr(), scratch.numel())
        if n_keep < 0:
            raise RuntimeError("ibgs_compact_plan failed (%d): %s" % (n_keep, _lib.last_error()))
        outs = [torch.empty((int(n_keep) + n_app,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev) for t in srcs]
        if int(n_keep) + n_app == 0:
            return outs
        for i in range(0, len(srcs), _lib.COMPACT_MAX_TENSORS):
            part = []
            for t, a, o in zip(srcs[i:i + _lib.COMPACT_MAX_TENSORS], apps[i:i + _lib.COMPACT_MAX_TENSORS], outs[i:i + _lib.COMPACT_MAX_TENSORS]):
                d = _lib.CompactTensor()
                d.src, d.append, d.dst = t.data_ptr(), (a.data_ptr() if a is not None else None), o.data_ptr()
                d.width = max(1, t.numel() // N)
                part.append(d)
            arr = (_lib.CompactTensor * len(part))(*part)
            rc = lib.ibgs_compact_apply(stream, len(part), ctypes.cast(arr, ctypes.c_void_p), N, n_app, scratch.data_ptr())
            if rc < 0:
                raise RuntimeError("ibgs_compact_apply failed (%d): %s" % (rc, _lib.last_error()))
    return outs


def prune_and_extend_optimizer(optimizer, keep_mask=None, extension=None, extra=None):
    """`_prune_optimizer(keep_mask)` followed by `cat_tensors_to_optimizer(extension)` of the reference
    (scene/gaussian_model.py:377-395, 423-444) as ONE pass over every parameter group and its Adam moments.
    optimizer: torch.optim.Adam / FusedAdam whose groups hold one Parameter each and carry a "name";
    keep_mask: bool (N,) of rows to KEEP (None = all); extension: {group name: new rows} (None = nothing appended);
    extra: optional list of per-point tensors (gradient accumulators, radii ...) that are masked alike (appended rows = zeros).
    Returns ({group name: new Parameter}, [new extra tensors]) -- the reference's `optimizable_tensors`."""
    groups = optimizer.param_groups
    tensors, appends, slots = [], [], []
    n_app = 0
    if extension:
        n_app = int(next(iter(extension.values())).shape[0])
    for gi, g in enumerate(groups):
        assert len(g["params"]) == 1, "one Parameter per group, as in GaussianModel.training_setup"
        p = g["params"][0]
        ext = extension[g["name"]] if extension else None
        tensors.append(p.data); appends.append(ext if ext is not None else (n_app if n_app else None)); slots.append((gi, "param"))
        st = optimizer.state.get(p, None)
        if st is not None and "exp_avg" in st:
            for k in ("exp_avg", "exp_avg_sq"):
                tensors.append(st[k]); appends.append(n_app if n_app else None); slots.append((gi, k))
    for t in (extra or []):
        tensors.append(t); appends.append(n_app if n_app else None); slots.append((None, "extra"))
    outs = compact_append(tensors, keep_mask, appends)
    result, new_extra, new_state = {}, [], {}
    for (gi, kind), o in zip(slots, outs):
        if kind == "extra":
            new_extra.append(o)
        elif kind == "param":
            new_state[gi] = {"param": torch.nn.Parameter(o.requires_grad_(True))}
        else:
            new_state[gi][kind] = o
    for gi, g in enumerate(groups):
        old = g["params"][0]
        st = optimizer.state.pop(old, None)
        newp = new_state[gi]["param"]
        if st is not None:
            for k in ("exp_avg", "exp_avg_sq"):
                if k in new_state[gi]:
                    st[k] = new_state[gi][k]
            optimizer.state[newp] = st
        g["params"][0] = newp
        result[g["name"]] = newp
    return result, new_extra


# ---- the reference's densification POLICY on top of the fused pass ---------------------------------------------------------------------------
# scene/gaussian_model.py:471-522 (densify_and_split), :549-577 (densify_and_clone), :580-597 (densify_and_prune) select rows and build the new
# ones with plain torch ops; the point set then changes through `prune_and_extend_optimizer` (one pass per change instead of a torch.cat / boolean
# index per tensor and per Adam moment).  The selection functions below are device-agnostic torch code and reproduce the reference decision for
# decision, quirks included (tests/golden/densify.npz holds the reference's own masks, new rows and final state for three runs):
#   * every `densification_postfix` zeroes the per-point statistics INCLUDING max_radii2D (:463-468), and the split always calls it, so the
#     final prune's screen-size test `max_radii2D > max_screen_size` (:593) only ever sees zeros;
#   * the clone is skipped entirely -- no postfix, statistics untouched -- when nothing is selected (:562), the split is not (:505-519);
#   * the caps compare `selected + n` with `max_all_points` and re-select by a quantile of the masked gradients with a strict `>` (:487-503, :554-560).
class DensifyConfig:
    """The four `training_args` fields the policy reads (scene/gaussian_model.py:217-223; defaults of arguments/__init__.py:99-116)."""

    def __init__(self, percent_dense=0.001, abs_split_radii2D_threshold=20, max_abs_split_points=50_000, max_all_points=5_000_000):
        self.percent_dense, self.abs_split_radii2D_threshold = percent_dense, abs_split_radii2D_threshold
        self.max_abs_split_points, self.max_all_points = max_abs_split_points, max_all_points


STAT_NAMES = ("xyz_gradient_accum", "xyz_gradient_accum_abs", "denom", "denom_abs", "max_radii2D", "max_weight")


def _rotation_matrix(q):
    """utils/general_utils.py:81-102 (`build_rotation`): normalises the quaternion (w, x, y, z), then the usual matrix."""
    q = q / torch.sqrt((q * q).sum(dim=1, keepdim=True))
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1).view(-1, 3, 3)


def _named(optimizer):
    return {g["name"]: g["params"][0] for g in optimizer.param_groups}


def select_clone(p, grads, grad_threshold, scene_extent, cfg):
    """Rows `densify_and_clone` duplicates (:549-560): small Gaussians with a large mean screen-space gradient, capped by a quantile."""
    n = p["xyz"].shape[0]
    sel = torch.norm(grads, dim=-1) >= grad_threshold
    sel = torch.logical_and(sel, torch.max(torch.exp(p["scaling"]), dim=1).values <= cfg.percent_dense * scene_extent)
    if sel.sum() + n > cfg.max_all_points:
        g = grads.squeeze().clone()
        g[~sel] = 0
        sel = g > torch.quantile(g, 1.0 - (cfg.max_all_points - n) / float(n))
    return sel


def clone_rows(p, sel, sampler=torch.normal):
    """The appended rows of `densify_and_clone` (:563-575): raw copies, the position moved by one draw from the Gaussian itself."""
    stds = torch.exp(p["scaling"][sel])
    samples = sampler(mean=torch.zeros((stds.size(0), 3), device=stds.device), std=stds)
    ext = {k: v.detach()[sel] for k, v in p.items()}
    ext["xyz"] = torch.bmm(_rotation_matrix(p["rotation"].detach()[sel]), samples.unsqueeze(-1)).squeeze(-1) + p["xyz"].detach()[sel]
    return ext


def select_split(p, grads, grad_threshold, grads_abs, grad_abs_threshold, scene_extent, max_radii2D, cfg):
    """Rows `densify_and_split` replaces by N children (:471-503); `grads`, `grads_abs`, `max_radii2D` may be shorter than the point set
    (rows appended by the clone come after them and count as zero)."""
    n = p["xyz"].shape[0]
    dev = p["xyz"].device
    pad = lambda t: torch.cat((t.reshape(-1), torch.zeros(n - t.numel(), device=dev)))
    g, ga, mr = pad(grads), pad(grads_abs), pad(max_radii2D)
    big = torch.max(torch.exp(p["scaling"]), dim=1).values > cfg.percent_dense * scene_extent
    sel = torch.logical_and(g >= grad_threshold, big)
    if sel.sum() + n > cfg.max_all_points:
        g[~sel] = 0
        sel = g > torch.quantile(g, 1.0 - (cfg.max_all_points - n) / float(n))
    else:
        ga[sel] = 0
        ga[~(big & (mr > cfg.abs_split_radii2D_threshold))] = 0
        sel_abs = ga >= grad_abs_threshold
        limited = min(cfg.max_all_points - n - sel.sum(), cfg.max_abs_split_points)
        if sel_abs.sum() > limited:
            sel_abs = ga > torch.quantile(ga, 1.0 - limited / float(n))
        sel = torch.logical_or(sel, sel_abs)
    return sel


def split_rows(p, sel, sampler=torch.normal, N=2):
    """The N children per selected row (:505-518): positions drawn from the parent, scales divided by 0.8 N, everything else copied."""
    stds = torch.exp(p["scaling"].detach()[sel]).repeat(N, 1)
    samples = sampler(mean=torch.zeros((stds.size(0), 3), device=stds.device), std=stds)
    rots = _rotation_matrix(p["rotation"].detach()[sel]).repeat(N, 1, 1)
    ext = {k: v.detach()[sel].repeat(*([N] + [1] * (v.dim() - 1))) for k, v in p.items()}
    ext["xyz"] = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + p["xyz"].detach()[sel].repeat(N, 1)
    ext["scaling"] = torch.log(stds / (0.8 * N))
    return ext


def select_prune(p, max_radii2D, min_opacity, extent, max_screen_size):
    """Rows the last step of `densify_and_prune` removes (:589-596): transparent, or too large on screen / in the world."""
    mask = (torch.sigmoid(p["opacity"]) < min_opacity).squeeze()
    if max_screen_size:
        mask = torch.logical_or(torch.logical_or(mask, max_radii2D > max_screen_size), torch.exp(p["scaling"]).max(dim=1).values > 0.1 * extent)
    return mask


def densify_and_prune(optimizer, stats, max_grad, abs_max_grad, min_opacity, extent, max_screen_size, cfg=None, sampler=torch.normal,
                      surgery=None):
    """`GaussianModel.densify_and_prune` (scene/gaussian_model.py:580-597) over an optimiser with the reference's eight named groups.
    stats: dict with STAT_NAMES (accumulators (n, 1), max_radii2D / max_weight (n,)).  Three data-movement passes (clone = append; split =
    append the children AND drop the parents in one pass; prune) through `surgery` (default: `prune_and_extend_optimizer`, HIP).
    Returns ({group name: new Parameter}, new stats dict)."""
    cfg = cfg or DensifyConfig()
    surgery = surgery or prune_and_extend_optimizer
    with torch.no_grad():
        grads = stats["xyz_gradient_accum"] / stats["denom"]
        grads_abs = stats["xyz_gradient_accum_abs"] / stats["denom_abs"]
        grads[grads.isnan()] = 0.0
        grads_abs[grads_abs.isnan()] = 0.0
        max_radii2D = stats["max_radii2D"].clone()
        stats = dict(stats)

        def fresh(n):          # densification_postfix (:463-468)
            dev = grads.device
            return {k: torch.zeros((n, 1) if k in STAT_NAMES[:4] else (n,), device=dev) for k in STAT_NAMES}

        p = _named(optimizer)
        sel = select_clone(p, grads, max_grad, extent, cfg)
        if sel.sum() > 0:
            p, _ = surgery(optimizer, None, clone_rows(p, sel, sampler))
            stats = fresh(p["xyz"].shape[0])
        sel = select_split(p, grads, max_grad, grads_abs, abs_max_grad, extent, max_radii2D, cfg)
        p, _ = surgery(optimizer, ~sel, split_rows(p, sel, sampler))
        stats = fresh(p["xyz"].shape[0])
        mask = select_prune(p, stats["max_radii2D"], min_opacity, extent, max_screen_size)
        p, extra = surgery(optimizer, ~mask, None, extra=[stats[k] for k in STAT_NAMES])
        return p, dict(zip(STAT_NAMES, extra))


def add_densification_stats(stats, viewspace_points, viewspace_points_abs, radii):
    """`gaussians.max_radii2D[vis] = max(...)` + `GaussianModel.add_densification_stats(viewspace_point_tensor, viewspace_point_tensor_abs, visibility_filter)`
    (train.py:400-405, scene/gaussian_model.py:600-604) for one view, in ONE launch and without the reference's five boolean-indexed updates (each a
    nonzero + gather + scatter and a host sync).  stats: dict with STAT_NAMES tensors (updated IN PLACE); viewspace_points*: the sinks `render()` returned
    (their `.grad` is read; `viewspace_points_abs` may be None or carry no gradient: its statistics are then left alone); radii: `render()`'s int32 radii
    (visibility = radii > 0)."""
    lib = _lib.load()
    g = viewspace_points.grad if torch.is_tensor(viewspace_points) else None
    ga = viewspace_points_abs.grad if (torch.is_tensor(viewspace_points_abs) and viewspace_points_abs.grad is not None) else None
    if g is None:
        raise RuntimeError("add_densification_stats: viewspace_points has no .grad (call it after loss.backward())")
    if not radii.is_cuda:
        raise RuntimeError("add_densification_stats runs on the MI355X only (no CPU path)")
    P = int(radii.shape[0])
    r = radii if radii.dtype == torch.int32 else radii.to(torch.int32)
    ptr = lambda t: None if t is None else t.data_ptr()
    for k in STAT_NAMES[:5]:
        t = stats.get(k)
        if t is not None and not (t.is_contiguous() and t.dtype == torch.float32 and t.numel() == P):
            raise ValueError("add_densification_stats: stats[%r] must be a contiguous float32 tensor with one value per Gaussian" % k)
    with torch.cuda.device(radii.device):
        rc = lib.ibgs_densify_stats(torch.cuda.current_stream(radii.device).cuda_stream, P, r.contiguous().data_ptr(), g.contiguous().data_ptr(),
                                    None if ga is None else ga.contiguous().data_ptr(), ptr(stats.get("xyz_gradient_accum")),
                                    ptr(stats.get("xyz_gradient_accum_abs")) if ga is not None else None, ptr(stats.get("denom")),
                                    ptr(stats.get("denom_abs")) if ga is not None else None, ptr(stats.get("max_radii2D")))
    if rc < 0:
        raise RuntimeError("ibgs_densify_stats failed (%d): %s" % (rc, _lib.last_error()))
    return stats
