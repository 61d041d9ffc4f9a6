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
