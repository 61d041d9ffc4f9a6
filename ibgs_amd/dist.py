"""View-parallel training step: one process per GPU, one training view per rank, Gaussian
parameters replicated, gradients summed with ONE all-reduce per step (RCCL over xGMI on MI355X;
``gloo`` on CPU for the tests).

The reference has no distributed layer at all -- it renders exactly one view per iteration
(train.py:275-292).  The semantics implemented here: the all-reduced
gradient equals the SUM of the single-view gradients that the reference would accumulate by
running those views sequentially without an optimizer step in between.  Densification statistics
are per-view non-linear (scene/gaussian_model.py:600-604), so their norms are computed locally
and only then reduced.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns (rank, world_size, local_rank).  A single process needs no initialisation."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def views_for_rank(step, rank, world_size, n_views):
    """Rank r renders view (world_size * step + r) mod n_views."""
    return (world_size * step + rank) % n_views


class GradBucket:
    """A persistent flat fp32 buffer that the per-tensor gradients are packed into so that the step
    issues a single large all-reduce (xGMI is point-to-point; one 200+ MB collective keeps all seven
    links busy, many small ones are latency-bound)."""

    def __init__(self, tensors):
        self.shapes = [t.shape for t in tensors]
        self.numels = [t.numel() for t in tensors]
        total = sum(self.numels)
        ref = tensors[0]
        self.flat = torch.zeros(total, dtype=torch.float32, device=ref.device)

    def pack(self, grads):
        off = 0
        for g, n in zip(grads, self.numels):
            if g is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(g.reshape(-1))
            off += n
        return self.flat

    def unpack(self):
        out, off = [], 0
        for shp, n in zip(self.shapes, self.numels):
            out.append(self.flat[off:off + n].view(shp))
            off += n
        return out


def allreduce_gradients(params, bucket=None, group=None, average=False):
    """Sum (or average) ``p.grad`` of every tensor in ``params`` over all ranks, in place.
    Returns the bucket so that callers can reuse it across steps."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return bucket
    if bucket is None:
        bucket = GradBucket(params)
    flat = bucket.pack([p.grad for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat.div_(dist.get_world_size(group))
    for p, g in zip(params, bucket.unpack()):
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
    return bucket


class ViewParallelReducer:
    """Gradient exchange of one view-parallel step, sized for xGMI (point-to-point links: bytes per rank are
    what costs).  For SH degree 3 the SH coefficients are 48 of the 59 gradient floats per Gaussian, but the
    dL/dsh of ONE view is rank one -- basis(direction to that camera) x dL/dRGB (reference backward.cu:114-160)
    -- so ranks all-gather the 3-float dL/dRGB of their views (plus the camera centres) and every rank rebuilds
    the summed dL/dsh locally (`ibgs_sh_grad_from_views`); only the remaining parameters go through the flat
    all-reduce.  Per rank at P = 1M, 8 ranks: ~77 MB + 84 MB on the links instead of ~413 MB.

        red = ViewParallelReducer(params, sh=shs_param, means3D=xyz_param)
        with red.capture():            # one or more backward passes (views) per rank
            loss.backward()
        red.reduce()                   # every p.grad now holds the sum over all ranks' views

    The result equals the sequential accumulation of the single-view gradients (summation order differs)."""

    def __init__(self, params, sh=None, means3D=None, group=None, expand=None):
        # `sh`: the SH tensor handed to the rasterizer, or -- when that is a torch.cat of leaves along dim 1 as in the
        # reference's GaussianModel.get_features (scene/gaussian_model.py:140-143) -- the list of those leaves
        self.sh_parts = list(sh) if isinstance(sh, (list, tuple)) else ([sh] if sh is not None else [])
        self.sh, self.means3D, self.group = (self.sh_parts[0] if self.sh_parts else None), means3D, group
        self.dense = [p for p in params if not any(p is q for q in self.sh_parts)]
        self.bucket = None
        self.items = None
        self._expand = expand

    def capture(self):
        from . import rasterizer
        red = self

        class _Ctx(rasterizer.capture_sh_factors):
            def __enter__(self):
                red.items = super().__enter__()
                return red.items
        return _Ctx()

    def reduce(self, average=False):
        world = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        if world > 1:
            self.bucket = allreduce_gradients(self.dense, self.bucket, self.group, average)
        items, self.items = self.items or [], None
        if self.sh is None or not items:
            return
        expand = self._expand
        if expand is None:
            from .shgrad import sh_grad_from_views as expand
        degree, M = items[0]["degree"], items[0]["M"]
        n_local, P = len(items), int(items[0]["dcolor"].shape[0])
        # one buffer per exchange: each view's (P, 3) factor followed by its camera centre -> ONE all-gather
        buf = torch.empty(n_local, P + 1, 3, dtype=torch.float32, device=items[0]["dcolor"].device)
        for i, it in enumerate(items):
            buf[i, :P] = it["dcolor"]
            buf[i, P] = it["campos"].to(buf.device)
        if world > 1:
            allb = torch.empty(world * n_local, P + 1, 3, dtype=buf.dtype, device=buf.device)
            dist.all_gather_into_tensor(allb, buf, group=self.group)
            buf = allb
        dcolor, campos = buf[:, :P], buf[:, P].contiguous()      # dcolor keeps the (P + 1) * 3 view stride
        g = expand(self.means3D.detach(), campos, dcolor, degree, M)
        if average:
            g = g / world
        off = 0
        for part in self.sh_parts:                      # (P, M_i, 3) slices of the (P, M, 3) result
            m = part.shape[1]
            gp = g[:, off:off + m, :].contiguous() if len(self.sh_parts) > 1 else g.view_as(part)
            part.grad = gp if part.grad is None else part.grad + gp
            off += m


def allreduce_densification_stats(viewspace_grad, viewspace_grad_abs, radii, group=None):
    """Per-view statistics consumed by GaussianModel.add_densification_stats
    (scene/gaussian_model.py:600-604; train.py:400-410), reduced over the views of this step:
    returns (sum of ||grad[:, :2]||, sum of ||grad_abs[:, :2]||, visible count, max radii)."""
    vis = radii > 0
    gn = torch.norm(viewspace_grad[:, :2], dim=-1, keepdim=True) * vis[:, None]
    gna = torch.norm(viewspace_grad_abs[:, :2], dim=-1, keepdim=True) * vis[:, None]
    cnt = vis.to(torch.float32)[:, None]
    rmax = radii.clone()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        packed = torch.cat([gn, gna, cnt], dim=1).contiguous()
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        gn, gna, cnt = packed[:, 0:1], packed[:, 1:2], packed[:, 2:3]
        dist.all_reduce(rmax, op=dist.ReduceOp.MAX, group=group)
    return gn, gna, cnt, rmax
