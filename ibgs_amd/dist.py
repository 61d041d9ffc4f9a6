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


def init_from_env(backend=None, force=False):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns (rank, world_size, local_rank).  A single process needs no initialisation; `force` initialises a group of one anyway (the exchange classes
    below then issue their collectives at world size 1 when built with force=True: the RCCL path on a one-GPU box, tests/test_gpu_rccl_world1.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
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
    if bucket is None or bucket.numels != [p.numel() for p in params] or bucket.flat.device != params[0].device:
        bucket = GradBucket(params)            # first call, or the trainer replaced its tensors (densification changes P)
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


def _exchanging(group, force):
    """True when collectives are to be issued: a process group of more than one rank, or any initialised group under `force`."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or bool(force)


def _resolve(x):
    """tensor | list of tensors | zero-argument callable returning either -> list of tensors (or [])."""
    if callable(x) and not isinstance(x, torch.Tensor):
        x = x()
    if x is None:
        return []
    return list(x) if isinstance(x, (list, tuple)) else [x]


class ViewParallelReducer:
    """Gradient exchange of one view-parallel step, sized for xGMI (point-to-point links: bytes per rank are
    what costs).  For SH degree 3 the SH coefficients are 48 of the 59 gradient floats per Gaussian, but the
    dL/dsh of ONE view is rank one -- basis(direction to that camera) x dL/dRGB (reference backward.cu:114-160)
    -- so ranks all-gather the 3-float dL/dRGB of their views (plus the camera centres) and every rank rebuilds
    the summed dL/dsh locally (`ibgs_sh_grad_from_views`); only the remaining parameters go through the flat
    all-reduce.  Per rank at P = 1M, 8 ranks: ~77 MB + 84 MB on the links instead of ~413 MB.

        red = ViewParallelReducer(params, sh=[f_dc, f_rest], means3D=xyz)
        with red.capture():            # one or more backward passes (views) per rank
            loss.backward()
        red.reduce()                   # every p.grad now holds the sum over all ranks' views

    The result equals the sequential accumulation of the single-view gradients (summation order differs).

    `params`, `sh`, `means3D` may be tensors / lists OR zero-argument callables that return them.  Pass callables
    (e.g. ``lambda: [g["params"][0] for g in optimizer.param_groups]``) when the trainer replaces its Parameters --
    the reference's densification does (scene/gaussian_model.py:377-463: cat_tensors_to_optimizer, _prune_optimizer):
    they are re-resolved on every reduce() and the flat bucket is rebuilt when sizes change.  A fixed list that has
    gone stale (no gradient on any tensor, or a Gaussian count that differs from the captured views') raises -- on
    EVERY rank, after the agreement below (a rank that raised on its own would leave its peers waiting in a collective).

    Order of one reduce() (collectives are issued with async_op, i.e. on the backend's own stream):
      1. agreement: a few integers (views captured, P, M, degree, sizes, "has a dense SH gradient", "local error") are MAX-reduced
         on the HOST over a gloo side group, so ranks can never enter different collective sequences (a rank that captured nothing
         would otherwise leave the others hanging in the all-gather).  It is started asynchronously before the local buffers are
         built and waited for just before the first data collective; the GPU is still running the backward kernels enqueued before,
         so it costs host time only (`last_agree_ms`).  `agree_every = N > 1` runs it only when this rank's own numbers changed since
         the last call, on the first call and every N-th call -- a lone rank whose numbers change in between then hangs with its
         peers until the side group's timeout instead of raising at once; the default (1) checks every step;
      2. all-gather of the dL/dRGB factors (ready first);  3. flat all-reduce of the dense gradients;
      4. SH expansion on the compute stream as soon as the gather has landed, WHILE the all-reduce is in flight;
      5. the reduced gradients are handed back as VIEWS of the flat bucket (`p.grad = view`, no copy back).
    The copy INTO the bucket is skipped as well for every parameter whose `.grad` already lives there: `direct=` names the leaves
    that are fed to the rasterizer as they are ({"means3D": xyz, "opacities": o, "scales": s, "rotations": q}); inside capture() the
    first backward then writes those four gradients straight into the bucket (rasterizer._grad_out_sink) and autograd adopts them.
    A trainer whose leaves pass through activations first (the reference's exp / sigmoid / normalize) can call `attach_grads()`
    instead of `zero_grad()`: every dense `p.grad` becomes a zeroed view of the bucket and autograd accumulates in place.
    SH gradients that did not come through the factored path (a backward outside capture(), colours converted in
    Python, another loss term on the SH leaves) are detected in step 1 and all-reduced densely on every rank."""

    def __init__(self, params, sh=None, means3D=None, group=None, expand=None, factored=True, agree_every=1, direct=None, force=False):
        """force: issue every collective (agreement, all-gather, all-reduce) also in a process group of ONE rank -- the sums are then the local gradients,
        bit for bit, and the whole exchange path has run on its backend (what a one-GPU box can show of RCCL)."""
        self._params, self._sh, self._means3D = params, sh, means3D
        self.force = bool(force)
        self.group, self._expand, self.factored = group, expand, factored
        self.agree_every = max(1, int(agree_every))
        self._direct = direct
        self.bucket = None
        self.items = None
        self.last_bytes = 0
        self.last_agree_ms = 0.0
        self.last_packed = 0          # parameters whose gradient had to be copied into the bucket by the last reduce()
        self._calls = 0
        self.n_agreements = 0
        self._last_sig = None
        self._agree = None
        if _exchanging(group, force):
            # host-side agreement channel: the default group when it is gloo already, else a gloo twin of it
            if dist.get_backend(group) == "gloo":
                self._agree = group if group is not None else dist.group.WORLD
            else:
                self._agree = dist.new_group(ranks=dist.get_process_group_ranks(group if group is not None else dist.group.WORLD), backend="gloo")

    # kept for callers that used the old attribute names
    @property
    def sh_parts(self):
        return _resolve(self._sh)

    @property
    def means3D(self):
        m = _resolve(self._means3D)
        return m[0] if m else None

    def _dense(self):
        params, sh_parts = _resolve(self._params), self.sh_parts
        return [p for p in params if not any(p is q for q in sh_parts)]

    def attach_grads(self):
        """Instead of zero_grad(): every dense parameter's `.grad` becomes a zeroed view of the flat bucket (ONE memset), so autograd
        accumulates this step's gradients in place and reduce() has nothing to copy.  Returns the number of bytes attached."""
        dense = self._dense()
        for q in self.sh_parts:          # the SH leaves get theirs from the factored exchange (a dense one would be all-reduced, see reduce())
            q.grad = None
        if not dense:
            return 0
        self._bucket_for(dense)
        self.bucket.flat.zero_()
        for p, v in zip(dense, self.bucket.unpack()):
            p.grad = v
        return self.bucket.flat.numel() * 4

    def capture(self):
        from . import rasterizer
        red = self

        def sink():
            direct = _resolve_dict(red._direct)
            if not direct:
                return None
            dense = red._dense()
            if not dense:
                return None
            red._bucket_for(dense)
            out = {}
            for p, v in zip(dense, red.bucket.unpack()):
                for name, t in direct.items():
                    if t is p and p.grad is None:          # autograd only adopts a returned tensor when the leaf has no gradient yet
                        out[name] = v
            return out or None

        class _Sinked:
            def __enter__(self):
                self._prev_sink = rasterizer._grad_out_sink
                rasterizer._grad_out_sink = sink()

            def __exit__(self, *exc):
                rasterizer._grad_out_sink = self._prev_sink

        if not self.factored:
            class _Null(_Sinked):
                def __enter__(self):
                    super().__enter__()
                    red.items = []
                    return red.items

                def __exit__(self, *exc):
                    super().__exit__(*exc)
                    return False
            return _Null()

        class _Ctx(rasterizer.capture_sh_factors):
            def __enter__(self):
                self._s = _Sinked(); self._s.__enter__()
                red.items = super().__enter__()
                return red.items

            def __exit__(self, *exc):
                self._s.__exit__(*exc)
                return super().__exit__(*exc)
        return _Ctx()

    def _bucket_for(self, tensors):
        numels = [t.numel() for t in tensors]
        if self.bucket is None or self.bucket.numels != numels or self.bucket.flat.device != tensors[0].device:
            self.bucket = GradBucket(tensors)
        return self.bucket

    def reduce(self, average=False):
        import time
        world = dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1
        multi = _exchanging(self.group, self.force)          # collectives are issued (world > 1, or forced at world 1)
        params, sh_parts, means3D = _resolve(self._params), self.sh_parts, self.means3D
        items, self.items = self.items or [], None
        dense = [p for p in params if not any(p is q for q in sh_parts)]
        # ---- local sanity (stale references after densification surface here, not as silently frozen parameters).  Nothing is
        # raised before the agreement: a rank that left now would never enter the collectives its peers are about to wait in.
        err = None
        if params and all(p.grad is None for p in params) and not items:
            err = ("ViewParallelReducer.reduce(): no tensor in `params` has a gradient -- were the Parameters replaced "
                   "(densification)? Pass callables for params / sh / means3D or rebuild the reducer.")
        n_local = len(items)
        P = int(items[0]["dcolor"].shape[0]) if items else (int(means3D.shape[0]) if means3D is not None else -1)
        M = int(items[0]["M"]) if items else -1
        degree = int(items[0]["degree"]) if items else -1
        for it in items:
            if err is None and (int(it["dcolor"].shape[0]) != P or int(it["M"]) != M or int(it["degree"]) != degree):
                err = "ViewParallelReducer: captured views disagree on (P, M, degree)"
        if err is None and items and (means3D is None or int(means3D.shape[0]) != P):
            err = ("ViewParallelReducer: means3D has %s rows but the captured views have P = %d (stale reference after "
                   "densification?)" % ("no" if means3D is None else int(means3D.shape[0]), P))
        if err is None and items and sum(int(q.shape[1]) for q in sh_parts) != M:
            err = ("ViewParallelReducer: the `sh` leaves hold %d coefficients, the captured views M = %d"
                   % (sum(int(q.shape[1]) for q in sh_parts), M))
        # an SH gradient that did not come through the factored path must be reduced densely
        sh_dense = any(q.grad is not None for q in sh_parts)
        # ---- 1. agreement across ranks, on the host, asynchronously: it travels while the buffers below are set up
        mine = [n_local, P, M, degree, sum(p.numel() for p in dense), sum(q.numel() for q in sh_parts)]
        sig = (tuple(mine), bool(sh_dense))
        self._calls += 1
        agree = multi and (self.agree_every == 1 or self._calls == 1 or (self._calls - 1) % self.agree_every == 0
                               or sig != self._last_sig or err is not None)
        self._last_sig = sig
        work_a, v, t_agree = None, None, 0.0
        if agree:
            v = torch.tensor(mine + [-x for x in mine] + [int(sh_dense), int(err is not None)], dtype=torch.int64)
            t0 = time.perf_counter()
            work_a = dist.all_reduce(v, op=dist.ReduceOp.MAX, group=self._agree, async_op=True)
            self.n_agreements += 1
            t_agree = time.perf_counter() - t0
        elif not multi and err is not None:
            raise RuntimeError(err)
        # ---- buffers: factors of the local views; the flat bucket (gradients that already live in it are not copied)
        buf = None
        if items and err is None:
            dev = items[0]["dcolor"].device
            # one buffer per exchange: each view's (P, 3) factor followed by its camera centre -> ONE all-gather
            buf = torch.empty(n_local, P + 1, 3, dtype=torch.float32, device=dev)
            for i, it in enumerate(items):
                buf[i, :P] = it["dcolor"]
                buf[i, P] = it["campos"].to(dev)
        if work_a is not None:
            t0 = time.perf_counter()
            work_a.wait()
            self.last_agree_ms = (t_agree + time.perf_counter() - t0) * 1e3
            k = len(mine)
            hi, lo = v[:k].tolist(), [-x for x in v[k:2 * k].tolist()]
            if bool(v[2 * k + 1].item()):
                raise RuntimeError(err if err is not None else "ViewParallelReducer: another rank failed its local checks (stale parameters "
                                   "after densification, or views that disagree on (P, M, degree)); no collective was started")
            if hi != lo:
                raise RuntimeError("ViewParallelReducer: ranks disagree on (views captured, P, M, degree, dense numel, sh numel): "
                                   "max %s min %s -- every rank must run the same number of backward passes inside capture()" % (hi, lo))
            sh_dense = bool(v[2 * k].item())
        elif err is not None:
            raise RuntimeError(err)
        # ---- 2. factors on the wire first
        work_g = None
        if buf is not None and multi:
            allb = torch.empty(world * n_local, P + 1, 3, dtype=buf.dtype, device=buf.device)
            work_g = dist.all_gather_into_tensor(allb, buf, group=self.group, async_op=True)
            buf = allb
        # ---- 3. dense gradients (plus the SH leaves when some rank holds a dense SH gradient)
        flat_list = dense + (sh_parts if sh_dense else [])
        work_d, bucket, views = None, None, None
        self.last_packed = 0
        if multi and flat_list:
            bucket = self._bucket_for(flat_list)
            views = bucket.unpack()
            for p, view in zip(flat_list, views):
                g = p.grad
                if g is None:
                    view.zero_(); self.last_packed += 1
                elif not (g.data_ptr() == view.data_ptr() and g.numel() == view.numel() and g.is_contiguous()):
                    view.copy_(g.reshape(view.shape)); self.last_packed += 1          # a gradient that lives elsewhere: one copy in
            work_d = dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.last_bytes = (0 if bucket is None else bucket.flat.numel() * 4) + (0 if (buf is None or not multi) else n_local * (P + 1) * 12)
        # ---- 4. SH expansion overlaps the all-reduce
        g_sh = None
        if items:
            if work_g is not None:
                work_g.wait()
            expand = self._expand
            if expand is None:
                from .shgrad import sh_grad_from_views as expand
            dcolor, campos = buf[:, :P], buf[:, P].contiguous()      # dcolor keeps the (P + 1) * 3 view stride
            g_sh = expand(means3D.detach(), campos, dcolor, degree, M)
        # ---- 5. hand the sums back: views of the bucket, no copy
        if work_d is not None:
            work_d.wait()
            if average:
                bucket.flat.div_(world)
            for p, view in zip(flat_list, views):
                p.grad = view
        if g_sh is not None:
            if average:
                g_sh = g_sh / world
            off = 0
            for part in sh_parts:                      # (P, M_i, 3) slices of the (P, M, 3) result
                m = part.shape[1]
                gp = g_sh[:, off:off + m, :].contiguous() if len(sh_parts) > 1 else g_sh.view_as(part)
                part.grad = gp if part.grad is None else part.grad + gp      # part.grad (if any) is already the all-rank sum
                off += m


def _resolve_dict(d):
    """{"name": tensor | callable} or a callable returning such a dict -> {"name": tensor}."""
    if d is None:
        return {}
    if callable(d):
        d = d()
    return {k: (v() if (callable(v) and not isinstance(v, torch.Tensor)) else v) for k, v in (d or {}).items()}


def allreduce_densification_stats(viewspace_grad, viewspace_grad_abs, radii, group=None, force=False):
    """Per-view statistics consumed by GaussianModel.add_densification_stats
    (scene/gaussian_model.py:600-604; train.py:400-410), reduced over the views of this step:
    returns (sum of ||grad[:, :2]||, sum of ||grad_abs[:, :2]||, visible count, max radii)."""
    vis = radii > 0
    gn = torch.norm(viewspace_grad[:, :2], dim=-1, keepdim=True) * vis[:, None]
    gna = torch.norm(viewspace_grad_abs[:, :2], dim=-1, keepdim=True) * vis[:, None]
    cnt = vis.to(torch.float32)[:, None]
    rmax = radii.clone()
    if _exchanging(group, force):
        packed = torch.cat([gn, gna, cnt], dim=1).contiguous()
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
        gn, gna, cnt = packed[:, 0:1], packed[:, 1:2], packed[:, 2:3]
        dist.all_reduce(rmax, op=dist.ReduceOp.MAX, group=group)
    return gn, gna, cnt, rmax


# ---- densification on replicated parameters ---------------------------------------------------------------------------
# Every rank runs densify_and_prune itself on its replica.  The decisions (which points to clone / split / prune) are functions of the
# reduced statistics above, so they agree; what does NOT agree by itself are the random samples of densify_and_split
# (`torch.normal(mean=means, std=stds)`, scene/gaussian_model.py:498-502, 562-566) -- every process has its own generator state.
def seed_for_densification(iteration, base_seed=0, devices=None):
    """Give every rank the SAME generator state before a densification step: seeds torch's CPU generator and the generators of
    `devices` (default: the current HIP device, if any) with a value that depends only on (base_seed, iteration).  Call it on every rank
    right before `gaussians.densify_and_prune(...)`; the replicas then draw identical samples and stay bit-identical.
    This reseeds the GLOBAL generators for good: everything drawn afterwards (the trainer's random background, augmentation noise) is
    then identical on all ranks and restarts from a predictable value -- prefer `synchronized_densification_rng`, which puts the
    generators back."""
    seed = (int(base_seed) * 1000003 + int(iteration)) & 0x7FFFFFFF
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        for d in (devices if devices is not None else [torch.cuda.current_device()]):
            torch.cuda.manual_seed(seed) if d == torch.cuda.current_device() else torch.cuda.default_generators[d].manual_seed(seed)
    return seed


class synchronized_densification_rng:
    """`with synchronized_densification_rng(iteration, base_seed): gaussians.densify_and_prune(...)` -- inside the block every rank
    draws from generators seeded with the same (base_seed, iteration) value, so densify_and_split's `torch.normal` samples agree; on
    exit the CPU and device generators continue exactly where they were (torch.random.fork_rng), so the ranks' other random draws stay
    independent of each other and of the densification schedule."""

    def __init__(self, iteration, base_seed=0, devices=None):
        self.iteration, self.base_seed = iteration, base_seed
        if devices is None:
            devices = [torch.cuda.current_device()] if torch.cuda.is_available() else []
        self.devices = list(devices)

    def __enter__(self):
        self._fork = torch.random.fork_rng(devices=self.devices)
        self._fork.__enter__()
        return seed_for_densification(self.iteration, self.base_seed, self.devices if self.devices else None)

    def __exit__(self, *exc):
        return self._fork.__exit__(*exc)


def assert_replicas_identical(tensors, group=None, what="parameters", force=False):
    """Cheap guard for the replicated-parameter invariant: one all-reduce (MIN and MAX) of a float64 checksum and the element count per
    tensor.  Raises on EVERY rank when any replica differs (e.g. a densification that ran with unsynchronised generators)."""
    if not _exchanging(group, force):
        return
    vals = []
    for t in _resolve(tensors):
        vals += [t.detach().double().sum().reshape(1), torch.tensor([float(t.numel())], dtype=torch.float64, device=t.device)]
    cs = torch.cat([v.to(vals[0].device) for v in vals])
    lo, hi = cs.clone(), cs.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if not torch.equal(lo, hi):
        bad = [i // 2 for i in range(0, cs.numel(), 2) if not (lo[i] == hi[i] and lo[i + 1] == hi[i + 1])]
        raise RuntimeError("view-parallel replicas diverged: %s %s differ between ranks (unsynchronised densification?)" % (what, bad))


# ---- the step without the replicated optimiser (round 4) ---------------------------------------------------------------
# With the rasterizer at ~1.7 ms per view the serial terms of an 8-GPU step are the exchange and the optimiser: every rank running Adam over
# all P x 59 floats (0.32 ms at 1 M Gaussians, ibgs_adam_step) repeats work eight times.  ShardedOptimizerStep splits the Gaussians into
# N contiguous row ranges, one per rank:
#   1. dense gradients:  REDUCE-SCATTER per parameter -- a rank receives the all-rank sum of ITS rows only (half the bytes of an all-reduce);
#   2. SH gradients:     the dL/dRGB factors of the views travel by ALL-TO-ALL -- a rank receives every view's factors for its rows only
#                        (P x 3 floats in total instead of N x P x 3 with the all-gather of ViewParallelReducer) and rebuilds dL/dsh for its rows;
#   3. Adam on the rank's rows of every parameter (1 / N of the work; the moments of the other rows are not touched);
#   4. ALL-GATHER of the updated rows: every rank holds the full, identical parameters again.
# The optimiser's state tensors stay full-sized (the reference's densification surgery on them keeps working: scene/gaussian_model.py:
# 377-463) but only a rank's own rows are current: call gather_state() before densify_and_prune, before a checkpoint, and whenever the
# row ranges change.  Same sums as ViewParallelReducer + a replicated step.  Every row's sum is formed on ONE rank (its owner) and travels to the
# others as bits, so the replicas are identical by construction at any world size; with `ordered` (default) that sum is also formed in rank order,
# i.e. it equals a single process accumulating the N views one after the other, bit for bit.
def _rows(P, world, rank):
    chunk = (P + world - 1) // world
    return chunk, min(P, rank * chunk), min(P, (rank + 1) * chunk)


def _fused_adam_rows(entries):
    """Default `adam`: ibgs_adam_step (csrc/adam.hip) on row slices.  entries: dicts with param / grad / exp_avg / exp_avg_sq (contiguous
    slices), lr, betas, eps, step (the 1-based step count)."""
    import ctypes, math
    from . import _lib
    lib = _lib.load()
    ds = []
    for e in entries:
        if e["param"].numel() == 0:
            continue
        if not e["param"].is_cuda:
            raise RuntimeError("ShardedOptimizerStep: the default Adam runs on the MI355X only (ibgs_adam_step); there is no CPU path")
        d = _lib.AdamTensor()
        d.param, d.grad, d.exp_avg, d.exp_avg_sq = e["param"].data_ptr(), e["grad"].data_ptr(), e["exp_avg"].data_ptr(), e["exp_avg_sq"].data_ptr()
        d.numel = e["param"].numel()
        b1, b2 = e["betas"]
        d.lr, d.beta1, d.beta2, d.eps = float(e["lr"]), float(b1), float(b2), float(e["eps"])
        d.bias_correction1 = 1.0 - math.pow(b1, e["step"]); d.bias_correction2 = 1.0 - math.pow(b2, e["step"])
        ds.append(d)
    if not ds:
        return
    dev = entries[0]["param"].device
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        for i in range(0, len(ds), 16):
            part = ds[i:i + 16]
            arr = (_lib.AdamTensor * len(part))(*part)
            rc = lib.ibgs_adam_step(stream, len(part), ctypes.cast(arr, ctypes.c_void_p))
            if rc < 0:
                raise RuntimeError("ibgs_adam_step failed (%d): %s" % (rc, _lib.last_error()))


class ShardedOptimizerStep:
    """Exchange + optimiser step of one view-parallel iteration with the Adam work split over the ranks (see above).

        opt = FusedAdam(param_groups, lr=0.0, eps=1e-15)            # or the reference's torch.optim.Adam: only its state layout is used
        sh = ShardedOptimizerStep(opt, sh=[f_dc, f_rest], means3D=xyz)
        with sh.capture():                                          # one or more backward passes (views) per rank
            loss.backward()
        sh.step()                                                   # instead of reducer.reduce(); optimizer.step()
        ...
        sh.gather_state(); gaussians.densify_and_prune(...)         # the moments of all rows, before anything reads or reshapes them

    Every parameter of `optimizer` must be a per-Gaussian tensor (first dimension P); `sh` names the SH leaves whose gradient arrives
    factored (rasterizer.capture_sh_factors), `means3D` the positions.  `expand` / `adam` replace the HIP kernels in the CPU tests."""

    def __init__(self, optimizer, sh=None, means3D=None, group=None, expand=None, adam=None, ordered=True, force=False):
        """ordered (default): the dense gradients travel by ALL-TO-ALL -- every rank sends each owner its block of rows, point to point -- and the
        owner adds the N blocks IN RANK ORDER: ((g0 + g1) + g2) + ..., the order of a single process accumulating the views one after the other,
        whatever the backend and however many ranks (round 5; bit-identical to that sequential sum at any world size, tests/test_dist_gloo.py at
        world 4).  Same bytes on the wire as a reduce-scatter, and on the xGMI full mesh the pattern SURVEY 8(e) asks for (7 concurrent
        point-to-point transfers per rank instead of a ring).  ordered=False: the backend's reduce_scatter_tensor and its summation order."""
        self.opt, self._sh, self._means3D, self.group = optimizer, sh, means3D, group
        self.ordered = bool(ordered)
        self.force = bool(force)          # issue the collectives in a group of one rank as well (ViewParallelReducer: force)
        self._expand, self._adam = expand, (adam or _fused_adam_rows)
        self.items = None
        self.last_bytes = 0
        self.state_is_gathered = True          # nothing is sharded before the first step
        self._agree = None
        for g in optimizer.param_groups:          # the update below is plain Adam from lr / betas / eps (the reference's only use, gaussian_model.py:241)
            if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
                raise NotImplementedError("ShardedOptimizerStep covers Adam without weight_decay / amsgrad / maximize (group %r)" % g.get("name"))
        # after step() only this rank's rows of exp_avg / exp_avg_sq are current: a state_dict() taken then would save stale moments for the others
        # (gather_state() is a collective, and checkpoints are often written by rank 0 alone: raise rather than start an all-gather the others never join)
        if hasattr(optimizer, "register_state_dict_pre_hook"):
            def _guard(opt, me=self):
                if not me.state_is_gathered:
                    raise RuntimeError("ShardedOptimizerStep: the Adam moments are sharded over the ranks; call gather_state() on EVERY rank before optimizer.state_dict()")
            optimizer.register_state_dict_pre_hook(_guard)
        if _exchanging(group, force):
            if dist.get_backend(group) == "gloo":
                self._agree = group if group is not None else dist.group.WORLD
            else:
                self._agree = dist.new_group(ranks=dist.get_process_group_ranks(group if group is not None else dist.group.WORLD), backend="gloo")

    def _world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.group), dist.get_rank(self.group)
        return 1, 0

    def _params(self):
        return [(g, p) for g in self.opt.param_groups for p in g["params"]]

    def capture(self):
        from . import rasterizer
        me = self

        class _Ctx(rasterizer.capture_sh_factors):
            def __enter__(self):
                me.items = super().__enter__()
                return me.items
        return _Ctx()

    def _state(self, p):
        st = self.opt.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, average=False):
        """average: divide the summed gradients by the number of ranks (ViewParallelReducer.reduce(average=True))"""
        world, rank = self._world()
        multi = _exchanging(self.group, self.force)
        items, self.items = self.items or [], None
        sh_parts = _resolve(self._sh)
        m3 = _resolve(self._means3D)
        means3D = m3[0] if m3 else None
        pairs = self._params()
        if not pairs:
            return
        P = int(pairs[0][1].shape[0])
        err = None
        for _, p in pairs:
            if int(p.shape[0]) != P or not p.is_contiguous():
                err = "ShardedOptimizerStep: every parameter must be a contiguous per-Gaussian tensor with the same first dimension"
        M = int(items[0]["M"]) if items else -1
        degree = int(items[0]["degree"]) if items else -1
        if err is None and items and (means3D is None or int(means3D.shape[0]) != P or any(int(it["dcolor"].shape[0]) != P for it in items)):
            err = "ShardedOptimizerStep: the captured views / means3D do not have the optimiser's P = %d rows (stale references after densification?)" % P
        if err is None and items and sum(int(q.shape[1]) for q in sh_parts) != M:
            err = "ShardedOptimizerStep: the `sh` leaves hold %d coefficients, the captured views M = %d" % (sum(int(q.shape[1]) for q in sh_parts), M)
        has_grad = [int(p.grad is not None) for _, p in pairs]
        if multi:          # ranks must agree on the collective sequence before the first one starts (as ViewParallelReducer does)
            mine = [len(items), P, M, degree] + has_grad
            v = torch.tensor(mine + [-x for x in mine] + [int(err is not None)], dtype=torch.int64)
            dist.all_reduce(v, op=dist.ReduceOp.MAX, group=self._agree)
            k = len(mine)
            if bool(v[2 * k].item()):
                raise RuntimeError(err if err is not None else "ShardedOptimizerStep: another rank failed its local checks; no collective was started")
            if v[:k].tolist() != [-x for x in v[k:2 * k].tolist()]:
                raise RuntimeError("ShardedOptimizerStep: ranks disagree on (views captured, P, M, degree, which parameters have gradients)")
        elif err is not None:
            raise RuntimeError(err)
        chunk, lo, hi = _rows(P, world, rank)
        n = hi - lo
        dev = pairs[0][1].device
        self.last_bytes = 0

        def scatter_sum(g):          # (P, ...) local gradient -> (n, ...) all-rank sum of this rank's rows
            flat = g.reshape(P, -1)
            if not multi:
                return flat
            k = flat.shape[1]
            if P == chunk * world and flat.is_contiguous():
                src = flat
            else:
                src = torch.zeros(chunk * world, k, dtype=flat.dtype, device=flat.device); src[:P] = flat
            if self.ordered:
                recv = torch.empty(world, chunk, k, dtype=flat.dtype, device=flat.device)
                dist.all_to_all_single(recv.view(-1), src.view(-1), group=self.group)          # recv[q] = rank q's gradient of MY rows
                out = recv[0]
                for q in range(1, world):          # rank order: the sum a single process would form view by view
                    out += recv[q]
            else:
                out = torch.empty(chunk, k, dtype=flat.dtype, device=flat.device)
                dist.reduce_scatter_tensor(out, src, op=dist.ReduceOp.SUM, group=self.group)
            self.last_bytes += out.numel() * 4 * (world - 1)
            return out[:n]

        # ---- 2. the SH leaves: factors by all-to-all, expansion for the own rows
        g_sh = None
        if items:
            n_local = len(items)
            if multi:
                send = torch.zeros(world, n_local, chunk, 3, dtype=torch.float32, device=dev)
                for i, it in enumerate(items):
                    pad = torch.zeros(chunk * world, 3, dtype=torch.float32, device=dev); pad[:P] = it["dcolor"]
                    send[:, i] = pad.view(world, chunk, 3)
                recv = torch.empty_like(send)
                dist.all_to_all_single(recv.view(-1), send.view(-1), group=self.group)          # recv[q, i] = view i of rank q, my rows
                cam = torch.stack([it["campos"].to(dev).reshape(3) for it in items]).contiguous()
                cams = torch.empty(world * n_local, 3, dtype=torch.float32, device=dev)
                dist.all_gather_into_tensor(cams, cam, group=self.group)
                dcolor = recv.view(world * n_local, chunk, 3)[:, :n].contiguous()
                self.last_bytes += (recv.numel() // world) * (world - 1) * 4
            else:
                dcolor = torch.stack([it["dcolor"] for it in items]).contiguous()
                cams = torch.stack([it["campos"].to(dev).reshape(3) for it in items]).contiguous()
            expand = self._expand
            if expand is None:
                from .shgrad import sh_grad_from_views as expand
            with torch.enable_grad():          # (a replacement `expand` may differentiate through an SH evaluation: tests)
                g_sh = expand(means3D.detach()[lo:hi].contiguous(), cams, dcolor, degree, M) if n > 0 else torch.zeros(0, M, 3, device=dev)
            g_sh = g_sh.detach()
        # ---- 1. + 3. per parameter: the sum of the own rows, then Adam on them
        entries, off = [], 0
        sh_off = {}
        for q in sh_parts:
            sh_off[id(q)] = off; off += int(q.shape[1])
        for (group, p), hg in zip(pairs, has_grad):
            is_sh = id(p) in sh_off
            g_rows = scatter_sum(p.grad) if hg else None
            if is_sh and g_sh is not None:
                o = sh_off[id(p)]
                part = g_sh[:, o:o + int(p.shape[1])].reshape(n, -1)
                g_rows = part if g_rows is None else g_rows + part          # a dense SH gradient (another loss term) on top of the factored one
            if g_rows is None:
                continue
            if average and world > 1:
                g_rows = g_rows / world
            st = self._state(p)
            st["step"] += 1
            pr = p.data.reshape(P, -1)
            entries.append({"param": pr[lo:hi], "grad": g_rows.contiguous(), "exp_avg": st["exp_avg"].reshape(P, -1)[lo:hi],
                            "exp_avg_sq": st["exp_avg_sq"].reshape(P, -1)[lo:hi], "lr": group["lr"], "betas": group["betas"], "eps": group["eps"],
                            "step": float(st["step"]), "full": p})
        self._adam(entries)
        # ---- 4. every rank gets every row back
        if multi:
            for e in entries:
                self._gather_rows(e["full"].data, P, chunk, lo, hi, world)
            self.state_is_gathered = False
        for _, p in pairs:
            p.grad = None

    def _gather_rows(self, full, P, chunk, lo, hi, world):
        flat = full.reshape(P, -1)
        k = flat.shape[1]
        if P == chunk * world:
            dist.all_gather_into_tensor(flat, flat[lo:hi].clone(), group=self.group)
        else:
            mine = torch.zeros(chunk, k, dtype=flat.dtype, device=flat.device); mine[:hi - lo] = flat[lo:hi]
            allr = torch.empty(chunk * world, k, dtype=flat.dtype, device=flat.device)
            dist.all_gather_into_tensor(allr, mine, group=self.group)
            flat.copy_(allr[:P])
        self.last_bytes += chunk * k * 4 * (world - 1)

    @torch.no_grad()
    def gather_state(self):
        """All-gather the Adam moments so that every rank holds them for every row (before densification surgery, checkpoints, or a change
        of the row ranges).  No-op when nothing has been stepped since the last call."""
        world, rank = self._world()
        if not _exchanging(self.group, self.force) or self.state_is_gathered:
            self.state_is_gathered = True
            return
        for _, p in self._params():
            st = self.opt.state.get(p)
            if not st:
                continue
            P = int(p.shape[0])
            chunk, lo, hi = _rows(P, world, rank)
            for key in ("exp_avg", "exp_avg_sq"):
                self._gather_rows(st[key], P, chunk, lo, hi, world)
        self.state_is_gathered = True
