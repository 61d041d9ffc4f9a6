"""View-parallel gradient reduction on CPU: 2 processes over gloo (the GPU path uses RCCL through the
same code).  The all-reduced gradients must equal the sequential accumulation of the per-view gradients."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ibgs_amd import dist as vdist


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = vdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    params = [torch.randn(50, 3, generator=g).requires_grad_(True), torch.randn(50, 16, 3, generator=g).requires_grad_(True),
              torch.randn(50, 1, generator=g).requires_grad_(True)]
    view = vdist.views_for_rank(step=3, rank=rank, world_size=world, n_views=8)
    # a stand-in "render loss" that depends on the view so that ranks produce different gradients
    loss = sum(((p * (view + 1.0 + i)) ** 2).sum() for i, p in enumerate(params))
    loss.backward()
    local = [p.grad.clone() for p in params]
    bucket = vdist.allreduce_gradients(params)
    bucket = vdist.allreduce_gradients(params, bucket) if False else bucket
    radii = torch.tensor([0, 3, 5, 0, 2][rank:] + [1] * (45 + rank), dtype=torch.int32)[:50]
    vg = torch.randn(50, 3, generator=torch.Generator().manual_seed(10 + rank))
    gn, gna, cnt, rmax = vdist.allreduce_densification_stats(vg, vg.abs(), radii)
    torch.save({"view": view, "local": local, "reduced": [p.grad.clone() for p in params], "gn": gn, "cnt": cnt, "rmax": rmax,
                "vg": vg, "radii": radii}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_allreduce_equals_sequential_accumulation(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, "r%d.pt" % r)) for r in range(world)]
    assert [r["view"] for r in res] == [6, 7]
    for i in range(3):
        want = res[0]["local"][i] + res[1]["local"][i]
        for r in res:
            np.testing.assert_allclose(r["reduced"][i].numpy(), want.numpy(), rtol=1e-6)
    gn = sum(torch.norm(r["vg"][:, :2], dim=-1, keepdim=True) * (r["radii"] > 0)[:, None] for r in res)
    np.testing.assert_allclose(res[0]["gn"].numpy(), gn.numpy(), rtol=1e-6)
    np.testing.assert_allclose(res[1]["cnt"].numpy(), sum((r["radii"] > 0).float()[:, None] for r in res).numpy())
    assert torch.equal(res[0]["rmax"], torch.maximum(res[0]["radii"], res[1]["radii"]))


def test_single_process_is_a_noop():
    p = [torch.ones(4, requires_grad=True)]
    p[0].grad = torch.full((4,), 2.0)
    assert vdist.allreduce_gradients(p) is None
    assert torch.all(p[0].grad == 2.0)
    b = vdist.GradBucket(p)
    flat = b.pack([p[0].grad])
    assert flat.numel() == 4 and torch.all(b.unpack()[0] == 2.0)


def _expand_ref(means3D, camposes, dcolor, degree, M):
    """Independent restatement of ibgs_sh_grad_from_views through autograd of the renderer's eval_sh."""
    from ibgs_amd.renderer import eval_sh
    P = means3D.shape[0]
    out = torch.zeros(P, M, 3)
    for v in range(camposes.shape[0]):
        d = means3D - camposes[v]
        d = d / d.norm(dim=1, keepdim=True)
        sh0 = torch.zeros(P, 3, M, requires_grad=True)
        col = eval_sh(degree, sh0, d)
        (g,) = torch.autograd.grad(col, sh0, grad_outputs=dcolor[v])
        out += g.transpose(1, 2)
    return out


def _worker_factored(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    vdist.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(0)
    P, M, deg = 40, 16, 2
    xyz = torch.randn(P, 3, generator=g).requires_grad_(True)
    # SH split into two leaves like the reference's _features_dc / _features_rest (get_features = cat along dim 1)
    f_dc = torch.randn(P, 1, 3, generator=g).requires_grad_(True)
    f_rest = torch.randn(P, M - 1, 3, generator=g).requires_grad_(True)
    opa = torch.rand(P, 1, generator=g).requires_grad_(True)
    red = vdist.ViewParallelReducer([xyz, f_dc, f_rest, opa], sh=[f_dc, f_rest], means3D=xyz, expand=_expand_ref)
    gr = torch.Generator().manual_seed(100 + rank)
    items = []
    with red.capture() as sink:                    # two local views per rank; the HIP backward would fill `sink`
        for v in range(2):
            it = {"dcolor": torch.randn(P, 3, generator=gr), "campos": torch.randn(3, generator=gr) * 4.0, "degree": deg, "M": M}
            sink.append(it); items.append(it)
    xyz.grad = torch.randn(P, 3, generator=gr); opa.grad = torch.randn(P, 1, generator=gr)
    local = {"xyz": xyz.grad.clone(), "opa": opa.grad.clone()}
    red.reduce()
    torch.save({"items": items, "local": local, "xyz": xyz.grad.clone(), "opa": opa.grad.clone(), "shs": torch.cat([f_dc.grad, f_rest.grad], dim=1),
                "means": xyz.detach().clone()}, os.path.join(out_dir, "f%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_factored_sh_exchange_over_gloo(tmp_path):
    world = 2
    mp.spawn(_worker_factored, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, "f%d.pt" % r)) for r in range(world)]
    campos = torch.stack([it["campos"] for r in res for it in r["items"]])          # rank-major view order
    dcolor = torch.stack([it["dcolor"] for r in res for it in r["items"]])
    want = _expand_ref(res[0]["means"], campos, dcolor, 2, 16)
    for r in res:
        np.testing.assert_allclose(r["shs"].numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
        assert not r["shs"][:, 9:].any()                                             # above the active degree
        np.testing.assert_allclose(r["xyz"].numpy(), (res[0]["local"]["xyz"] + res[1]["local"]["xyz"]).numpy(), rtol=1e-6)
        np.testing.assert_allclose(r["opa"].numpy(), (res[0]["local"]["opa"] + res[1]["local"]["opa"]).numpy(), rtol=1e-6)
    assert torch.equal(res[0]["shs"], res[1]["shs"])                                 # every rank holds the same bits


def test_capture_restores_previous_sink():
    from ibgs_amd import rasterizer
    assert rasterizer._sh_factor_sink is None
    with rasterizer.capture_sh_factors() as outer:
        with rasterizer.capture_sh_factors() as inner:
            assert rasterizer._sh_factor_sink is inner
        assert rasterizer._sh_factor_sink is outer
    assert rasterizer._sh_factor_sink is None


# ---- robustness of the reducer: ranks can never take different collective sequences, parameters that the
# trainer replaced (densification) are picked up, dense SH gradients are never left rank-local ---------------------------------
def _worker_robust(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    vdist.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(0)
    P, M, deg = 30, 16, 3
    state = {"xyz": torch.randn(P, 3, generator=g).requires_grad_(True), "sh": torch.randn(P, M, 3, generator=g).requires_grad_(True),
             "opa": torch.rand(P, 1, generator=g).requires_grad_(True)}
    red = vdist.ViewParallelReducer(lambda: [state["xyz"], state["sh"], state["opa"]], sh=lambda: state["sh"], means3D=lambda: state["xyz"],
                                    expand=_expand_ref)
    gr = torch.Generator().manual_seed(100 + rank)
    out = {}

    # (a) nothing captured (backward outside capture / colours converted in Python): the SH gradient is dense and must be summed
    for k in state:
        state[k].grad = torch.randn(state[k].shape, generator=gr)
    out["a_local"] = {k: v.grad.clone() for k, v in state.items()}
    red.reduce()
    out["a"] = {k: v.grad.clone() for k, v in state.items()}

    # (b) factored views PLUS another loss term that left a dense gradient on the SH leaf
    for k in state:
        state[k].grad = torch.randn(state[k].shape, generator=gr)
    out["b_local"] = {k: v.grad.clone() for k, v in state.items()}
    with red.capture() as sink:
        it = {"dcolor": torch.randn(P, 3, generator=gr), "campos": torch.randn(3, generator=gr) * 4.0, "degree": deg, "M": M}
        sink.append(it)
    out["b_item"] = it
    red.reduce()
    out["b"] = {k: v.grad.clone() for k, v in state.items()}
    out["b_means"] = state["xyz"].detach().clone()

    # (c) the trainer replaces its Parameters with longer ones (densification): callables pick them up, the bucket is rebuilt
    P2 = 41
    state = {"xyz": torch.randn(P2, 3, generator=g).requires_grad_(True), "sh": torch.randn(P2, M, 3, generator=g).requires_grad_(True),
             "opa": torch.rand(P2, 1, generator=g).requires_grad_(True)}
    red._params = lambda: [state["xyz"], state["sh"], state["opa"]]; red._sh = lambda: state["sh"]; red._means3D = lambda: state["xyz"]
    state["xyz"].grad = torch.randn(P2, 3, generator=gr); state["opa"].grad = torch.randn(P2, 1, generator=gr)
    out["c_local"] = {k: state[k].grad.clone() for k in ("xyz", "opa")}
    with red.capture() as sink:
        sink.append({"dcolor": torch.randn(P2, 3, generator=gr), "campos": torch.randn(3, generator=gr), "degree": deg, "M": M})
    red.reduce()
    out["c"] = {k: state[k].grad.clone() for k in ("xyz", "opa")}
    out["c_sh_shape"] = tuple(state["sh"].grad.shape)

    # (d) ranks capture different numbers of views: every rank raises instead of hanging in the all-gather
    with red.capture() as sink:
        for _ in range(1 + rank):
            sink.append({"dcolor": torch.randn(P2, 3, generator=gr), "campos": torch.randn(3, generator=gr), "degree": deg, "M": M})
    try:
        red.reduce()
        out["d"] = "no error"
    except RuntimeError as ex:
        out["d"] = str(ex)

    # (e) a fixed list gone stale (no gradients anywhere) raises locally
    stale = vdist.ViewParallelReducer([torch.zeros(3, requires_grad=True)])
    try:
        stale.reduce()
        out["e"] = "no error"
    except RuntimeError as ex:
        out["e"] = str(ex)
    torch.save(out, os.path.join(out_dir, "g%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_reducer_never_diverges_or_hangs(tmp_path):
    world = 2
    mp.spawn(_worker_robust, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, "g%d.pt" % r)) for r in range(world)]
    for r in res:
        for k in ("xyz", "sh", "opa"):       # (a) everything, the SH leaf included, is the two-rank sum
            np.testing.assert_allclose(r["a"][k].numpy(), (res[0]["a_local"][k] + res[1]["a_local"][k]).numpy(), rtol=1e-6)
        # (b) dense part summed over ranks + the expansion of both ranks' factors
        campos = torch.stack([x["b_item"]["campos"] for x in res]); dcolor = torch.stack([x["b_item"]["dcolor"] for x in res])
        want = res[0]["b_local"]["sh"] + res[1]["b_local"]["sh"] + _expand_ref(res[0]["b_means"], campos, dcolor, 3, 16)
        np.testing.assert_allclose(r["b"]["sh"].numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r["b"]["xyz"].numpy(), (res[0]["b_local"]["xyz"] + res[1]["b_local"]["xyz"]).numpy(), rtol=1e-6)
        for k in ("xyz", "opa"):             # (c)
            np.testing.assert_allclose(r["c"][k].numpy(), (res[0]["c_local"][k] + res[1]["c_local"][k]).numpy(), rtol=1e-6)
        assert r["c_sh_shape"] == (41, 16, 3)
        assert "ranks disagree" in r["d"], r["d"]
        assert "no tensor in `params` has a gradient" in r["e"], r["e"]
    assert torch.equal(res[0]["b"]["sh"], res[1]["b"]["sh"])


# ---- densification on replicas: identical random samples on every rank, and a guard that notices when they are not ----------------
def _densify_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    vdist.init_from_env(backend="gloo")
    torch.manual_seed(1000 + rank)                      # ranks arrive with different generator states, as in a real run
    _ = torch.rand(3 + rank)
    base = torch.arange(12, dtype=torch.float32).reshape(4, 3)
    vdist.seed_for_densification(iteration=700, base_seed=5)
    split = torch.normal(mean=torch.zeros(4, 3), std=torch.ones(4, 3))          # densify_and_split's sampling (gaussian_model.py:498-502)
    params = [base + split]
    vdist.assert_replicas_identical(params)              # passes: same samples everywhere
    other = torch.normal(mean=torch.zeros(4, 3), std=torch.ones(4, 3), generator=torch.Generator().manual_seed(rank))
    err = None
    try:
        vdist.assert_replicas_identical([base + other], what="split samples")
    except RuntimeError as ex:
        err = str(ex)
    torch.save({"split": split, "err": err}, os.path.join(out_dir, "d%d.pt" % rank))
    dist.barrier(); dist.destroy_process_group()


def test_densification_samples_agree_across_ranks(tmp_path):
    port = _free_port()
    mp.spawn(_densify_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "d0.pt"), torch.load(tmp_path / "d1.pt")
    assert torch.equal(r0["split"], r1["split"])
    assert r0["err"] and r1["err"] and "diverged" in r0["err"] and "diverged" in r1["err"]      # the guard fires on both ranks


# ---- round 3: a local error on ONE rank reaches every rank through the agreement; sparse agreement; gradients that live in the bucket ----
def _worker_round3(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    vdist.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(0)
    P, M, deg = 24, 16, 3
    state = {"xyz": torch.randn(P, 3, generator=g).requires_grad_(True), "sh": torch.randn(P, M, 3, generator=g).requires_grad_(True),
             "opa": torch.rand(P, 1, generator=g).requires_grad_(True)}
    gr = torch.Generator().manual_seed(100 + rank)
    out = {}

    # (a) rank 1 alone holds a stale means3D (P differs from its captured views): BOTH ranks must raise, nobody may hang
    stale = torch.randn(P + 5, 3)
    red = vdist.ViewParallelReducer([state["xyz"], state["sh"], state["opa"]], sh=state["sh"], means3D=(stale if rank == 1 else state["xyz"]), expand=_expand_ref)
    for k in ("xyz", "opa"):
        state[k].grad = torch.randn(state[k].shape, generator=gr)
    with red.capture() as sink:
        sink.append({"dcolor": torch.randn(P, 3, generator=gr), "campos": torch.randn(3, generator=gr), "degree": deg, "M": M})
    try:
        red.reduce()
        out["a"] = "no error"
    except RuntimeError as ex:
        out["a"] = str(ex)

    # (b) agree_every = 4: the agreement runs on call 1, when the own numbers change, and every 4th call; sums stay right
    red = vdist.ViewParallelReducer(lambda: [state["xyz"], state["sh"], state["opa"]], sh=lambda: state["sh"], means3D=lambda: state["xyz"],
                                    expand=_expand_ref, agree_every=4)
    out["b"] = []
    for step in range(6):
        if step == 2:          # "densification": every rank replaces its tensors, sizes change everywhere
            state = {"xyz": torch.randn(P + 3, 3, generator=g).requires_grad_(True), "sh": torch.randn(P + 3, M, 3, generator=g).requires_grad_(True),
                     "opa": torch.rand(P + 3, 1, generator=g).requires_grad_(True)}
        n = state["xyz"].shape[0]
        red.attach_grads()                                        # gradients accumulate inside the flat bucket
        lx, lo = torch.randn(n, 3, generator=gr), torch.randn(n, 1, generator=gr)
        state["xyz"].grad += lx; state["opa"].grad += lo          # what autograd's in-place accumulation does
        with red.capture() as sink:
            sink.append({"dcolor": torch.randn(n, 3, generator=gr), "campos": torch.randn(3, generator=gr), "degree": deg, "M": M})
        red.reduce()
        out["b"].append({"lx": lx, "xyz": state["xyz"].grad.clone(), "agreements": red.n_agreements, "packed": red.last_packed,
                         "alias": state["xyz"].grad.data_ptr() == red.bucket.unpack()[0].data_ptr()})

    # (c) the scoped generator for densification: same samples on every rank, generators back where they were afterwards
    torch.manual_seed(1000 + rank)
    _ = torch.rand(2 + rank)
    before = torch.get_rng_state().clone()
    with vdist.synchronized_densification_rng(iteration=700, base_seed=5):
        out["c_split"] = torch.normal(mean=torch.zeros(4, 3), std=torch.ones(4, 3))
    out["c_restored"] = torch.equal(before, torch.get_rng_state())
    out["c_next"] = torch.rand(3)
    torch.save(out, os.path.join(out_dir, "h%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_local_errors_sparse_agreement_and_bucket_resident_gradients(tmp_path):
    world = 2
    mp.spawn(_worker_round3, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, "h%d.pt" % r)) for r in range(world)]
    assert "stale reference" in res[1]["a"], res[1]["a"]                       # the rank that saw it reports it ...
    assert "another rank failed its local checks" in res[0]["a"], res[0]["a"]  # ... and its peer raises too instead of waiting in a collective
    for step in range(6):
        want = res[0]["b"][step]["lx"] + res[1]["b"][step]["lx"]
        for r in res:
            np.testing.assert_allclose(r["b"][step]["xyz"].numpy(), want.numpy(), rtol=1e-6)
            assert r["b"][step]["packed"] == 0 and r["b"][step]["alias"]       # nothing copied in, the sum is handed back as a view
    assert [x["agreements"] for x in res[0]["b"]] == [1, 1, 2, 2, 3, 3]        # call 1, the size change at call 3, the 4th-call schedule at call 5
    assert torch.equal(res[0]["c_split"], res[1]["c_split"])
    assert res[0]["c_restored"] and res[1]["c_restored"]
    assert not torch.equal(res[0]["c_next"], res[1]["c_next"])                 # the ranks' own streams stay independent


# ---- round 4: the step without the replicated optimiser: reduce-scatter -> Adam on the rank's rows -> all-gather -------------------------
def _adam_ref(entries):
    """torch.optim.Adam's single-tensor update (no weight decay / amsgrad), in place on whatever rows it is handed."""
    import math
    for e in entries:
        p, g, m, v = e["param"], e["grad"], e["exp_avg"], e["exp_avg_sq"]
        b1, b2 = e["betas"]
        m.lerp_(g, 1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** e["step"], 1 - b2 ** e["step"]
        denom = (v.sqrt() / math.sqrt(bc2)).add_(e["eps"])
        p.addcdiv_(m, denom, value=-e["lr"] / bc1)


def _worker_sharded(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    vdist.init_from_env(backend="gloo")
    out = {}
    for P in (40, 41):          # rows divide evenly over the ranks / they do not (padded collectives)
        M, deg = 16, 3

        def fresh():
            g = torch.Generator().manual_seed(5)
            return {"xyz": torch.randn(P, 3, generator=g).requires_grad_(True), "f_dc": torch.randn(P, 1, 3, generator=g).requires_grad_(True),
                    "f_rest": torch.randn(P, M - 1, 3, generator=g).requires_grad_(True), "opa": torch.rand(P, 1, generator=g).requires_grad_(True)}

        def groups(s):
            return [{"params": [s["xyz"]], "lr": 1e-2, "name": "xyz"}, {"params": [s["f_dc"]], "lr": 3e-3, "name": "f_dc"},
                    {"params": [s["f_rest"]], "lr": 2e-4, "name": "f_rest"}, {"params": [s["opa"]], "lr": 5e-2, "name": "opacity"}]
        A, B = fresh(), fresh()          # A: sharded step; B: today's path (ViewParallelReducer + the same Adam on every row, on every rank)
        optA = torch.optim.Adam(groups(A), lr=0.0, eps=1e-15); optB = torch.optim.Adam(groups(B), lr=0.0, eps=1e-15)
        sh = vdist.ShardedOptimizerStep(optA, sh=[A["f_dc"], A["f_rest"]], means3D=A["xyz"], expand=_expand_ref, adam=_adam_ref)
        red = vdist.ViewParallelReducer([B["xyz"], B["f_dc"], B["f_rest"], B["opa"]], sh=[B["f_dc"], B["f_rest"]], means3D=B["xyz"], expand=_expand_ref)
        gr = torch.Generator().manual_seed(100 + rank)
        for step in range(3):
            views = [{"dcolor": torch.randn(P, 3, generator=gr), "campos": torch.randn(3, generator=gr) * 4.0, "degree": deg, "M": M} for _ in range(2)]
            gx, go = torch.randn(P, 3, generator=gr), torch.randn(P, 1, generator=gr)
            extra = torch.randn(P, 1, 3, generator=gr) if step == 1 else None          # another loss term on an SH leaf: a dense gradient on top of the factored one
            for s, drv in ((A, sh), (B, red)):
                with drv.capture() as sink:
                    for it in views:
                        sink.append(dict(it))
                s["xyz"].grad = gx.clone(); s["opa"].grad = go.clone()
                if extra is not None:
                    s["f_dc"].grad = extra.clone()
            sh.step()
            red.reduce()
            entries = []
            for gdict in optB.param_groups:
                p = gdict["params"][0]
                st = optB.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0); st["exp_avg"] = torch.zeros_like(p); st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                entries.append({"param": p.data, "grad": p.grad, "exp_avg": st["exp_avg"], "exp_avg_sq": st["exp_avg_sq"], "lr": gdict["lr"],
                                "betas": gdict["betas"], "eps": gdict["eps"], "step": float(st["step"])})
            _adam_ref(entries)
            for p in B.values():
                p.grad = None
        chunk, lo, hi = vdist._rows(P, world, rank)
        stale = {k: optA.state[A[k]]["exp_avg"].clone() for k in A}
        sh.gather_state()
        out[P] = {"A": {k: v.detach().clone() for k, v in A.items()}, "B": {k: v.detach().clone() for k, v in B.items()},
                  "mA": {k: optA.state[A[k]]["exp_avg"].clone() for k in A}, "mB": {k: optB.state[B[k]]["exp_avg"].clone() for k in B},
                  "vA": {k: optA.state[A[k]]["exp_avg_sq"].clone() for k in A}, "vB": {k: optB.state[B[k]]["exp_avg_sq"].clone() for k in B},
                  "stale_other_rows_untouched": all(not stale[k][:lo].any() and not stale[k][hi:].any() for k in A), "rows": (lo, hi), "bytes": sh.last_bytes}
    # a rank whose captured views disagree makes EVERY rank raise before a collective starts
    s = {"xyz": torch.randn(8, 3).requires_grad_(True)}
    opt = torch.optim.Adam([{"params": [s["xyz"]], "lr": 1e-2}])
    bad = vdist.ShardedOptimizerStep(opt, means3D=s["xyz"], expand=_expand_ref, adam=_adam_ref)
    s["xyz"].grad = torch.randn(8, 3)
    with bad.capture() as sink:
        for _ in range(rank):
            sink.append({"dcolor": torch.randn(8, 3), "campos": torch.randn(3), "degree": 0, "M": 1})
    try:
        bad.step(); out["bad"] = "no error"
    except RuntimeError as ex:
        out["bad"] = str(ex)
    torch.save(out, os.path.join(out_dir, "s%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_optimizer_step_equals_the_replicated_step(tmp_path):
    world = 2
    mp.spawn(_worker_sharded, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, "s%d.pt" % r)) for r in range(world)]
    for P in (40, 41):
        for k in ("xyz", "f_dc", "f_rest", "opa"):
            for r in res:
                assert torch.equal(r[P]["A"][k], r[P]["B"][k]), (P, k)          # sharded == replicated, bit for bit (two ranks: a + b = b + a)
                assert torch.equal(r[P]["mA"][k], r[P]["mB"][k]) and torch.equal(r[P]["vA"][k], r[P]["vB"][k]), (P, k)      # ... the gathered moments too
            assert torch.equal(res[0][P]["A"][k], res[1][P]["A"][k])            # identical replicas
        assert all(r[P]["stale_other_rows_untouched"] for r in res)              # before gather_state a rank only holds its own rows' moments
        assert res[0][P]["rows"][1] == res[1][P]["rows"][0] and res[1][P]["rows"][1] == P
    # rank 1 captured a view (and has no SH leaf for it), rank 0 none: rank 1 reports its local error, rank 0 raises too instead of waiting in a collective
    assert "sh` leaves hold 0 coefficients" in res[1]["bad"] and "another rank failed its local checks" in res[0]["bad"], [r["bad"] for r in res]


# ---- round 5: world 4 -- the sharded step equals ONE process accumulating the four ranks' views in rank order, bit for bit ----------------------
def _worker_sharded4(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    vdist.init_from_env(backend="gloo")
    out = {}
    M, deg = 16, 3
    for P in (64, 67):          # rows divide evenly over the four ranks / they do not
        def fresh():
            g = torch.Generator().manual_seed(9)
            return {"xyz": torch.randn(P, 3, generator=g).requires_grad_(True), "f_dc": torch.randn(P, 1, 3, generator=g).requires_grad_(True),
                    "f_rest": torch.randn(P, M - 1, 3, generator=g).requires_grad_(True), "opa": torch.rand(P, 1, generator=g).requires_grad_(True)}

        def groups(s):
            return [{"params": [s["xyz"]], "lr": 1e-2, "name": "xyz"}, {"params": [s["f_dc"]], "lr": 3e-3, "name": "f_dc"},
                    {"params": [s["f_rest"]], "lr": 2e-4, "name": "f_rest"}, {"params": [s["opa"]], "lr": 5e-2, "name": "opacity"}]
        A, S = fresh(), fresh()          # A: the sharded step over 4 ranks; S: ONE process that sees every rank's views, in rank order
        optA = torch.optim.Adam(groups(A), lr=0.0, eps=1e-15); optS = torch.optim.Adam(groups(S), lr=0.0, eps=1e-15)
        sh = vdist.ShardedOptimizerStep(optA, sh=[A["f_dc"], A["f_rest"]], means3D=A["xyz"], expand=_expand_ref, adam=_adam_ref)
        gens = [torch.Generator().manual_seed(300 + r) for r in range(world)]          # every worker can replay every rank's random stream
        for step in range(3):
            per_rank = []
            for r in range(world):
                gr = gens[r]
                views = [{"dcolor": torch.randn(P, 3, generator=gr), "campos": torch.randn(3, generator=gr) * 4.0, "degree": deg, "M": M} for _ in range(2)]
                per_rank.append((views, torch.randn(P, 3, generator=gr), torch.randn(P, 1, generator=gr)))
            views, gx, go = per_rank[rank]
            with sh.capture() as sink:
                for it in views:
                    sink.append(dict(it))
            A["xyz"].grad = gx.clone(); A["opa"].grad = go.clone()
            sh.step()
            # the single process: gradients added view by view, rank 0's first
            gx_s, go_s = per_rank[0][1].clone(), per_rank[0][2].clone()
            for r in range(1, world):
                gx_s += per_rank[r][1]; go_s += per_rank[r][2]
            all_views = [it for r in range(world) for it in per_rank[r][0]]
            g_sh = _expand_ref(S["xyz"].detach(), torch.stack([it["campos"] for it in all_views]), torch.stack([it["dcolor"] for it in all_views]), deg, M)
            grads = {"xyz": gx_s, "f_dc": g_sh[:, :1].contiguous(), "f_rest": g_sh[:, 1:].contiguous(), "opa": go_s}
            entries = []
            for gdict in optS.param_groups:
                p = gdict["params"][0]
                st = optS.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0); st["exp_avg"] = torch.zeros_like(p); st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                key = [k for k, v in S.items() if v is p][0]
                entries.append({"param": p.data, "grad": grads[key], "exp_avg": st["exp_avg"], "exp_avg_sq": st["exp_avg_sq"], "lr": gdict["lr"],
                                "betas": gdict["betas"], "eps": gdict["eps"], "step": float(st["step"])})
            _adam_ref(entries)
        sh.gather_state()
        out[P] = {"A": {k: v.detach().clone() for k, v in A.items()}, "S": {k: v.detach().clone() for k, v in S.items()},
                  "mA": {k: optA.state[A[k]]["exp_avg"].clone() for k in A}, "mS": {k: optS.state[S[k]]["exp_avg"].clone() for k in S}}
    torch.save(out, os.path.join(out_dir, "q%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_step_at_world_4_equals_one_process_accumulating_in_rank_order(tmp_path):
    world = 4
    mp.spawn(_worker_sharded4, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(os.path.join(tmp_path, "q%d.pt" % r)) for r in range(world)]
    for P in (64, 67):
        for k in ("xyz", "f_dc", "f_rest", "opa"):
            for r in res:
                assert torch.equal(r[P]["A"][k], r[P]["S"][k]), (P, k)          # four ranks == one process, bit for bit: the owner adds in rank order
                assert torch.equal(r[P]["mA"][k], r[P]["mS"][k]), (P, k)
            for r in res[1:]:
                assert torch.equal(r[P]["A"][k], res[0][P]["A"][k]), (P, k)      # identical replicas


# ---- round 6: force=True issues every collective in a process group of ONE rank (what tests/test_gpu_rccl_world1.py runs over nccl on the GPU box) ----------
def _worker_forced(rank, world, port, out_dir):
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = vdist.init_from_env(backend="gloo", force=True)
    assert (r, w) == (0, 1) and dist.is_initialized()
    P, M, deg = 33, 16, 3
    out = {}
    for force in (False, True):
        g = torch.Generator().manual_seed(0)
        xyz = torch.randn(P, 3, generator=g).requires_grad_(True)
        f_dc = torch.randn(P, 1, 3, generator=g).requires_grad_(True); f_rest = torch.randn(P, M - 1, 3, generator=g).requires_grad_(True)
        opa = torch.rand(P, 1, generator=g).requires_grad_(True)
        red = vdist.ViewParallelReducer([xyz, f_dc, f_rest, opa], sh=[f_dc, f_rest], means3D=xyz, expand=_expand_ref, force=force)
        assert (red._agree is not None) == force
        gr = torch.Generator().manual_seed(100)
        with red.capture() as sink:
            for v in range(2):
                sink.append({"dcolor": torch.randn(P, 3, generator=gr), "campos": torch.randn(3, generator=gr) * 4.0, "degree": deg, "M": M})
        xyz.grad = torch.randn(P, 3, generator=gr); opa.grad = torch.randn(P, 1, generator=gr)
        red.reduce()
        out[force] = {"xyz": xyz.grad.clone(), "opa": opa.grad.clone(), "sh": torch.cat([f_dc.grad, f_rest.grad], dim=1), "agreements": red.n_agreements, "bytes": red.last_bytes}
        # the sharded step: all-to-all / reduce-scatter / all-gather at world 1
        for ordered in (True, False):
            s = {"xyz": torch.randn(P, 3, generator=g).requires_grad_(True), "opa": torch.rand(P, 1, generator=g).requires_grad_(True)}
            opt = torch.optim.Adam([{"params": [s["xyz"]], "lr": 1e-2}, {"params": [s["opa"]], "lr": 5e-2}], lr=0.0, eps=1e-15)
            sh = vdist.ShardedOptimizerStep(opt, means3D=s["xyz"], expand=_expand_ref, adam=_adam_ref, ordered=ordered, force=force)
            for k in s:
                s[k].grad = torch.randn(s[k].shape, generator=gr)
            sh.step()
            assert sh.state_is_gathered == (not force)
            sh.gather_state()
            out[force]["sharded_%d" % ordered] = {k: v.detach().clone() for k, v in s.items()}
        st = vdist.allreduce_densification_stats(torch.randn(P, 3, generator=gr), torch.rand(P, 3, generator=gr), torch.arange(P, dtype=torch.int32) % 3, force=force)
        out[force]["stats"] = [t.clone() for t in st]
        vdist.assert_replicas_identical([xyz], force=force)
    torch.save(out, os.path.join(out_dir, "forced.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_forced_exchange_at_world_one_is_the_identity(tmp_path):
    mp.spawn(_worker_forced, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    res = torch.load(os.path.join(tmp_path, "forced.pt"))
    assert res[True]["agreements"] == 1 and res[False]["agreements"] == 0 and res[True]["bytes"] > 0 and res[False]["bytes"] == 0
    for k in ("xyz", "opa", "sh"):
        assert torch.equal(res[True][k], res[False][k]), k
    for o in (0, 1):
        for k in ("xyz", "opa"):
            assert torch.equal(res[True]["sharded_%d" % o][k], res[False]["sharded_%d" % o][k])
    for a, b in zip(res[True]["stats"], res[False]["stats"]):
        assert torch.equal(a, b)
