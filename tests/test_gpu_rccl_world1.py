"""SURVEY 8(e) on the one GPU a test box has: the view-parallel exchange over the `nccl` backend (= RCCL) in a process group of ONE rank.

Every collective of the N-GPU step is issued (`force=True`: the gloo agreement twin beside nccl, the all-gather of the dL/dRGB factors, the flat all-reduce,
all_to_all_single / reduce_scatter_tensor / all_gather_into_tensor of the sharded optimiser step, the densification-statistics reductions); at world size 1 each
of them is the identity, so the results must equal the local ones BIT FOR BIT -- what is tested is that the product code runs on its real backend, on device
tensors, with the bucket-resident gradient sinks under RCCL's streams.  (World sizes 2 and 4 are covered over gloo on the CPU: tests/test_dist_gloo.py.)

One process group per test process: the module initialises it once and destroys it at the end."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from ibgs_amd import dist as vdist, rasterizer, synthetic as syn
from ibgs_amd.optim import FusedAdam
from tests import hipref

pytestmark = pytest.mark.gpu
KEYS = ("means3D", "shs", "opacities", "scales", "rotations")


@pytest.fixture(scope="module")
def nccl_world1():
    assert not dist.is_initialized(), "another test left a process group behind"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    old = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    rank, world, _ = vdist.init_from_env(backend="nccl", force=True)
    assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl"
    yield
    dist.barrier()
    dist.destroy_process_group()
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _views(P, W, H, deg, n_views):
    out = []
    for v in range(n_views):
        inp = syn.make_scene(P, W, H, sh_degree=deg, seed=21, view=v, opacity="trained")
        inp["shs"] = (inp["shs"] * 3.0).astype(np.float32)
        out.append(inp)
    return out


def _backward_views(lv, views, H, W):
    for v, inp in enumerate(views):
        st = hipref.settings_from(inp, "cuda")
        outs = rasterizer.GaussianRasterizer(st)(means3D=lv["means3D"], means2D=lv["means2D"], means2D_abs=lv["means2D_abs"], opacities=lv["opacities"],
                                                 shs=lv["shs"], scales=lv["scales"], rotations=lv["rotations"])
        tgt = torch.rand(3, H, W, device="cuda", generator=torch.Generator(device="cuda").manual_seed(v))
        (outs[0] - tgt).abs().mean().backward()


@pytest.mark.parametrize("factored", [True, False])
def test_reducer_over_rccl_at_world_one_returns_the_local_gradients(nccl_world1, factored):
    """ViewParallelReducer(force=True) over nccl: agreement on the gloo twin, all-gather + all-reduce on RCCL.  Deterministic backward on both sides, so the
    forced exchange must hand back exactly the bits of the unforced (collective-free) reduce."""
    P, W, H, n_views = 3000, 128, 96, 2
    views = _views(P, W, H, 3, n_views)
    old = rasterizer.DETERMINISTIC
    rasterizer.DETERMINISTIC = True
    try:
        got = {}
        for force in (False, True):
            lv = hipref.leaf_inputs(views[0], "cuda")
            red = vdist.ViewParallelReducer([lv[k] for k in KEYS], sh=lv["shs"], means3D=lv["means3D"], factored=factored, force=force,
                                            direct={k: lv[k] for k in ("means3D", "opacities", "scales", "rotations")})
            if force:
                assert red._agree is not None and dist.get_backend(red._agree) == "gloo", "the host agreement runs on a gloo twin beside nccl"
            with red.capture():
                _backward_views(lv, views, H, W)
            red.reduce()
            torch.cuda.synchronize()
            if force:
                assert red.n_agreements == 1 and red.last_bytes > 0
                if factored:
                    assert red.last_bytes == sum(lv[k].numel() for k in KEYS if k != "shs") * 4 + n_views * (P + 1) * 12
            got[force] = {k: lv[k].grad.detach().clone() for k in KEYS}
        for k in KEYS:
            assert got[True][k].abs().sum() > 0, k
            assert torch.equal(got[True][k], got[False][k]), "%s: the forced exchange changed bits" % k
    finally:
        rasterizer.DETERMINISTIC = old


def _model(P, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    mk = lambda *s: torch.randn(*s, device="cuda", generator=g).requires_grad_(True)
    return {"xyz": mk(P, 3), "f_dc": mk(P, 1, 3), "f_rest": mk(P, 15, 3), "opa": mk(P, 1), "scale": mk(P, 3), "rot": mk(P, 4)}


@pytest.mark.parametrize("ordered", [True, False])
def test_sharded_step_over_rccl_at_world_one_equals_fused_adam(nccl_world1, ordered):
    """ShardedOptimizerStep(force=True): all_to_all_single (ordered) / reduce_scatter_tensor, Adam on the rank's rows (= all rows), all_gather_into_tensor -- against
    FusedAdam.step() on the same gradients: parameters and both moments bit for bit, three steps, P not a multiple of anything."""
    P = 4099
    lrs = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opa": 5e-2, "scale": 5e-3, "rot": 1e-3}
    A, B = _model(P, 5), _model(P, 5)
    optA = FusedAdam([{"params": [A[k]], "lr": lrs[k], "name": k} for k in A], lr=0.0, eps=1e-15)
    optB = FusedAdam([{"params": [B[k]], "lr": lrs[k], "name": k} for k in B], lr=0.0, eps=1e-15)
    sh = vdist.ShardedOptimizerStep(optA, sh=[A["f_dc"], A["f_rest"]], means3D=A["xyz"], ordered=ordered, force=True)
    assert sh._agree is not None and dist.get_backend(sh._agree) == "gloo"
    gg = torch.Generator(device="cuda").manual_seed(77)
    for it in range(3):
        grads = {k: torch.randn(A[k].shape, device="cuda", generator=gg) for k in A}
        for k in A:
            A[k].grad = grads[k].clone(); B[k].grad = grads[k].clone()
        sh.step()
        optB.step()
        assert sh.last_bytes == 0          # (world - 1) x bytes: nothing leaves a group of one -- but every collective was issued
        assert not sh.state_is_gathered
        sh.gather_state()
        for k in A:
            assert torch.equal(A[k].data, B[k].data), (it, k)
            assert torch.equal(optA.state[A[k]]["exp_avg"], optB.state[B[k]]["exp_avg"]) and torch.equal(optA.state[A[k]]["exp_avg_sq"], optB.state[B[k]]["exp_avg_sq"]), (it, k)


def test_sharded_step_with_factored_views_over_rccl(nccl_world1):
    """... and with the SH gradient arriving as dL/dRGB factors from real backward passes (all_to_all_single of the factors, all-gather of the camera centres,
    expansion on the device): equal to the unforced step, bit for bit (deterministic backward)."""
    P, W, H = 2500, 112, 80
    views = _views(P, W, H, 3, 2)
    old = rasterizer.DETERMINISTIC
    rasterizer.DETERMINISTIC = True
    try:
        res = {}
        for force in (False, True):
            lv = hipref.leaf_inputs(views[0], "cuda")
            leaves = [lv[k] for k in KEYS]
            opt = FusedAdam([{"params": [p], "lr": 1e-3, "name": k} for k, p in zip(KEYS, leaves)], lr=0.0, eps=1e-15)
            sh = vdist.ShardedOptimizerStep(opt, sh=[lv["shs"]], means3D=lv["means3D"], force=force)
            with sh.capture():
                _backward_views(lv, views, H, W)
            sh.step()
            sh.gather_state()
            torch.cuda.synchronize()
            res[force] = {k: lv[k].detach().clone() for k in KEYS}
        for k in KEYS:
            assert torch.equal(res[True][k], res[False][k]), k
    finally:
        rasterizer.DETERMINISTIC = old


def test_densification_statistics_and_replica_guard_over_rccl(nccl_world1):
    P = 5000
    g = torch.Generator(device="cuda").manual_seed(3)
    vg = torch.randn(P, 3, device="cuda", generator=g)
    radii = torch.randint(0, 40, (P,), device="cuda", generator=g, dtype=torch.int32)
    a = vdist.allreduce_densification_stats(vg, vg.abs(), radii, force=False)
    b = vdist.allreduce_densification_stats(vg, vg.abs(), radii, force=True)          # SUM and MAX all-reduces on RCCL
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    vdist.assert_replicas_identical([vg, radii.float()], force=True)          # MIN / MAX all-reduce of the checksums
    # plain allreduce_gradients keeps its world-1 early exit; the bucketed all-reduce itself is the reducer's (first test)
    t = torch.ones(7, device="cuda")
    dist.all_reduce(t)
    assert torch.equal(t, torch.ones(7, device="cuda"))
