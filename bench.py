#!/usr/bin/env python3
"""Headline benchmark of the plane rasterizer hot path (BASELINE.json metric):
train-step (rasterizer forward + backward) at 1920x1080 on 1M random-init Gaussians, SH degree 3
(BASELINE.json configs[2], "C3" of SURVEY.md 8(d)); whole-job views per second over N GPUs.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one view per rank: preprocess -> depth sort -> emit ->
tile sort -> render (forward), L1 loss against a fixed random target, render backward -> preprocess
backward, and for N > 1 one all-reduce (sum) of the Gaussian gradients over RCCL (view-parallel,
SURVEY.md 8(e)).  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ibgs_amd import _lib, synthetic as syn  # noqa: E402
from ibgs_amd import dist as vdist  # noqa: E402
from ibgs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402

HBM_PEAK = 8.0e12  # bytes/s, MI355X_MICROARCH.md "HBM3E peak BW" (spec)


def algorithmic_bytes(P, R, HW, Mc, tiles):
    """SURVEY.md 8(d) byte model (implementation independent). Returns (B_fwd, B_bwd, B_render_bwd)."""
    bit = int(np.ceil(np.log2(max(tiles, 2))))
    n_pass = -(-(32 + bit) // 8)
    b_fwd = P * (44 + 12 * Mc + 79) + R * (12 + 24 * n_pass + 8 + 40) + HW * 20
    b_render_bwd = HW * 20 + R * 40 + P * 112
    b_bwd = b_render_bwd + P * (44 + 12 * Mc + 56) + P * (12 * Mc + 12 + 12 + 16 + 24)
    return b_fwd, b_bwd, b_render_bwd


def build_inputs(cfg, view, dev, opacity):
    c = syn.CONFIGS[cfg]
    inp = syn.make_scene(c["P"], c["W"], c["H"], sh_degree=c["sh_degree"], seed=c["seed"], view=view, opacity=opacity)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    P = c["P"]
    leaves = {
        "means3D": t(inp["means3D"]).requires_grad_(True), "shs": t(inp["shs"]).requires_grad_(True),
        "opacities": t(inp["opacities"]).reshape(P, 1).requires_grad_(True),
        "scales": t(inp["scales"]).requires_grad_(True), "rotations": t(inp["rotations"]).requires_grad_(True),
        "means2D": torch.zeros(P, 3, device=dev, requires_grad=True),
        "means2D_abs": torch.zeros(P, 3, device=dev, requires_grad=True),
    }
    H, W = c["H"], c["W"]
    z = lambda *s: torch.zeros(*s, device=dev)
    st = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=float(inp["tanfovx"]), tanfovy=float(inp["tanfovy"]), bg=t(inp["bg"]),
        scale_modifier=1.0, viewmatrix=t(inp["viewmatrix"]), projmatrix=t(inp["projmatrix"]),
        ref_to_src_list=z(1, 16), src_cam_pos=z(1, 3), src_images=z(1, 3, 1), src_rendered_depths=z(1, 1, 1),
        nb_src_images=1, buffer_length=4, depth_error_threshold=0.01, sh_degree=c["sh_degree"], campos=t(inp["campos"]),
        prefiltered=False, render_geo=False, render_depth_only=False, debug=False)
    return inp, leaves, st, c


def cpu_baseline(inp, c):
    """The oracle (scalar C restatement, OpenMP over pixels) timed on the host cores: ONE full step
    (forward + backward) of the same workload.  Reported baseline only -- never the product path."""
    import oracle
    lib = oracle.lib()
    cores = int(lib.orc_num_threads())
    t0 = time.time()
    f = oracle.forward(inp)
    t1 = time.time()
    g = np.sign(f["color"] - 0.5).astype(np.float32) / f["color"].size
    oracle.backward(inp, f, g)
    t2 = time.time()
    return {"value": 1.0 / (t2 - t0), "unit": "fps", "cores": cores, "kind": "port",
            "sample": "1 full step (fwd %.2f s + bwd %.2f s) of the same %s workload, oracle/ibgs_oracle.c with OpenMP"
                      % (t1 - t0, t2 - t1, c)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C3", choices=list(syn.CONFIGS))
    ap.add_argument("--opacity", default="init", choices=["init", "trained"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", default="factored", choices=["factored", "dense"],
                    help="N > 1: factored = all-gather 3-float dL/dRGB per view + local SH expansion (default); dense = all-reduce of every gradient")
    ap.add_argument("--forward-only", action="store_true", help="time the forward render alone (BASELINE configs[1]: --config C2 --forward-only)")
    ap.add_argument("--geo", action="store_true", help="second line of SURVEY 8(d): render_geo=True, n_src=4, L=4")
    a = ap.parse_args()

    rank, world, local_rank = vdist.init_from_env(backend=os.environ.get("IBGS_DIST_BACKEND"))   # default: nccl (= RCCL)
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    local_rank %= max(torch.cuda.device_count(), 1)      # lets a 1-GPU box run the N > 1 code path over gloo
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    _lib.load()

    inp, leaves, st, c = build_inputs(a.config, rank % 8, dev, a.opacity)
    H, W, P = c["H"], c["W"], c["P"]
    if a.geo:
        # sources = the 4 nearest other orbit views; their images / depth maps are this op's own renders (untimed)
        tt = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)
        ref_cam = inp["_cam"]
        src_ids = [(rank + d) % 8 for d in (1, 7, 2, 6)]
        src_cams = [syn.make_camera(W, H, azimuth_deg=45.0 * k) for k in src_ids]
        imgs, deps = [], []
        with torch.no_grad():
            for sc in src_cams:
                am = tt(syn.plane_all_map(inp["means3D"], inp["scales"], inp["rotations"], sc))
                for depth_only in (False, True):
                    sst = st._replace(viewmatrix=tt(sc["viewmatrix"]), projmatrix=tt(sc["projmatrix"]), campos=tt(sc["campos"]),
                                      render_depth_only=depth_only)
                    o = GaussianRasterizer(sst)(means3D=leaves["means3D"], means2D=leaves["means2D"], means2D_abs=leaves["means2D_abs"],
                                                opacities=leaves["opacities"], shs=leaves["shs"], scales=leaves["scales"],
                                                rotations=leaves["rotations"], all_map=am)
                    (deps if depth_only else imgs).append(o[3].clone() if depth_only else o[0].clone())
        r2s, scp = syn.ref_to_src(ref_cam, src_cams)
        leaves["all_map"] = tt(syn.plane_all_map(inp["means3D"], inp["scales"], inp["rotations"], ref_cam)).requires_grad_(True)
        st = st._replace(ref_to_src_list=tt(r2s), src_cam_pos=tt(scp), src_images=torch.stack(imgs), src_rendered_depths=torch.stack(deps),
                         nb_src_images=4, buffer_length=4, depth_error_threshold=0.01, render_geo=True)
    rast = GaussianRasterizer(st)
    target = torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(1234 + rank))
    params = [leaves[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]
    bucket = vdist.GradBucket(params) if world > 1 else None
    reducer = vdist.ViewParallelReducer(params, sh=leaves["shs"], means3D=leaves["means3D"]) if (world > 1 and a.exchange == "factored") else None
    R_seen = [0]

    def fwd_only():
        with torch.no_grad():
            rast(means3D=leaves["means3D"], means2D=leaves["means2D"], means2D_abs=leaves["means2D_abs"],
                 opacities=leaves["opacities"], shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"],
                 all_map=leaves.get("all_map"))

    def step():
        if a.forward_only:
            return fwd_only()
        for v in leaves.values():
            v.grad = None
        outs = rast(means3D=leaves["means3D"], means2D=leaves["means2D"], means2D_abs=leaves["means2D_abs"],
                    opacities=leaves["opacities"], shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"],
                    all_map=leaves.get("all_map"))
        loss = torch.nn.functional.l1_loss(outs[0], target)
        if a.geo:   # every differentiable geo output takes part: normal map, median depth, warped source colours
            loss = loss + outs[2].abs().mean() + outs[3].abs().mean() + (outs[5] - 0.5).abs().mean()
        R_seen[0] = outs[0].grad_fn.num_rendered
        if reducer is not None:
            with reducer.capture():
                loss.backward()
            reducer.reduce()
        else:
            loss.backward()
            if world > 1:
                vdist.allreduce_gradients(params, bucket)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    _lib.timing_enable(["render_fwd" if a.forward_only else "render_bwd"])   # hipEvents around the dominant kernel only, on the op's stream
    _lib.timing_collect()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    tm = _lib.timing_collect()
    _lib.timing_enable([])
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # forward-only time of the same workload ("report fwd-only and fwd+bwd separately"), untimed for `value`
    fence()
    tf = time.perf_counter()
    for _ in range(10):
        fwd_only()
    fence()
    fwd_ms = (time.perf_counter() - tf) / 10 * 1e3

    # per-stage breakdown from a few extra (untimed) steps
    _lib.timing_enable(_lib.STAGES)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    stages = {k: (v[0] / max(v[1], 1)) * (v[1] / 3.0) for k, v in _lib.timing_collect().items()}
    _lib.timing_enable([])

    if rank == 0:
        ms_step = dt / a.steps * 1e3
        if a.forward_only:
            with torch.no_grad():
                o = rast(means3D=leaves["means3D"], means2D=leaves["means2D"], means2D_abs=leaves["means2D_abs"], opacities=leaves["opacities"],
                         shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"], all_map=leaves.get("all_map"))
            from ibgs_amd import rasterizer as _r
            R_seen[0] = _r.LAST_NUM_RENDERED
        R = int(R_seen[0]); HW = H * W; Mc = int(inp["shs"].shape[1]); tiles = ((W + 15) // 16) * ((H + 15) // 16)
        b_fwd, b_bwd, b_rbwd = algorithmic_bytes(P, R, HW, Mc, tiles)
        kstage = "render_fwd" if a.forward_only else "render_bwd"
        k_ms = tm[kstage][0] / max(tm[kstage][1], 1)
        if a.forward_only:
            b_rbwd = HW * 20 + R * 40          # the forward blend's share of B_fwd (records + per-pixel outputs)
        achieved = b_rbwd / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath) and a.config == "C3" and not (a.geo or a.forward_only or a.opacity != "init"):   # the profiled workload only
            try:
                traffic = json.load(open(tpath)).get("render_bwd_bytes_per_launch")
            except Exception:
                traffic = None
        # what the dominant kernel is actually bound by: wave-VALU instructions per launch from the committed PMC pass,
        # divided by this run's kernel time -> instructions per cycle per SIMD (the chip issues at most one per ~2.6 cycles)
        valu = None
        cpath = os.path.join(ROOT, "profiles", "r01_counters.json")
        if traffic is not None and os.path.exists(cpath) and k_ms > 0:
            try:
                ctr = json.load(open(cpath))["per_launch_counters"]["render_bwd_color_kernel"]
                ipc = ctr["SQ_INSTS_VALU"] / (1024.0 * k_ms * 1e-3 * 2.4e9)
                valu = {"wave_valu_insts_per_launch": ctr["SQ_INSTS_VALU"], "insts_per_cycle_per_simd": ipc, "cycles_per_inst": 1.0 / ipc,
                        "clock_ghz": 2.4, "simds": 1024}
            except Exception:
                valu = None
        out = {
            "metric": ("forward render fps " + a.config) if a.forward_only else
                      ("train-step fps (fwd+bwd raster) @1080p, 1M Gaussians" if a.config == "C3" else "train-step fps (fwd+bwd raster) " + a.config),
            "value": world * a.steps / dt, "unit": "fps", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %d random-init Gaussians, %dx%d, SH degree %d, rasterizer %s, "
                                   "opacity=%s%s, one view per GPU%s" % (a.config, P, W, H, c["sh_degree"], "forward only" if a.forward_only else "fwd+bwd, L1 loss vs fixed random target", a.opacity, ", render_geo n_src=4 L=4" if a.geo else "",
                                                                        (", RCCL gradient exchange (%s)" % a.exchange) if world > 1 else ""),
                       "num_rendered": R, "parallelism": "view-parallel x%d" % world},
            "forward_only_ms": fwd_ms,
            "roofline": {"bound": "hbm", "kernel": ("render_fwd_kernel" if a.forward_only else (("render_bwd_geo2_kernel" if tiles >= 4096 else "render_bwd_geo_kernel") if a.geo else ("render_bwd_color_kernel" if tiles >= 4096 else "render_bwd_color_small_kernel"))), "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved * 1e9 / HBM_PEAK, "traffic": traffic,
                         "algorithmic_bytes_per_launch": b_rbwd, "kernel_ms": k_ms,
                         "valu_issue": valu,
                         "mfma_utilisation": 0.0,        # by design: no kernel of this path is a dense contraction (DESIGN.md "MFMA")
                         "step_algorithmic_bytes": b_fwd if a.forward_only else b_fwd + b_bwd,
                         "step_frac": (b_fwd if a.forward_only else b_fwd + b_bwd) / (ms_step * 1e-3) / HBM_PEAK},
            "stages_ms": stages,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(inp, a.config)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
