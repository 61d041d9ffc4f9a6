#!/usr/bin/env python3
"""The reference's training schedule in miniature (train.py:260-430) on a synthetic multi-view scene, wired from this repository's
pieces only:

  * `renderer.render` with the plane-map glue fused into the preprocess kernels; colour only during the warm-up, `render_geo=True`
    afterwards (train.py:289-292), with the multi-view terms on the warped source colours and the median depth;
  * the cached source depth maps refreshed from the CURRENT Gaussians every `--depth-refresh` iterations in ONE batched depth-only
    pass (`renderer.render_depth_batch`; the reference overwrites the cache view by view, train.py:298-299);
  * densification every `--densify-every` iterations between `--densify-from` and `--densify-until` (train.py:400-418): the per-view
    statistics go through `dist.allreduce_densification_stats`, the point set changes through `densify.prune_and_extend_optimizer`
    (parameters AND Adam moments in one pass), the sampled offsets of the copies come from generators that every rank seeds alike
    (`dist.synchronized_densification_rng`).  The POLICY here is a small stand-in (double the Gaussians with the largest mean
    screen-space gradient, prune the transparent ones) -- the reference's own policy is plain torch code outside the rasterizer path;
  * `FusedAdam` with the reference's per-group learning rates (arguments/__init__.py);
  * plane-like Gaussians (one axis 10^-2 .. 10^-3 of the others: what IBGS / PGSR train towards) unless `--anisotropy none`;
  * under torchrun: one view per GPU, gradients exchanged by `ViewParallelReducer`.

An example and integration check, not a trainer: no SSIM, no appearance model, no colour network.

    python examples/train_synthetic.py --iters 200
    python -m torch.distributed.run --standalone --nproc-per-node 8 examples/train_synthetic.py --iters 200
"""
import argparse
import gc
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibgs_amd import densify, dist as vdist, renderer, simple_scene, synthetic as syn  # noqa: E402
from ibgs_amd.losses import l1_loss  # noqa: E402
from ibgs_amd.optim import FusedAdam  # noqa: E402

GROUPS = (("xyz", "_xyz", 1.6e-4), ("f_dc", "_features_dc", 2.5e-3), ("f_rest", "_features_rest", 2.5e-3 / 20.0), ("opacity", "_opacity", 5e-2),
          ("scaling", "_scaling", 5e-3), ("rotation", "_rotation", 1e-3), ("normal", "_normal", 1e-3), ("offset", "_offset", 1e-3))


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--anisotropy", default="plane", choices=["none", "plane", "needle", "mixed"])
    ap.add_argument("--geo-from", type=int, default=30, help="iteration from which render_geo is on (train.py: 7000 - 2 * #cameras)")
    ap.add_argument("--depth-refresh", type=int, default=16, help="refresh every cached source depth map from the current Gaussians every this many iterations (0 = never)")
    ap.add_argument("--densify-every", type=int, default=0, help="0 = no densification")
    ap.add_argument("--densify-from", type=int, default=20)
    ap.add_argument("--densify-until", type=int, default=10 ** 9)
    ap.add_argument("--sharded-optimizer", action="store_true",
                    help="more than one rank: reduce-scatter + Adam on the rank's rows + all-gather (dist.ShardedOptimizerStep) instead of all-reduce + the same Adam on every rank")
    ap.add_argument("--quiet", action="store_true")
    return ap.parse_args(argv)


def psnr(a, b):
    return float(20.0 * math.log10(1.0 / math.sqrt(float(((a - b) ** 2).mean()) + 1e-20)))


def densify_step(pc, opt, stats, it, world, frac=0.03, min_opacity=0.005, seed=0):
    """Stand-in policy on the reference's primitives (densify_and_clone / densify_and_split / prune, scene/gaussian_model.py:484-597):
    the `frac` of the Gaussians with the largest mean screen-space gradient are doubled -- the copy sits a sampled fraction of a
    standard deviation away (the samples come from generators every rank seeds alike), both halves get the opacity that composites to
    the old one, so the image barely moves -- and what has become transparent is pruned.  `stats` = [gradient accumulator (P, 1),
    visibility count (P, 1)], masked / extended alongside the parameters and the Adam moments in ONE pass."""
    accum, denom = stats
    P = pc._xyz.shape[0]
    avg = (accum / denom.clamp_min(1.0)).squeeze(-1)
    k = max(1, int(frac * P))
    thr = torch.topk(avg, k).values[-1]
    sel = (avg >= thr) & (denom.squeeze(-1) > 0)
    n = int(sel.sum())
    with torch.no_grad():
        o = torch.sigmoid(pc._opacity[sel])
        half = (1.0 - torch.sqrt(1.0 - o.clamp(max=1.0 - 1e-6))).clamp(1e-6, 1.0 - 1e-6)      # 1 - (1 - o')^2 = o
        raw_half = torch.log(half / (1.0 - half))
        pc._opacity.data[sel] = raw_half
        with vdist.synchronized_densification_rng(it, base_seed=seed):                # identical samples on every rank, generators restored afterwards
            std = 0.3 * torch.exp(pc._scaling[sel])
            samples = torch.normal(mean=torch.zeros_like(std), std=std)
        new_xyz = torch.bmm(pc.rotation_matrices()[sel], samples.unsqueeze(-1)).squeeze(-1) + pc._xyz[sel]
        ext = {"xyz": new_xyz, "f_dc": pc._features_dc[sel], "f_rest": pc._features_rest[sel], "opacity": raw_half, "scaling": pc._scaling[sel],
               "rotation": pc._rotation[sel], "normal": pc._normal[sel], "offset": pc._offset[sel]}
        ext = {k_: v.detach().contiguous() for k_, v in ext.items()}
        keep = torch.sigmoid(pc._opacity).squeeze(-1) > min_opacity
    new, extra = densify.prune_and_extend_optimizer(opt, keep, ext, extra=[accum, denom])
    for name, attr, _ in GROUPS:
        setattr(pc, attr, new[name])
    extra[0].zero_(); extra[1].zero_()                                             # reset like densification_postfix (gaussian_model.py:463-466)
    return extra, n, int((~keep).sum())


def run(a, rasterize=None):
    """One training run; returns {"loss": [...], "psnr": [...], "points": [...], "ms_per_iter": float}.  `rasterize` replaces
    ibgs_amd.rasterizer.rasterize_gaussians for the run (tests drive the same loop with the oracle's gradients)."""
    from ibgs_amd import rasterizer as _rast
    rank, world, local = vdist.init_from_env(backend=os.environ.get("IBGS_DIST_BACKEND"))      # default nccl (= RCCL); gloo lets two ranks share one GPU
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)
    orig_rasterize = _rast.rasterize_gaussians

    # ground truth scene -> target images; the model starts from a perturbed copy
    aniso = None if a.anisotropy == "none" else a.anisotropy
    gt = syn.make_gaussians(a.points, 5, sh_degree=3, max_coeffs=16, opacity="trained", extent=0.9, anisotropy=aniso)
    gt["scales"] = (gt["scales"] * 1.5).astype(np.float32)
    cams = simple_scene.orbit_cameras(a.width, a.height, n_views=a.views, device=dev, nearest=3)
    pipe, args = simple_scene.default_pipe(), simple_scene.default_args()
    bg = torch.zeros(3, device=dev)
    truth = simple_scene.SimpleGaussians(gt, sh_degree=3, device=dev)
    scene = simple_scene.SimpleScene(cams, device=dev)
    with torch.no_grad():
        targets = torch.stack([renderer.render(c, truth, scene, pipe, args, bg, False, 3, 4, render_geo=False, return_depth_normal=False)["render"] for c in cams])
        scene.original_image_list = targets.clone()
    rng = np.random.default_rng(11)
    init = dict(gt)
    init["means3D"] = (gt["means3D"] + rng.normal(0, 0.02, gt["means3D"].shape)).astype(np.float32)
    init["shs"] = (gt["shs"] + rng.normal(0, 0.2, gt["shs"].shape)).astype(np.float32)
    init["opacities"] = np.clip(gt["opacities"] * 0.7 + 0.1, 0.02, 0.98).astype(np.float32)
    pc = simple_scene.SimpleGaussians(init, sh_degree=3, device=dev)
    opt = FusedAdam([{"params": [getattr(pc, attr)], "lr": lr, "name": name} for name, attr, lr in GROUPS], lr=0.0, eps=1e-15)
    params = lambda: [g["params"][0] for g in opt.param_groups]          # densification replaces the Parameters: always ask the optimiser
    red = vdist.ViewParallelReducer(params, sh=lambda: [pc._features_dc, pc._features_rest], means3D=lambda: pc._xyz) if world > 1 else None
    sharded = None
    if world > 1 and a.sharded_optimizer:          # the same sums and the same update, the optimiser's work split over the ranks
        sharded = vdist.ShardedOptimizerStep(opt, sh=lambda: [pc._features_dc, pc._features_rest], means3D=lambda: pc._xyz)
        red = None
    stats = [torch.zeros(a.points, 1, device=dev), torch.zeros(a.points, 1, device=dev)]

    def refresh_depths():
        with torch.no_grad():
            scene.rendered_depth_list = renderer.render_depth_batch(cams, pc, scene, pipe, args, bg, False, 3, 4).detach()

    hist = {"loss": [], "psnr": [], "points": [], "split": 0, "pruned": 0}
    if rasterize is not None:
        _rast.rasterize_gaussians = rasterize
    try:
        gc.collect(); gc.freeze()          # keep full collections (33-42 ms with torch imported) out of a 2 ms iteration
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for it in range(a.iters):
            vid = vdist.views_for_rank(it, rank, world, len(cams))
            cam = cams[vid]
            geo = it >= a.geo_from
            if geo and (it == a.geo_from or (a.depth_refresh and (it - a.geo_from) % a.depth_refresh == 0)):
                refresh_depths()           # the source depth maps follow the model (one batched depth-only pass for all cameras)
            opt.zero_grad(set_to_none=True)
            box = {}

            def fwd_bwd():
                out = renderer.render(cam, pc, scene, pipe, args, bg, learnt_normal=False, nb_src_frames=3, buffer_length=4,
                                      render_geo=geo, return_depth_normal=False)
                loss = l1_loss(out["render"], targets[vid])
                if geo:   # multi-view terms in the spirit of train.py:319-338: warped source colours should match the image where a source is valid
                    m = (out["cam_feat"][3:4] != 0).float()
                    loss = loss + 0.05 * ((out["warped_image"][0:3] - targets[vid]).abs() * m).mean()
                loss.backward()
                box.update(out=out)
                return loss

            if sharded is not None:
                with sharded.capture():
                    loss = fwd_bwd()
            elif red is not None:
                with red.capture():
                    loss = fwd_bwd()
                red.reduce(average=True)
            else:
                loss = fwd_bwd()
            out = box["out"]
            if a.densify_every and it < a.densify_until:          # train.py:400-410: statistics of every view of the step
                gn, _gna, cnt, _rmax = vdist.allreduce_densification_stats(out["viewspace_points"].grad, out["viewspace_points_abs"].grad, out["radii"])
                stats[0] += gn; stats[1] += cnt
            if sharded is not None:
                sharded.step(average=True)          # reduce-scatter, Adam on this rank's rows, all-gather of the parameters
            else:
                opt.step()
            hist["loss"].append(float(loss.detach())); hist["psnr"].append(psnr(out["render"].detach(), targets[vid])); hist["points"].append(int(pc._xyz.shape[0]))
            if a.densify_every and a.densify_from <= it < a.densify_until and (it + 1) % a.densify_every == 0:
                if sharded is not None:
                    sharded.gather_state()          # the surgery below reads and reshapes the moments of EVERY row
                stats, n_split, n_pruned = densify_step(pc, opt, stats, it, world)
                hist["split"] += n_split; hist["pruned"] += n_pruned
                if world > 1:
                    vdist.assert_replicas_identical(params(), what="parameters after densification")
            if not a.quiet and rank == 0 and (it % 20 == 0 or it == a.iters - 1):
                print("iter %4d  view %d  geo %d  points %d  loss %.5f  psnr %.2f" % (it, vid, geo, hist["points"][-1], hist["loss"][-1], hist["psnr"][-1]), flush=True)
        torch.cuda.synchronize()
        hist["ms_per_iter"] = (time.perf_counter() - t0) / max(a.iters, 1) * 1e3
    finally:
        _rast.rasterize_gaussians = orig_rasterize
        gc.unfreeze()
    if rank == 0:
        k = max(1, min(10, a.iters // 4))
        print("train_synthetic: %d iterations, %.2f ms / iteration, loss %.5f -> %.5f, psnr %.2f -> %.2f dB, points %d -> %d (split %d, pruned %d)"
              % (a.iters, hist["ms_per_iter"], float(np.mean(hist["loss"][:k])), float(np.mean(hist["loss"][-k:])), float(np.mean(hist["psnr"][:k])),
                 float(np.mean(hist["psnr"][-k:])), a.points, hist["points"][-1] if hist["points"] else a.points, hist["split"], hist["pruned"]))
    if world > 1:
        # replicas must stay bit-identical: same summed gradients on every rank, same optimiser step, same densification samples
        cs = torch.stack([p.detach().double().sum() for p in params()])
        lo, hi = cs.clone(), cs.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN); torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        if rank == 0:
            print("replicas in sync: %s" % bool(torch.equal(lo, hi)))
        torch.distributed.barrier(); torch.distributed.destroy_process_group()
    return hist


def main():
    return run(parse())


if __name__ == "__main__":
    main()
