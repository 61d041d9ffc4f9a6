#!/usr/bin/env python3
"""A miniature of the reference's training loop (train.py:260-430) on a synthetic scene, wired from this repository's pieces:
`renderer.render` (fused plane glue, geo outputs after a warm-up like train.py:289-292), the L1 photometric loss, `FusedAdam`
with the reference's per-group learning rates (arguments/__init__.py), and -- under torchrun -- one view per GPU with
`ViewParallelReducer`.  It is an example / integration check, not a trainer: no densification, no SSIM, no appearance model.

    python examples/train_synthetic.py --iters 200
    python -m torch.distributed.run --standalone --nproc-per-node 8 examples/train_synthetic.py --iters 200
"""
import argparse
import gc
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ibgs_amd import dist as vdist, renderer, simple_scene, synthetic as syn  # noqa: E402
from ibgs_amd.losses import l1_loss  # noqa: E402
from ibgs_amd.optim import FusedAdam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--geo-from", type=int, default=30, help="iteration from which render_geo is on (train.py: 7000 - 2 * #cameras)")
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    rank, world, local = vdist.init_from_env(backend=os.environ.get("IBGS_DIST_BACKEND"))      # default nccl (= RCCL); gloo lets two ranks share one GPU
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))
    torch.cuda.set_device(dev)

    # ground truth scene -> target images; the model starts from a perturbed copy
    gt = syn.make_gaussians(a.points, 5, sh_degree=3, max_coeffs=16, opacity="trained", extent=0.9)
    gt["scales"] = (gt["scales"] * 1.5).astype(np.float32)
    cams = simple_scene.orbit_cameras(a.width, a.height, n_views=a.views, device=dev, nearest=3)
    pipe, args = simple_scene.default_pipe(), simple_scene.default_args()
    bg = torch.zeros(3, device=dev)
    truth = simple_scene.SimpleGaussians(gt, sh_degree=3, device=dev)
    scene = simple_scene.SimpleScene(cams, device=dev)
    with torch.no_grad():
        targets = torch.stack([renderer.render(c, truth, scene, pipe, args, bg, False, 3, 4, render_geo=False, return_depth_normal=False)["render"] for c in cams])
        scene.original_image_list = targets.clone()
        for j, c in enumerate(cams):
            scene.rendered_depth_list[j] = renderer.render_depth(c, truth, scene, pipe, args, bg, False, 3, 4)
    rng = np.random.default_rng(11)
    init = dict(gt)
    init["means3D"] = (gt["means3D"] + rng.normal(0, 0.02, gt["means3D"].shape)).astype(np.float32)
    init["shs"] = (gt["shs"] + rng.normal(0, 0.2, gt["shs"].shape)).astype(np.float32)
    init["opacities"] = np.clip(gt["opacities"] * 0.7 + 0.1, 0.02, 0.98).astype(np.float32)
    pc = simple_scene.SimpleGaussians(init, sh_degree=3, device=dev)
    groups = [{"params": [pc._xyz], "lr": 1.6e-4, "name": "xyz"}, {"params": [pc._features_dc], "lr": 2.5e-3, "name": "f_dc"},
              {"params": [pc._features_rest], "lr": 2.5e-3 / 20.0, "name": "f_rest"}, {"params": [pc._opacity], "lr": 5e-2, "name": "opacity"},
              {"params": [pc._scaling], "lr": 5e-3, "name": "scaling"}, {"params": [pc._rotation], "lr": 1e-3, "name": "rotation"},
              {"params": [pc._normal], "lr": 1e-3, "name": "normal"}, {"params": [pc._offset], "lr": 1e-3, "name": "offset"}]
    opt = FusedAdam(groups, lr=0.0, eps=1e-15)
    params = [g["params"][0] for g in groups]
    red = vdist.ViewParallelReducer(params, sh=[pc._features_dc, pc._features_rest], means3D=pc._xyz) if world > 1 else None

    hist = []
    gc.collect(); gc.freeze()          # keep full collections (33-42 ms with torch imported) out of a 2 ms iteration
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(a.iters):
        vid = vdist.views_for_rank(it, rank, world, len(cams))
        cam = cams[vid]
        geo = it >= a.geo_from
        opt.zero_grad(set_to_none=True)

        def fwd_bwd():
            out = renderer.render(cam, pc, scene, pipe, args, bg, learnt_normal=False, nb_src_frames=3, buffer_length=4,
                                  render_geo=geo, return_depth_normal=False)
            loss = l1_loss(out["render"], targets[vid])
            if geo:   # multi-view term in the spirit of train.py:319-338: the first source's warped colours should match the image
                m = (out["cam_feat"][3:4] != 0).float()
                loss = loss + 0.05 * ((out["warped_image"][0:3] - targets[vid]).abs() * m).mean()
                scene.rendered_depth_list[vid] = out["median_intersected_depth"].detach()     # train.py:298-299
            loss.backward()
            return loss

        if red is not None:
            with red.capture():
                loss = fwd_bwd()
            red.reduce(average=True)
        else:
            loss = fwd_bwd()
        opt.step()
        hist.append(float(loss.detach()))
        if not a.quiet and rank == 0 and (it % 20 == 0 or it == a.iters - 1):
            print("iter %4d  view %d  geo %d  loss %.5f" % (it, vid, geo, hist[-1]), flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank == 0:
        k = max(1, min(10, a.iters // 4))
        print("train_synthetic: %d iterations, %.2f ms / iteration, loss %.5f -> %.5f" % (a.iters, dt / a.iters * 1e3, float(np.mean(hist[:k])), float(np.mean(hist[-k:]))))
    if world > 1:
        # replicas must stay bit-identical: same summed gradients on every rank, same optimiser step
        cs = torch.stack([p.detach().double().sum() for p in params])
        lo, hi = cs.clone(), cs.clone()
        torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN); torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
        if rank == 0:
            print("replicas in sync: %s" % bool(torch.equal(lo, hi)))
        torch.distributed.barrier(); torch.distributed.destroy_process_group()
    return hist


if __name__ == "__main__":
    main()
