#!/usr/bin/env python3
"""Headline benchmark of the plane rasterizer hot path (BASELINE.json metric):
train-step (rasterizer forward + backward) at 1920x1080 on 1M random-init Gaussians, SH degree 3
(BASELINE.json configs[2], "C3" of SURVEY.md 8(d)); whole-job views per second over N GPUs.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one view per rank: preprocess -> depth sort -> emit ->
tile sort -> render (forward), L1 loss against a fixed random target, render backward -> preprocess
backward, and for N > 1 the exchange (sum) of the Gaussian gradients over RCCL (view-parallel,
SURVEY.md 8(e)).  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

`--gpus N` with no WORLD_SIZE in the environment spawns the N ranks itself as child processes BEFORE
anything in this process touches the GPU (a process that has initialised HIP must never exec), binds
rank r to GPU r and fails (exit code 2) when the box has fewer than N devices.  IBGS_BENCH_SHARE_GPU=1
with IBGS_DIST_BACKEND=gloo lets the ranks share one device (tests of the N > 1 code path on 1-GPU boxes).
"""
import argparse
import gc
import json
import os
import socket
import statistics
import subprocess
import sys
import time


def cpu_budget():
    """Host CPUs this process may really use: the cgroup's CPU bandwidth quota (cpu.max = "<quota> <period>"), the affinity mask, the core count.
    The GPU boxes of this pool show 256 CPUs and grant 16 (cpu.max 1600000 100000)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            q, per = parse(open(path).read())
            if q != "max" and int(q) > 0:
                n = min(n, max(1, int(q) // int(per)))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


# Thread pools sized for every CPU the box SHOWS (OpenMP in torch and in the oracle, the BLAS under numpy) burn the cgroup's CPU quota in one burst -- their
# workers also spin for a while after each parallel region -- and the kernel then stops EVERY thread of the process until the next 100 ms period.  That was
# the "one step of 55-60 ms in about one run of three" of the `trained_geo` line (round 4: /sys/fs/cgroup/cpu.stat counted 13 throttled periods per bench
# run; the HIP API trace showed the main thread standing still between two kernel launches inside ibgs_forward, the autograd thread with it).  So: pools
# no larger than the quota, and no spinning.  Must happen before the libraries load.
CPU_BUDGET = max(1, cpu_budget() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))          # (ranks of one node share the quota)
for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_k, str(CPU_BUDGET))
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

import numpy as np
import torch
import torch.distributed as dist

torch.set_num_threads(CPU_BUDGET)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from ibgs_amd import _lib, synthetic as syn  # noqa: E402
from ibgs_amd._build import csrc_sha, tu_of, tu_shas  # noqa: E402
from ibgs_amd import dist as vdist  # noqa: E402
from ibgs_amd.losses import l1_loss  # noqa: E402
from ibgs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402

HBM_PEAK = 8.0e12   # bytes/s, MI355X_MICROARCH.md "HBM3E peak BW" (spec)
CLOCK_HZ = 2.4e9    # MI355X_MICROARCH.md "Max clock"
SIMDS = 1024        # 256 CUs x 4 SIMDs
VALU_CYCLES = 2.0   # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles (MI355X_MICROARCH.md "Wave scheduling")


def algorithmic_bytes(P, R, HW, Mc, tiles, geo=False, n_src=0):
    """SURVEY.md 8(d) byte model (implementation independent). Returns (B_fwd, B_bwd, B_render_fwd, B_render_bwd)."""
    bit = int(np.ceil(np.log2(max(tiles, 2))))
    n_pass = -(-(32 + bit) // 8)
    b_fwd = P * (44 + 12 * Mc + 79) + R * (12 + 24 * n_pass + 8 + 40) + HW * 20
    b_render_fwd = HW * 20 + R * 40
    b_render_bwd = HW * 20 + R * 40 + P * 112
    b_bwd = b_render_bwd + P * (44 + 12 * Mc + 56) + P * (12 * Mc + 12 + 12 + 16 + 24)
    if geo:
        b_fwd += P * 20 + R * 20 + HW * 228 + n_src * HW * 16
        b_render_fwd += R * 20 + HW * 228 + n_src * HW * 16
        b_bwd += HW * (16 + 60 + 4 + 52) + R * 20 + P * 40
        b_render_bwd += HW * (16 + 60 + 4 + 52) + R * 20 + P * 40
    return b_fwd, b_bwd, b_render_fwd, b_render_bwd


def implementation_bytes(P, kept, R, C, HW, Mc):
    """Bytes THIS design moves per step (docs/EXPERIMENTS.md "Bytes of the implementation"): SURVEY's model charges the reference's
    R-sized 64-bit sort (R x 24 B x 6 passes); here the sort is P-sized and the lists are placed directly.
    kept = Gaussians with tiles, C = coarse binning entries (one per Gaussian and 8 x 8-tile cell)."""
    pre = P * (44 + 12 * Mc) + P * (64 + 16 + 4 + 4 + 24 + 1 + 8)          # inputs; record, footprint, depth, tiles, cov3D, clamp bits, sort pair
    sort = P * 8 + kept * 16 * 4                                           # first pass reads every pair; four passes move the kept ones (key + id, in + out)
    binning = kept * 16 * 3 + C * 16 * 3 + R * 4                           # footprints (gather, re-store, read); coarse entries (write, two reads); list write
    fwd = R * (4 + 48) + HW * 20                                           # list + three record quads per entry; colour, final_T, n_contrib out
    bwd = R * (4 + 48) + HW * (8 + 12) + kept * 128                        # the same walk; per-pixel state and dL/dC in; one 64-B row read-modify-write per touched Gaussian
    pre_bwd = kept * (64 + 32 + 44 + 12 * Mc) + kept * (12 * Mc + 12 + 12 + 4 + 12 + 16) + P * 64          # rows, record, inputs; outputs; re-zeroed rows
    return pre + sort + binning + fwd + bwd + pre_bwd


def _profiles_for(workload_tag):
    """Committed rocprofv3 summaries (profiles/*_counters.json, written by profiles/summarize.py) of this workload, newest first."""
    import glob
    res = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_counters.json")), key=os.path.getmtime, reverse=True):
        try:
            c = json.load(open(path))
        except Exception:   # noqa: BLE001
            continue
        if c.get("workload") == workload_tag:
            res.append(c)
    return res


def profile_counters(kernel_substr, workload_tag):
    """Per-launch PMC counters of one kernel from the committed rocprofv3 summaries.  Returns (counters, source) -- (None, reason) when no
    summary of this workload was taken on the CURRENT sources of the translation unit that holds the kernel (its .hip + the headers;
    ibgs_amd/_build.py: tu_shas): stale numbers are dropped, not quoted next to fresh timings.  (Until round 4 the stamp was one hash over
    every source file, so a host-only edit of api.hip made every kernel's counters "stale".)"""
    tu = tu_of(kernel_substr)
    now = tu_shas().get(tu) if tu else None
    reason = "no profiles/*_counters.json for workload %r" % workload_tag
    for d in _profiles_for(workload_tag):
        stamp = (d.get("tu_shas") or {}).get(tu) if tu else None
        fresh = (stamp is not None and stamp == now) or d.get("csrc_sha") == csrc_sha()
        if not fresh:
            reason = "profile of %r: %s.hip / headers changed since it was taken (%s -> %s)" % (workload_tag, tu, stamp or d.get("csrc_sha"), now)
            continue
        for k, c in d.get("per_launch_counters", {}).items():
            if kernel_substr in k:
                return c, "profiles/%s @ %s %s (%s)" % (d.get("tag", "counters_latest") + "_counters.json", tu, stamp or d.get("csrc_sha"), d.get("date", "?"))
        reason = "kernel %s not in the profile summary" % kernel_substr
    return None, reason


def step_traffic_measured(workload_tag):
    """HBM bytes of one step as the PMC passes measured them (profiles/summarize.py: sum over the library's kernels), quoted only while every translation
    unit THAT HOLDS ONE OF THE SUMMED KERNELS is what it was when the passes ran (round 6: until then every unit of the library had to be -- an edit of
    depth_normal.hip, whose kernels no bench step launches, dropped the figure)."""
    now = tu_shas()
    for d in _profiles_for(workload_tag):
        if "step_traffic_measured" not in d:
            continue
        stamp = d.get("tu_shas") or {}
        units = {tu_of(k) for k in (d["step_traffic_measured"].get("by_kernel") or {})}
        fresh = bool(units) and None not in units and all(stamp.get(u) is not None and stamp.get(u) == now.get(u) for u in units)
        if fresh or stamp == now or d.get("csrc_sha") == csrc_sha():
            return {"bytes_per_step": d["step_traffic_measured"]["bytes_per_step"], "source": "profiles/%s_counters.json (%s)" % (d.get("tag"), d.get("date", "?")),
                    "how": d["step_traffic_measured"].get("how")}
    return None


def workload_tag(cfg, geo, forward_only, opacity, cluster=0.0, anisotropy=None, scale_sigma=0.0):
    """Names a workload in profiles/*_counters.json (profiles/summarize.py stamps the same string)."""
    return "%s%s%s opacity=%s%s%s%s" % (cfg, " geo" if geo else "", " forward-only" if forward_only else "", opacity,
                                        (" cluster=%g" % cluster) if cluster > 0 else "", (" anisotropy=%s" % anisotropy) if anisotropy else "",
                                        (" scale_sigma=%g" % scale_sigma) if scale_sigma > 0 else "")


class Workload:
    """One view of one BASELINE config resident on `dev`: leaves, settings, the step closure."""
    CLUSTER = 0.0          # defaults of the scene shape (--cluster / --anisotropy / --scale-sigma); a Workload may be given its own
    ANISOTROPY = None
    SCALE_SIGMA = 0.0
    torch_l1 = True          # the loss of every timed step is the reference's own expression (utils/loss_utils.py:23-24); the one-pass fused L1 is an extra key

    def __init__(self, cfg, view, dev, opacity, geo, forward_only, target_seed, cluster=None, anisotropy="default", scale_sigma=None):
        c = syn.CONFIGS[cfg]
        self.cluster = Workload.CLUSTER if cluster is None else cluster
        self.anisotropy = Workload.ANISOTROPY if anisotropy == "default" else anisotropy
        self.scale_sigma = Workload.SCALE_SIGMA if scale_sigma is None else scale_sigma
        inp = syn.make_scene(c["P"], c["W"], c["H"], sh_degree=c["sh_degree"], seed=c["seed"], view=view, opacity=opacity,
                             anisotropy=self.anisotropy, scale_sigma=self.scale_sigma, cluster=self.cluster)
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
        P, H, W = c["P"], c["H"], c["W"]
        self.c, self.cfg, self.inp, self.geo, self.forward_only, self.dev = c, cfg, inp, geo, forward_only, dev
        self.P, self.H, self.W = P, H, W
        lv = {
            "means3D": t(inp["means3D"]).requires_grad_(True), "shs": t(inp["shs"]).requires_grad_(True),
            "opacities": t(inp["opacities"]).reshape(P, 1).requires_grad_(True),
            "scales": t(inp["scales"]).requires_grad_(True), "rotations": t(inp["rotations"]).requires_grad_(True),
            "means2D": torch.zeros(P, 3, device=dev, requires_grad=True),
            "means2D_abs": torch.zeros(P, 3, device=dev, requires_grad=True),
        }
        z = lambda *s: torch.zeros(*s, device=dev)
        st = GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=float(inp["tanfovx"]), tanfovy=float(inp["tanfovy"]), bg=t(inp["bg"]),
            scale_modifier=1.0, viewmatrix=t(inp["viewmatrix"]), projmatrix=t(inp["projmatrix"]),
            ref_to_src_list=z(1, 16), src_cam_pos=z(1, 3), src_images=z(1, 3, 1), src_rendered_depths=z(1, 1, 1),
            nb_src_images=1, buffer_length=4, depth_error_threshold=0.01, sh_degree=c["sh_degree"], campos=t(inp["campos"]),
            prefiltered=False, render_geo=False, render_depth_only=False, debug=False)
        if geo:
            # sources = the 4 nearest other orbit views; their images / depth maps are this op's own renders (untimed)
            ref_cam = inp["_cam"]
            src_ids = [(view + d) % 8 for d in (1, 7, 2, 6)]
            src_cams = [syn.make_camera(W, H, azimuth_deg=45.0 * k) for k in src_ids]
            imgs, deps = [], []
            with torch.no_grad():
                for sc in src_cams:
                    am = t(syn.plane_all_map(inp["means3D"], inp["scales"], inp["rotations"], sc))
                    for depth_only in (False, True):
                        sst = st._replace(viewmatrix=t(sc["viewmatrix"]), projmatrix=t(sc["projmatrix"]), campos=t(sc["campos"]),
                                          render_depth_only=depth_only)
                        o = GaussianRasterizer(sst)(means3D=lv["means3D"], means2D=lv["means2D"], means2D_abs=lv["means2D_abs"],
                                                    opacities=lv["opacities"], shs=lv["shs"], scales=lv["scales"],
                                                    rotations=lv["rotations"], all_map=am)
                        (deps if depth_only else imgs).append(o[3].clone() if depth_only else o[0].clone())
            r2s, scp = syn.ref_to_src(ref_cam, src_cams)
            lv["all_map"] = t(syn.plane_all_map(inp["means3D"], inp["scales"], inp["rotations"], ref_cam)).requires_grad_(True)
            st = st._replace(ref_to_src_list=t(r2s), src_cam_pos=t(scp), src_images=torch.stack(imgs), src_rendered_depths=torch.stack(deps),
                             nb_src_images=4, buffer_length=4, depth_error_threshold=0.01, render_geo=True)
        self.leaves, self.st = lv, st
        self.rast = GaussianRasterizer(st)
        self.target = torch.rand(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(target_seed))
        self.params = [lv[k] for k in ("means3D", "shs", "opacities", "scales", "rotations")]
        # geo line: L1 terms on the normal map (target 0), the median depth (0) and the warped source colours (0.5)
        self.geo_targets = (torch.zeros(3, H, W, device=dev), torch.zeros(1, H, W, device=dev), torch.full((15, H, W), 0.5, device=dev)) if geo else None
        self.R = 0

    def hop_to(self, view):
        """Point the (colour) workload at another orbit camera: same Gaussians, same target, new matrices."""
        cam = syn.make_camera(self.W, self.H, azimuth_deg=45.0 * view)
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=self.dev)
        if not hasattr(self, "_hop"):
            self._hop = {}
        if view not in self._hop:
            self._hop[view] = GaussianRasterizer(self.st._replace(viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), campos=t(cam["campos"])))
        self.rast = self._hop[view]

    def _call(self):
        lv = self.leaves
        return self.rast(means3D=lv["means3D"], means2D=lv["means2D"], means2D_abs=lv["means2D_abs"], opacities=lv["opacities"],
                         shs=lv["shs"], scales=lv["scales"], rotations=lv["rotations"], all_map=lv.get("all_map"))

    def fwd_only(self):
        from ibgs_amd import rasterizer as _r
        with torch.no_grad():
            self._call()
        self.R = _r.LAST_NUM_RENDERED

    def local_step(self, backward=None):
        """forward + loss + backward of this rank's view; `backward(loss)` lets the caller wrap autograd (gradient capture)."""
        if self.forward_only:
            return self.fwd_only()
        for v in self.leaves.values():
            v.grad = None
        outs = self._call()
        # utils/loss_utils.py:23-24 as the reference writes it, or the same value and gradient in one pass each (ibgs_amd.losses; SURVEY section 2: out of scope)
        l1 = (lambda x, y: torch.abs(x - y).mean()) if self.torch_l1 else l1_loss
        loss = l1(outs[0], self.target)
        if self.geo:   # every differentiable geo output takes part: normal map, median depth, warped source colours
            loss = loss + l1(outs[2], self.geo_targets[0]) + l1(outs[3], self.geo_targets[1]) + l1(outs[5], self.geo_targets[2])
        self.R = outs[0].grad_fn.num_rendered
        if backward is None:
            loss.backward()
        else:
            backward(loss)

    def kernel_names(self):
        tiles = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        big = tiles >= 4096
        if self.geo:
            return ("render_fwd_kernel<1, 2, 4>" if big else "render_fwd_kernel<1, 1, 4>"), ("render_bwd_geo4_kernel" if big else "render_bwd_geo_kernel")
        return ("render_fwd_kernel<0, 4, 4>" if big else "render_fwd_kernel<0, 1, 4>"), ("render_bwd_color_kernel" if big else "render_bwd_color_small_kernel")

    def describe(self, opacity, world, exchange):
        c = self.c
        shape = ((", %g of the Gaussians in one blob" % self.cluster) if self.cluster > 0 else "") \
            + ((", anisotropy=%s" % self.anisotropy) if self.anisotropy else "") \
            + ((", log-normal sizes sigma=%g" % self.scale_sigma) if self.scale_sigma > 0 else "")
        return "%s: %d random-init Gaussians, %dx%d, SH degree %d, rasterizer %s, opacity=%s%s, one view per GPU%s" % (
            self.cfg, self.P, self.W, self.H, c["sh_degree"], "forward only" if self.forward_only else "fwd+bwd, L1 loss vs fixed random target (torch: abs().mean(), the reference's utils/loss_utils.py:23-24)",
            opacity + shape,
            ", render_geo n_src=4 L=4" if self.geo else "", (", RCCL gradient exchange (%s)" % exchange) if world > 1 else "")


def fence(world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def measure(wl, step, steps, warmup, world, n_stage_steps=3, n_fwd=10):
    """W untimed warm-ups, then EXACTLY K steps between barrier + synchronize on both sides (wall clock, max over ranks);
    every timed step is also bracketed by hipEvents on the op's stream (= torch's current stream) for the median."""
    kstage = "render_fwd" if wl.forward_only else "render_bwd"
    # The interpreter's cyclic garbage collector is kept out of the timed steps: with torch imported a full collection walks ~170 k
    # objects = 33-42 ms, i.e. up to twenty C3 steps, whenever its allocation counters happen to trip (measured: 2 of 5 runs of the
    # 20-step geo line had one such step).  Everything alive now moves to the permanent generation (gc.freeze) and later collections
    # only see what the steps themselves allocate.  A training loop should do the same once after set-up.
    # The collection runs BEFORE the warm-up steps: it idles the GPU for those 40 ms, and the first ten steps after an idle gap run up
    # to 30 % slower while the clocks come back (per_step_ms_hipevent showed 2.31, 1.97, 1.94, 1.91 ... 1.79 at the start of the timed
    # region when the collection sat between warm-up and timing) -- that is what warm-up steps are for.  (Round 4: 40 ms of dense matrix products or
    # of 256 MB copies in front of the warm-up steps do NOT shorten that ramp -- it follows the workload's own kernels; with 25 warm-up steps the
    # timed steps start at the steady value, with the driver's 5 the first seven or so are 1-5 % slow.  Left as it is: W is the driver's.)
    gc.collect()
    gc.freeze()
    for _ in range(warmup):
        step()
    gc.freeze()          # what the warm-up steps left behind (caches, scratch): no collection, just out of the collector's sight
    _lib.timing_enable([kstage])     # hipEvents around the dominant kernel only, on the op's stream
    _lib.timing_collect()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    # ... and the AUTOMATIC collector is off for the K timed steps (round 4): a generation-2 collection stops the host for 5-11 ms (tools/spike_hunt.py), and
    # whether one falls into 20 timed steps is the interpreter's business, not the step's.  A training loop calls gc.disable() and collects by hand every
    # N iterations.
    # (The 55-60 ms step that one full run of three showed in the `trained_geo` line was neither: the cgroup's CPU bandwidth quota, exhausted by 128-thread
    # pools -- see CPU_BUDGET at the top of this file.  The line still carries median, maximum and every step's time; `ms_per_step` is the wall clock as always.)
    gc_was_on = gc.isenabled()
    gc.disable()
    fence(world)
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    fence(world)
    dt = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    tm = _lib.timing_collect()
    _lib.timing_enable([])
    per_step = [a.elapsed_time(b) for a, b in ev]
    if world > 1:
        tt = torch.tensor([dt], device=wl.dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # forward-only time of the same workload ("report fwd-only and fwd+bwd separately"), untimed for `value`
    fence(world)
    tf = time.perf_counter()
    for _ in range(n_fwd):
        wl.fwd_only()
    fence(world)
    fwd_ms = (time.perf_counter() - tf) / n_fwd * 1e3
    # per-stage breakdown from a few extra (untimed) steps
    _lib.timing_enable(_lib.STAGES)
    for _ in range(n_stage_steps):
        step()
    torch.cuda.synchronize()
    stages = {k: v[0] / float(n_stage_steps) for k, v in _lib.timing_collect().items()}
    _lib.timing_enable([])
    return {"dt": dt, "ms_step": dt / steps * 1e3, "median_ms": statistics.median(per_step), "min_ms": min(per_step), "max_ms": max(per_step),
            "per_step": [round(x, 3) for x in per_step],
            "kernel_ms": tm[kstage][0] / max(tm[kstage][1], 1), "fwd_ms": fwd_ms, "stages": stages}


def walked_entries(wl):
    """List entries the blend kernels really visit in one pass of `wl`: per tile, how far the forward walked its list (the largest contributor index of any
    pixel; one word per wave of the tile in the image arena -- ImgState::tile_walked, the words the backward orders its launch by)."""
    for v in wl.leaves.values():
        v.grad = None
    out = wl._call()[0]
    img = out.grad_fn.saved_tensors[-1]
    lib = _lib.load()
    tiles = ((wl.W + 15) // 16) * ((wl.H + 15) // 16)
    off, moff = lib.ibgs_img_offset(wl.W, wl.H, b"tile_walked"), lib.ibgs_img_offset(wl.W, wl.H, b"meta")
    ipt = int(img[moff:moff + 128].view(torch.int32)[10].item())          # waves per tile of the forward variant that ran: the words lie at [tile * ipt + wave]
    tw = img[off:off + tiles * ipt * 4].view(torch.int32).view(tiles, ipt)
    return int(tw.max(dim=1).values.to(torch.int64).sum().item())


def roofline(wl, m, workload_tag, walked=None):
    """Roofline object of the dominant kernel.  The blend kernels are VALU-issue bound: `bound` says so and
    `valu` holds the issue-rate fraction; achieved / peak / frac stay the HBM figures of SURVEY 8(d) (algorithmic bytes of
    that launch / its live hipEvent duration / 8 TB/s)."""
    P, H, W = wl.P, wl.H, wl.W
    R = int(wl.R); HW = H * W; Mc = int(wl.inp["shs"].shape[1]); tiles = ((W + 15) // 16) * ((H + 15) // 16)
    b_fwd, b_bwd, b_rfwd, b_rbwd = algorithmic_bytes(P, R, HW, Mc, tiles, wl.geo, 4 if wl.geo else 0)
    kf, kb = wl.kernel_names()
    kernel, kbytes = (kf, b_rfwd) if wl.forward_only else (kb, b_rbwd)
    k_ms = m["kernel_ms"]
    achieved = kbytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    ctr, src = profile_counters(kernel.split("<")[0] if not wl.forward_only else kernel, workload_tag)
    traffic, valu, mfma = None, None, None
    if ctr is not None:
        if "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
            # MI355X_MICROARCH.md "HBM": counters in KiB; FETCH_SIZE reports half the bytes of 16-B-per-lane reads on gfx950
            traffic = 2.0 * ctr["FETCH_SIZE"] * 1024.0 + ctr["WRITE_SIZE"] * 1024.0
        if "SQ_INSTS_VALU" in ctr and k_ms > 0:
            cyc = k_ms * 1e-3 * CLOCK_HZ
            valu = {"wave_valu_insts_per_launch": ctr["SQ_INSTS_VALU"], "frac": ctr["SQ_INSTS_VALU"] * VALU_CYCLES / (SIMDS * cyc),
                    "cycles_per_inst": SIMDS * cyc / ctr["SQ_INSTS_VALU"], "peak": "1 wave64 VALU instruction per 2 cycles per SIMD, 1024 SIMDs at 2.4 GHz",
                    "source": src}
        if "SQ_INSTS_VALU_MFMA_MOPS_F32" in ctr or "SQ_VALU_MFMA_BUSY_CYCLES" in ctr:
            mfma = {"busy_cycles": ctr.get("SQ_VALU_MFMA_BUSY_CYCLES"), "mops_f32": ctr.get("SQ_INSTS_VALU_MFMA_MOPS_F32"), "source": src}
    step_bytes = b_fwd if wl.forward_only else b_fwd + b_bwd
    rf = {"bound": "valu", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
          "frac": achieved * 1e9 / HBM_PEAK, "traffic": traffic, "traffic_source": src if traffic is not None else None,
          "algorithmic_bytes_per_launch": kbytes, "kernel_ms": k_ms,
          "valu": valu, "valu_note": None if valu is not None else src,
          "mfma_utilisation": 0.0 if mfma is None else mfma,   # no kernel of this path is a dense contraction (DESIGN.md "MFMA", measured probe)
          "step_algorithmic_bytes": step_bytes, "step_frac": step_bytes / (m["ms_step"] * 1e-3) / HBM_PEAK}
    # ... beside the fraction on the bytes the PMC passes saw the whole step move (SURVEY's model credits an R-sized 64-bit sort this design does not run)
    stm = None if wl.forward_only else step_traffic_measured(workload_tag)
    rf["step_frac_measured"] = None if stm is None else stm["bytes_per_step"] / (m["ms_step"] * 1e-3) / HBM_PEAK
    rf["step_traffic_measured"] = stm
    if walked is not None and k_ms > 0:
        # SURVEY's R x 40 (+ 20 geo) B charges every list entry; the blend walks a list only until its pixels are opaque (trained scenes: ~10 % of the
        # entries).  `walked` = the entries the kernel really visits (sum over tiles of how far the forward walked the list: ImgState::tile_walked).  The
        # fraction on max(bytes of the walked entries, bytes the counters saw) is `frac_walked`.
        per_entry = 40 + (20 if wl.geo else 0)
        wbytes = kbytes - (R - walked) * per_entry
        best = max(wbytes, traffic or 0.0)
        # (`achieved` / `frac` stay SURVEY 8(d)'s definition in EVERY roofline object -- the one BENCH_r01..r04 carried -- so that rounds compare like with like;
        # the figure on the bytes really moved has keys of its own.  Round 5 had overwritten `frac` with it for the geo objects.)
        rf.update({"walked_entries": int(walked), "walked_fraction_of_R": walked / max(R, 1), "walked_bytes_per_launch": wbytes,
                   "achieved_walked": best / (k_ms * 1e-3) / 1e9, "frac_walked": best / (k_ms * 1e-3) / HBM_PEAK,
                   "frac_walked_basis": "measured traffic" if (traffic or 0.0) >= wbytes else "walked entries"})
    return rf


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


def cpu_baseline_c1(dev):
    """config C1 (10 k Gaussians, 400 x 400, SH 3) on the host cores with 8 threads (what the reference pins:
    train.py:453, render.py:396) and with all cores, median of 5 after one warm-up, forward and forward + backward; the HIP path at the
    same inputs beside it.  The CPU path is the oracle (oracle/ibgs_oracle.c, OpenMP over pixels / Gaussians) -- the reference itself has
    no CPU rasterizer."""
    import oracle
    lib = oracle.lib()
    c = syn.CONFIGS["C1"]
    inp = syn.make_scene(c["P"], c["W"], c["H"], sh_degree=c["sh_degree"], seed=c["seed"])
    g = np.random.default_rng(1).standard_normal((3, c["H"], c["W"])).astype(np.float32)
    all_cores = int(lib.orc_num_threads())
    res = {}
    for n in (8, all_cores):
        lib.orc_set_num_threads(int(n))
        tf, tb = [], []
        for it in range(6):
            t0 = time.perf_counter(); f = oracle.forward(inp); t1 = time.perf_counter(); oracle.backward(inp, f, g); t2 = time.perf_counter()
            if it:
                tf.append((t1 - t0) * 1e3); tb.append((t2 - t0) * 1e3)
        res["threads_%d" % n] = {"fwd_ms": statistics.median(tf), "fwd_bwd_ms": statistics.median(tb)}
    lib.orc_set_num_threads(all_cores)
    wl = Workload("C1", 0, dev, "init", False, False, 7)
    gt = torch.as_tensor(g, device=dev)

    def hip_step(backward):
        for v in wl.leaves.values():
            v.grad = None
        o = wl._call()[0]
        if backward:
            (o * gt).sum().backward()
    out = {}
    for name, bw in (("fwd_ms", False), ("fwd_bwd_ms", True)):
        for _ in range(5):
            hip_step(bw)
        ts = []
        for _ in range(20):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); hip_step(bw); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        out[name] = statistics.median(ts)
    res["hip"] = out
    res["speedup_fwd_bwd_vs_8_threads"] = res["threads_8"]["fwd_bwd_ms"] / out["fwd_bwd_ms"]
    res["config"] = "C1: 10000 random-init Gaussians, 400x400, SH degree 3, render_geo=False, seed 1; median of 5 (CPU) / 20 (HIP)"
    return res


def test_frame(dev, c):
    """The reference's test-time frame (render.py:126-156, gaussian_renderer/__init__.py:228-267 with do_render_src_depth): the depth maps of
    the 4 source views are rendered first (here ONE batched depth-only pass), then the main geo pass warps into them.  No gradients.
    C3-sized scene through ibgs_amd.renderer.render() -- the Python surface the reference's render.py calls; the colour network that
    follows in the reference is outside the rasterizer path and not part of the number."""
    from ibgs_amd import renderer, simple_scene
    P, W, H = c["P"], c["W"], c["H"]
    g = syn.make_gaussians(P, c["seed"], sh_degree=3, max_coeffs=16, opacity="init")
    rng = np.random.default_rng(0)
    g["normal"] = rng.normal(size=(P, 3)).astype(np.float32); g["offset"] = (0.01 * rng.normal(size=(P, 1))).astype(np.float32)
    pc = simple_scene.SimpleGaussians(g, sh_degree=3, device=dev)
    cams = simple_scene.orbit_cameras(W, H, n_views=8, device=dev, nearest=4)
    scene = simple_scene.SimpleScene(cams, images=torch.rand(8, 3, H, W, device=dev), device=dev)
    pipe, args = simple_scene.default_pipe(), simple_scene.default_args()
    bg = torch.zeros(3, device=dev)
    with torch.no_grad():
        fn = lambda: renderer.render(cams[0], pc, scene, pipe, args, bg, True, 4, 4, render_geo=True, do_render_src_depth=True, return_depth_normal=False)
        for _ in range(3):
            fn()
        fence(1); t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            fn()
        fence(1)
        ms = (time.perf_counter() - t0) / n * 1e3
    return {"ms_per_frame": ms, "fps": 1000.0 / ms,
            "workload": "%d Gaussians, %dx%d: 4 source depth maps in one batched depth-only pass + the main render_geo pass (n_src 4, L 4), learnt normals, no gradients" % (P, W, H)}


def gpu_kernel_sum_ms(fn, n):
    """Sum of the durations of EVERY kernel and copy the GPU ran during `n` calls of `fn` (the library's and torch's), per call: torch.profiler's device
    events (roctracer).  What a step would take if the host never made the GPU wait.  None when the profiler is unavailable."""
    try:
        from torch.profiler import ProfilerActivity, profile
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
        tot = 0.0
        for e in prof.events():
            if e.device_type == torch.autograd.DeviceType.CUDA:
                tot += float(getattr(e, "device_time_total", 0.0) or getattr(e, "cuda_time_total", 0.0))
        return tot / n * 1e-3 if tot > 0 else None
    except Exception as ex:   # noqa: BLE001
        sys.stderr.write("bench.py: torch.profiler unavailable (%s): no kernel sums\n" % ex)
        return None


def timed_wall_ms(fn, n, warmup=5):
    for _ in range(warmup):
        fn()
    fence(1); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    fence(1)
    return (time.perf_counter() - t0) / n * 1e3


def small_frame(dev, steps):
    """A frame below the size at which the step is the GPU's alone: 1 M trained-opacity Gaussians at 1280 x 720 (3 600 tiles, the hybrid
    kernels' range), colour pass forward + L1 + backward: wall clock per step against the sum of its kernels -- the difference is what host / launch
    overhead and the wait for R still expose.  Extra key; never `value`."""
    wl = Workload("C3_720p", 0, dev, "trained", False, False, 1234)
    wall = timed_wall_ms(wl.local_step, steps, warmup=10)
    ksum = gpu_kernel_sum_ms(wl.local_step, 5)
    _lib.timing_enable(_lib.STAGES)
    for _ in range(3):
        wl.local_step()
    torch.cuda.synchronize()
    stages = {k: v[0] / 3.0 for k, v in _lib.timing_collect().items()}
    _lib.timing_enable([])
    c = wl.c
    return {"workload": "%d trained-opacity Gaussians, %dx%d, SH degree %d, colour pass fwd + L1 (torch) + bwd" % (c["P"], c["W"], c["H"], c["sh_degree"]),
            "steps": steps, "ms_per_step": wall, "kernel_sum_ms": ksum, "host_exposed_ms": None if ksum is None else max(0.0, wall - ksum),
            "num_rendered": int(wl.R), "stages_ms": stages}


class TrainIteration:
    """ONE iteration as the reference's trainer runs it in steady state (train.py:287-430), at C3 size on the `trained_geo` scene: `renderer.render(
    render_geo=True, return_depth_normal=True)` through the fused plane glue with 4 cached source depths, L1 against the view's image, backward, the depth
    cache write (:298-299), the densification statistics (:400-405; `densify.add_densification_stats`: one launch, no host syncs), `optimizer.step()` +
    `zero_grad` (:421-424), cameras round-robin.
    full = False: the loss is L1 on `render` alone -- the 2 x len(cameras) iterations before single_view_weight_from_iter in which train.py renders geo but has none of
    its geo losses yet (:289-316); the backward then receives no gradient for the normal map, the median depth and the warped images (and the library skips the
    window pass).  full = True: train.py's STEADY STATE (iteration > 7000, ~77 % of a 30 k run): + the single-view normal consistency (:309-316: 0.4 |dn - n| + 0.6 (1 -
    dn . n), weight 0.03 -- the depth -> normal backward runs) + the multi-view photometric term (:319-338) on the first three sources, restated in torch.  Left out,
    and said so in the workload string: the two SSIM terms (utils/loss_utils.py, out of scope; the photometric term therefore runs with its L1 part at photo_weight
    0.3 -- the reference's default photo_ssim_weight = 1.0 would leave only its SSIM part), the colour-aggregation network, and the host-synchronising
    `if torch.sum(valid_mask) > 0` (a clamp instead)."""
    ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation",
            "normal": "_normal", "offset": "_offset"}
    LRS = {"xyz": 1.6e-4, "f_dc": 2.5e-3, "f_rest": 2.5e-3 / 20.0, "opacity": 2.5e-2, "scaling": 5e-3, "rotation": 1e-3, "normal": 1e-3, "offset": 1.6e-5}   # arguments/__init__.py:90-98

    def __init__(self, dev, c, opt_cls, full=False, sh_factored=False):
        from ibgs_amd import densify, renderer, simple_scene
        self.renderer = renderer
        self.full = bool(full)
        # sh_factored (FusedAdam only, round 6): the backward leaves dL/dsh unwritten and optimizer.step() updates the SH coefficients straight from its factors
        # (ibgs_amd/optim.py: 192 B per Gaussian neither written nor read back) -- a two-line change of the trainer, INTEGRATION.md
        self.sh_factored = bool(sh_factored)
        P, W, H = c["P"], c["W"], c["H"]
        self.P, self.dev = P, dev
        g = syn.make_gaussians(P, c["seed"], sh_degree=3, max_coeffs=16, opacity="trained", anisotropy="plane", scale_sigma=1.0, cluster=0.3)
        rng = np.random.default_rng(0)
        g["normal"] = rng.normal(size=(P, 3)).astype(np.float32); g["offset"] = (0.01 * rng.normal(size=(P, 1))).astype(np.float32)
        self.cams = simple_scene.orbit_cameras(W, H, n_views=8, device=dev, nearest=4)
        self.scene = simple_scene.SimpleScene(self.cams, images=torch.rand(8, 3, H, W, device=dev), device=dev)
        self.pipe, self.args = simple_scene.default_pipe(), simple_scene.default_args()
        self.bg = torch.zeros(3, device=dev)
        self.pc = simple_scene.SimpleGaussians(g, sh_degree=3, device=dev)
        self.opt = opt_cls([{"params": [getattr(self.pc, self.ATTR[k])], "lr": lr, "name": k} for k, lr in self.LRS.items()], lr=0.0, eps=1e-15)   # gaussian_model.py:227-241
        with torch.no_grad():
            self.scene.rendered_depth_list = renderer.render_depth_batch(self.cams, self.pc, self.scene, self.pipe, self.args, self.bg, True, 4, 4)
        self.st = {k: torch.zeros((P, 1) if i < 4 else (P,), device=dev) for i, k in enumerate(densify.STAT_NAMES)}          # gaussian_model.py:218-221, 203-204
        self.densify = densify
        self.it = 0

    def __call__(self):
        k = self.it % 8
        self.it += 1
        st, scene = self.st, self.scene
        self.renderer.LEAF_SINKS = self.sh_factored          # (the tuned twins also take the opt-in leaf sinks: no retain_grad() clones of the two (P, 3) sink gradients)
        out = self.renderer.render(self.cams[k], self.pc, scene, self.pipe, self.args, self.bg, True, 4, 4, render_geo=True, return_depth_normal=True)
        self.renderer.LEAF_SINKS = False
        gt = scene.original_image_list[k]
        loss = torch.abs(out["render"] - gt).mean()
        if self.full:
            H, W = gt.shape[-2], gt.shape[-1]
            loss = (1.0 - 0.2) * loss          # lambda_dssim = 0.2 (arguments/__init__.py:100); the SSIM term itself is out of scope
            normal, dn = out["rendered_normal"], out["median_intersected_depth_normal"]          # train.py:309-316, single_view_weight 0.03
            loss = loss + 0.03 * (0.4 * (dn - normal).abs().sum(0).mean() + 0.6 * (1 - (dn * normal).sum(0)).mean())
            warped = out["warped_image"].view(-1, 3, H, W)[:3]          # train.py:319-338, nb_visible_src_frames 3, photo_weight 0.3
            valid = (torch.sum(out["cam_feat"].view(-1, 4, H, W)[:3], dim=1, keepdim=True) > 0).float()
            ref = gt.unsqueeze(0)
            masked = valid * warped + (1 - valid) * ref
            l1p = torch.abs(ref - masked).mean(1)
            loss = loss + 0.3 * (torch.sum(l1p * valid[:, 0]) / torch.sum(valid[:, 0]).clamp(min=1.0))
        if self.sh_factored:
            from ibgs_amd import rasterizer as _rz
            with _rz.capture_sh_factors() as sh_items:
                loss.backward()
        else:
            loss.backward()
        with torch.no_grad():
            scene.rendered_depth_list[k] = out["median_intersected_depth"].detach()
            # max_radii2D + add_densification_stats (train.py:400-405) in one launch (`densify.add_densification_stats`; the reference's five boolean-indexed
            # updates each stop the host for an index count)
            self.densify.add_densification_stats(st, out["viewspace_points"], out["viewspace_points_abs"], out["radii"])
        if self.sh_factored:
            self.opt.step(sh_factors=sh_items, sh_params=(self.pc._features_dc, self.pc._features_rest), means3D=self.pc._xyz)
        else:
            self.opt.step()
        self.opt.zero_grad(set_to_none=True)

    def densify_pass(self, keep):
        """The every-100th pass (:407-410): prune the rows outside `keep`, append as many (P stays what it was): all parameters, both Adam moments, the statistics."""
        from ibgs_amd import densify
        st = self.st
        n_app = self.P - int(keep.sum().item())
        ext = {gr["name"]: gr["params"][0].detach()[:n_app].clone() for gr in self.opt.param_groups}
        new, extra = densify.prune_and_extend_optimizer(self.opt, keep, ext, extra=[st[k] for k in densify.STAT_NAMES])
        for kname, a_ in self.ATTR.items():
            setattr(self.pc, a_, new[kname])
        self.st = dict(zip(densify.STAT_NAMES, extra))


def train_iter(dev, c, steps):
    """`TrainIteration` timed: wall clock per iteration against the sum of its kernels (the library's and torch's), with FusedAdam and with the reference's
    torch.optim.Adam; and one `compact_append` pass (1 % of the points pruned, as many appended) as the trainer runs it every 100th iteration.  Never `value`."""
    from ibgs_amd.optim import FusedAdam
    res = {}
    for name, opt_cls, full, shf in (("fused_adam", FusedAdam, False, False), ("torch_adam", torch.optim.Adam, False, False), ("full_fused_adam", FusedAdam, True, False),
                                     ("fused_adam_sh_factored", FusedAdam, False, True), ("full_fused_adam_sh_factored", FusedAdam, True, True)):
        ti = TrainIteration(dev, c, opt_cls, full=full, sh_factored=shf)
        wall = timed_wall_ms(ti, steps, warmup=10)
        ksum = gpu_kernel_sum_ms(ti, 4)
        res[name] = {"ms_per_iter": wall, "kernel_sum_ms": ksum, "host_exposed_ms": None if ksum is None else max(0.0, wall - ksum)}
        if full:
            _lib.timing_enable(_lib.STAGES)
            for _ in range(3):
                ti()
            torch.cuda.synchronize()
            res[name]["library_stages_ms"] = {k: v[0] / 3.0 for k, v in _lib.timing_collect().items() if v[0] > 0}
            _lib.timing_enable([])
        if opt_cls is FusedAdam and not full:
            keep = torch.rand(ti.P, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) > 0.01
            ti.densify_pass(keep); fence(1)
            t0 = time.perf_counter(); ti.densify_pass(keep); fence(1)
            res["densify_pass_ms"] = (time.perf_counter() - t0) * 1e3
            ti(); fence(1)          # the iteration still runs on the new Parameters
        del ti
        torch.cuda.empty_cache()
    res["ms_per_iter_incl_densify_every_100"] = res["fused_adam"]["ms_per_iter"] + res["densify_pass_ms"] / 100.0
    res["workload"] = ("%d trained plane-like Gaussians (30 %% in one blob, log-normal sizes), %dx%d, SH 3: renderer.render(render_geo, fused plane map, 4 cached sources, L 4, "
                       "depth normal) + L1 + backward + depth cache + densification statistics + optimizer.step(), 8 cameras round-robin" % (c["P"], c["W"], c["H"]))
    res["which_is_train_py"] = ("fused_adam / torch_adam: the loss is L1 on `render` alone = train.py's 2 x len(cameras) warm-in iterations of the geo pass (no upstream gradient for normal map, "
                                "median depth, warped images; the window pass of the backward is skipped).  full_fused_adam: train.py's steady state (iteration > 7000): + normal consistency "
                                "(train.py:309-316) + multi-view photometric L1 on 3 sources (train.py:319-338), all four upstream gradients reach the rasterizer backward and the depth -> normal "
                                "backward runs; NOT included: the two SSIM terms (out of scope) and the colour-aggregation network.  *_sh_factored: the same two iterations with the trainer's "
                                "backward inside `rasterizer.capture_sh_factors()` and `optimizer.step(sh_factors=...)`: the dense dL/dsh (192 B per Gaussian) is neither written by the "
                                "backward nor read by the optimiser; parameters bit-identical to the expanded gradient's (tests/test_gpu_adam.py); and `renderer.LEAF_SINKS = True` (the two gradient sinks "
                                "as leaf aliases: no retain_grad() clones)")
    res["steps"] = steps
    return res


def self_launch(a, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks as CHILD processes (this process has not touched the
    GPU and never will), rank r on GPU r, rendezvous on 127.0.0.1.  Exit code = first failing rank's, 2 if devices are missing."""
    n = a.gpus
    ndev = torch.cuda.device_count()          # counting devices does not initialise HIP on this image
    share = os.environ.get("IBGS_BENCH_SHARE_GPU") == "1"
    if ndev < n and not share:
        sys.stderr.write("bench.py: --gpus %d but this box has %d GPU(s); refusing to benchmark fewer ranks than asked for\n" % (n, ndev))
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):          # the ranks share this process's CPU quota (CPU_BUDGET above)
            env[k] = str(max(1, CPU_BUDGET // n))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = time.time() + float(os.environ.get("IBGS_BENCH_TIMEOUT", "1500"))
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in alive:       # one rank failed: the others would wait in a collective forever
                    q.terminate()
        if alive and time.time() > deadline:
            for q in alive:
                q.kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C3", choices=list(syn.CONFIGS))
    ap.add_argument("--opacity", default="init", choices=["init", "trained"])
    ap.add_argument("--cluster", type=float, default=0.0,
                    help="NOT the BASELINE workload: move this fraction of the Gaussians into one blob (a non-uniform image: what the tile -> XCD "
                         "mapping and the binning must cope with in real scenes)")
    ap.add_argument("--anisotropy", default=None, choices=["plane", "needle", "mixed"],
                    help="NOT the BASELINE workload: shape of the Gaussians (synthetic.make_gaussians): plane = what IBGS / PGSR train towards")
    ap.add_argument("--scale-sigma", type=float, default=0.0,
                    help="NOT the BASELINE workload: log-normal spread of the Gaussian sizes (a heavy tail of large ones, as densification leaves it)")
    ap.add_argument("--no-trained-geo-line", action="store_true", help="skip the `trained_geo` object (the trainer's steady-state workload) of the default line")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", default="factored", choices=["factored", "dense"],
                    help="N > 1: factored = all-gather 3-float dL/dRGB per view + local SH expansion (default); dense = all-reduce of every gradient")
    ap.add_argument("--agree-every", type=int, default=1,
                    help="N > 1 ranks: run the reducer's host-side agreement every K-th step only (and whenever sizes change); 1 = every step")
    ap.add_argument("--forward-only", action="store_true", help="time the forward render alone (BASELINE configs[1]: --config C2 --forward-only)")
    ap.add_argument("--geo", action="store_true", help="make render_geo=True, n_src=4, L=4 the timed workload (second line of SURVEY 8(d))")
    ap.add_argument("--wave-shape", default=None, choices=["tile", "quadrant"], help="force the blend kernels' work decomposition (experiments)")
    ap.add_argument("--no-geo-line", action="store_true", help="skip the extra (untimed for `value`) geo measurement in the default line")
    ap.add_argument("--force-exchange", action="store_true", help="--gpus 1 only: initialise a process group of ONE rank (nccl = RCCL) and run the view-parallel exchange through it -- every "
                    "collective of the N-GPU step is issued at world size 1; the line gains the `rccl` object (exchange_ms = the fixed cost of the path)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra keys (view hopping, test-time frame, torch-L1 step, C1 CPU baseline protocol)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a, sys.argv[1:]))

    # The contract is ONE JSON line on stdout.  Native libraries write there too (gloo's "[Gloo] Rank 0 is connected ..." when
    # a process group comes up): keep a private handle on the real stdout for the JSON and point fd 1 at stderr for the rest.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    forced = bool(a.force_exchange) and a.gpus == 1
    rank, world, local_rank = vdist.init_from_env(backend=os.environ.get("IBGS_DIST_BACKEND"), force=forced)   # default: nccl (= RCCL)
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    ndev = torch.cuda.device_count()
    backend = dist.get_backend() if (world > 1 or forced) else None
    if world > ndev and not (os.environ.get("IBGS_BENCH_SHARE_GPU") == "1" and backend == "gloo"):
        raise SystemExit("bench.py: %d ranks but %d GPU(s)" % (world, ndev))
    local_rank %= max(ndev, 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    _lib.load()

    Workload.CLUSTER, Workload.ANISOTROPY, Workload.SCALE_SIGMA = a.cluster, a.anisotropy, a.scale_sigma
    if a.wave_shape:
        from ibgs_amd import rasterizer as _rz
        _rz.WAVE_SHAPE = a.wave_shape
    wl = Workload(a.config, rank % 8, dev, a.opacity, a.geo, a.forward_only, 1234 + rank)
    reducer = None
    if world > 1 or forced:
        # the leaves go into the rasterizer as they are: its backward writes their gradients straight into the all-reduce bucket
        reducer = vdist.ViewParallelReducer(wl.params, sh=wl.leaves["shs"], means3D=wl.leaves["means3D"], factored=(a.exchange == "factored"),
                                            agree_every=a.agree_every, force=forced,
                                            direct={k: wl.leaves[k] for k in ("means3D", "opacities", "scales", "rotations")})

    def step():
        if reducer is None or a.forward_only:
            return wl.local_step()

        def bw(loss):
            with reducer.capture():
                loss.backward()
        wl.local_step(bw)
        reducer.reduce()

    m = measure(wl, step, a.steps, a.warmup, world)
    rccl = None
    if (world > 1 or forced) and not a.forward_only:
        # serial cost of the exchange alone (hipEvents around reduce() on the compute stream, which waits for the collectives),
        # and the step without any exchange; neither enters `value`
        n_x = 5
        xs = []
        for _ in range(n_x):
            def bw(loss):
                with reducer.capture():
                    loss.backward()
            wl.local_step(bw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); reducer.reduce(); e1.record()
            torch.cuda.synchronize()
            xs.append(e0.elapsed_time(e1))
        fence(world)
        t0 = time.perf_counter()
        for _ in range(n_x):
            wl.local_step()
        fence(world)
        local_ms = (time.perf_counter() - t0) / n_x * 1e3
        rccl = {"world": dist.get_world_size(), "backend": backend, "exchange": a.exchange, "exchange_ms": statistics.median(xs),
                "step_without_exchange_ms": local_ms, "exposed_ms": m["ms_step"] - local_ms,
                "bytes_per_rank": reducer.last_bytes,
                # host time of the agreement round trip (the GPU keeps running the backward meanwhile), how often it ran, and how many
                # parameter gradients still had to be copied into the flat bucket (0 = the backward wrote them there itself)
                "agree_every": a.agree_every, "agree_host_ms": reducer.last_agree_ms, "agreements": reducer.n_agreements,
                "grads_copied_into_bucket": reducer.last_packed,
                # where in the step the collectives can start: render_bwd leaves per-Gaussian MOMENT rows (16 floats), not gradients -- dL/dmeans3D, dL/dscales,
                # dL/drotations, dL/dopacity and the clamp-masked dL/dRGB factors all come out of preprocess_bwd (~0.09 ms at 1 M Gaussians), so nothing is final before it
                "exchange_start_after": "preprocess_bwd"}

    extras = {}
    skip = set(filter(None, os.environ.get("IBGS_BENCH_SKIP", "").split(",")))          # diagnostics: leave named extras out (torch_l1, abs, hint, impl, hop)
    if world == 1 and not (a.geo or a.forward_only) and not a.no_extras:
        from ibgs_amd import rasterizer as _r
        # (e) the same step with the one-pass fused L1 (ibgs_amd.losses: value + gradient in one kernel each) instead of the reference's expression (six
        # small torch kernels): what the out-of-scope loss costs the step.  Never `value` (round 5: until round 4 the fused loss WAS the timed step's)
        Workload.torch_l1 = "fused_l1" in skip
        for _ in range(3):
            wl.local_step()
        fence(1); t0 = time.perf_counter()
        for _ in range(a.steps):
            wl.local_step()
        fence(1)
        extras["ms_per_step_with_fused_l1"] = (time.perf_counter() - t0) / a.steps * 1e3
        Workload.torch_l1 = True
        # (f) the same step when nobody asks for the densification statistic |dL/dmean2D| (means2D_abs without requires_grad: after densify_until_iter,
        # train.py:400-410, and at test time): IBGS_FLAG_NO_ABS_GRAD, the colour blend skips the two |.| moments.  Never `value`.
        wl.leaves["means2D_abs"].requires_grad_("abs" in skip)
        for _ in range(3):
            wl.local_step()
        fence(1); t0 = time.perf_counter()
        for _ in range(a.steps):
            wl.local_step()
        fence(1)
        extras["ms_per_step_without_abs_grad"] = (time.perf_counter() - t0) / a.steps * 1e3
        wl.leaves["means2D_abs"].requires_grad_(True)
        # the timed loop renders ONE camera again and again, so the forward's launch order hint (the previous backward's balanced order for this
        # camera, rasterizer.ORDER_HINT) is always fresh; a trainer revisits a camera only every few hundred steps.  The same step without it:
        _r.ORDER_HINT = "hint" in skip
        for _ in range(3):
            wl.local_step()
        fence(1); t0 = time.perf_counter()
        for _ in range(a.steps):
            wl.local_step()
        fence(1)
        extras["ms_per_step_without_forward_order_hint"] = (time.perf_counter() - t0) / a.steps * 1e3
        _r.ORDER_HINT = True
        # (d) fraction of HBM peak on the bytes THIS design moves (its sort is P-sized; SURVEY's model charges the reference's R-sized one)
        wl.local_step(); torch.cuda.synchronize()
        R_, C_, _m = _lib.last_forward_stats()
        with torch.no_grad():
            kept = int((wl._call()[1] > 0).sum().item())
        if C_ > 0:
            ib = implementation_bytes(wl.P, kept, R_, C_, wl.H * wl.W, int(wl.inp["shs"].shape[1]))
            extras["implementation_bytes"] = {"bytes_per_step": ib, "coarse_entries": C_, "gaussians_with_tiles": kept,
                                              "step_frac_impl": ib / (m["ms_step"] * 1e-3) / HBM_PEAK}
            stm = step_traffic_measured(workload_tag(a.config, False, False, a.opacity, a.cluster, a.anisotropy, a.scale_sigma))
            if stm is not None:          # the model above beside what the counters saw
                stm["step_frac_measured"] = stm["bytes_per_step"] / (m["ms_step"] * 1e-3) / HBM_PEAK
                extras["implementation_bytes"]["step_traffic_measured"] = stm
        # (a) a trainer hops between cameras (train.py:275-281): 8 orbit views round-robin.  The first round fills the window of the
        # R hint (a miss = binning + render run twice); afterwards every call is sized by the largest R of the last 16
        miss0 = _r.HINT_MISSES
        for k in range(0 if "hop" in skip else 8):
            wl.hop_to(k); wl.local_step()
        fence(1)
        miss1 = _r.HINT_MISSES
        n_hop = 24
        t0 = time.perf_counter()
        for k in range(n_hop):
            wl.hop_to(0 if "hop" in skip else k % 8); wl.local_step()
        fence(1)
        extras["view_hopping"] = {"views": 8, "steps": n_hop, "ms_per_step": (time.perf_counter() - t0) / n_hop * 1e3,
                                  "hint_misses_first_round": miss1 - miss0, "hint_misses_steady": _r.HINT_MISSES - miss1,
                                  "note": "same Gaussians, 8 orbit cameras round-robin, fwd+bwd; R differs per camera"}
        wl.hop_to(rank % 8)

    def geo_object(gwl, opacity, gsteps):
        gm = measure(gwl, gwl.local_step, gsteps, 10, 1, n_fwd=5)
        grf = roofline(gwl, gm, workload_tag(a.config, True, False, opacity, gwl.cluster, gwl.anisotropy, gwl.scale_sigma), walked=walked_entries(gwl))
        with torch.no_grad():
            radii = gwl._call()[1]
            big = int((radii > 128).sum().item()); vis = int((radii > 0).sum().item())
        # the four L1 terms of this step (image, normal map, median depth, 15 warped planes) are the reference's torch expression; the same step with the
        # one-pass fused L1 (out of scope) beside it, and the library's own kernels alone (sum of the stage timers: loss-independent)
        Workload.torch_l1 = False
        fused_ms = timed_wall_ms(gwl.local_step, gsteps, warmup=5)
        Workload.torch_l1 = True
        return {"workload": gwl.describe(opacity, 1, a.exchange), "steps": gsteps, "ms_per_step": gm["ms_step"], "ms_per_step_with_fused_l1": fused_ms,
                "library_kernels_ms": sum(gm["stages"].values()),
                "median_ms_hipevent": gm["median_ms"], "max_ms_hipevent": gm["max_ms"], "per_step_ms_hipevent": gm["per_step"], "fps": 1000.0 / gm["ms_step"], "forward_only_ms": gm["fwd_ms"],
                "num_rendered": int(gwl.R), "gaussians_in_frustum": vis, "gaussians_radius_gt_128px": big, "stages_ms": gm["stages"], "roofline": grf}

    geo_line = trained_geo = None
    if world == 1 and not (a.geo or a.forward_only or a.no_geo_line):
        # second line of SURVEY 8(d) under the same clock: C3 + render_geo, n_src 4, L 4 (extra key; never part of `value`)
        del step
        torch.cuda.empty_cache()
        gwl = Workload(a.config, rank % 8, dev, a.opacity, True, False, 1234 + rank)
        geo_line = geo_object(gwl, a.opacity, max(5, min(20, a.steps)))
        del gwl
        torch.cuda.empty_cache()          # the next workload's blocks are then allocated during ITS warm-up (a 54 ms step appeared once in the middle of
                                          # the trained_geo line: the caching allocator releasing and re-allocating the previous workload's blocks)
    if world == 1 and not (a.geo or a.forward_only or a.no_trained_geo_line):
        # what train.py:289-292 runs for ~77 % of its iterations: render_geo on TRAINED Gaussians -- plane-like (scene/gaussian_model.py:156-173),
        # a heavy tail of sizes (densification, :580-604), an uneven image (30 % of them in one blob), trained-like opacities.  Extra key; never `value`
        twl = Workload(a.config, rank % 8, dev, "trained", True, False, 1234 + rank, cluster=0.3, anisotropy="plane", scale_sigma=1.0)
        trained_geo = geo_object(twl, "trained", max(5, min(20, a.steps)))
        del twl
        torch.cuda.empty_cache()

    if world == 1 and not (a.geo or a.forward_only) and not a.no_extras:
        extras["test_frame"] = test_frame(dev, wl.c)
        if "small_frame" not in skip:
            extras["small_frame"] = small_frame(dev, max(10, min(30, a.steps)))
        if "train_iter" not in skip and a.config == "C3":
            extras["train_iter"] = train_iter(dev, wl.c, max(8, min(16, a.steps)))
    if rank == 0:
        rf = roofline(wl, m, workload_tag(a.config, a.geo, a.forward_only, a.opacity, a.cluster, a.anisotropy, a.scale_sigma))
        out = {
            "metric": ("forward render fps " + a.config) if a.forward_only else
                      ("train-step fps (fwd+bwd raster) @1080p, 1M Gaussians" if a.config == "C3" else "train-step fps (fwd+bwd raster) " + a.config),
            "value": world * a.steps / m["dt"], "unit": "fps", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": m["ms_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl.describe(a.opacity, world, a.exchange), "num_rendered": int(wl.R), "parallelism": "view-parallel x%d" % world},
            "median_ms_hipevent": m["median_ms"], "min_ms_hipevent": m["min_ms"], "max_ms_hipevent": m["max_ms"],
            "per_step_ms_hipevent": m["per_step"],          # every timed step: where the wall clock differs from the median, it is these outliers
            "forward_only_ms": m["fwd_ms"],
            "roofline": rf,
            "stages_ms": m["stages"],
        }
        if rccl is not None:
            out["rccl"] = rccl
            if forced:
                out["forced_exchange"] = True          # the timed step ran the N-GPU exchange at world size 1 (its fixed cost is inside `value`)
        if geo_line is not None:
            out["geo"] = geo_line
        if trained_geo is not None:
            out["trained_geo"] = trained_geo
        out.update(extras)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl.inp, a.config)
            if not a.no_extras:
                out["cpu_baseline"]["c1_protocol"] = cpu_baseline_c1(dev)
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
    if world > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
