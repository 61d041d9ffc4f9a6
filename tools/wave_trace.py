"""Where and when did the waves of the blend kernels run?  (diagnostic build, docs/EXPERIMENTS.md section 7 round 3)
Builds a copy of the library with -DIBGS_TRACE_WAVES (every workgroup of render_fwd_kernel / render_bwd_color_kernel stamps its SIMD and its
start / end on the 100 MHz clock), runs C3 forward + backward once, and reports per kernel: how many waves each SIMD got, when the SIMDs
finished relative to the kernel's span (a SIMD that is done at 60 % idles for the rest), i.e. what perfect balance could buy.
usage: python tools/wave_trace.py [init|trained] [--cluster F]"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ibgs_amd import _build

obj = os.path.join(_build.OBJ)
lib = "/tmp/libibgs_trace.so"
objs = []
for name in _build.SOURCES:
    o = os.path.join(obj, name + ".o")
    if name in ("render_fwd", "render_bwd"):
        o = "/tmp/trace_%s.o" % name
        subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DIBGS_TRACE_WAVES", "-c",
                               os.path.join(_build.CSRC, name + ".hip"), "-o", o] + _build.EXTRA.get(name, []))
    objs.append(o)
subprocess.check_call([_build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)

import torch
from ibgs_amd import _lib
_lib.LIB_PATH = lib
from ibgs_amd import rasterizer, synthetic as syn
from tests import hipref

opacity = "trained" if "trained" in sys.argv else "init"
cluster = float(sys.argv[sys.argv.index("--cluster") + 1]) if "--cluster" in sys.argv else 0.0
c = syn.CONFIGS["C3"]
inp = syn.make_scene(c["P"], c["W"], c["H"], sh_degree=3, seed=c["seed"], opacity=opacity)
if cluster > 0:          # as bench.py --cluster: that share of the Gaussians in one blob
    k = int(cluster * c["P"])
    inp["means3D"] = inp["means3D"].copy(); inp["means3D"][:k] = inp["means3D"][:k] * 0.3 + np.array([0.5, 0.25, 0.0], np.float32)
L = _lib.load()
for _ in range(3):          # warm
    outs, lv, _ = hipref.run_forward(inp)
    g = torch.randn(3, c["H"], c["W"], device="cuda")
    (outs["color"] * g).sum().backward()
torch.cuda.synchronize()
N = 32768
raw = ctypes.CDLL(lib)
for which in ("fwd", "bwd"):
    buf = np.zeros((N, 4), np.uint32)
    rc = getattr(raw, "ibgs_debug_trace_" + which)(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
    assert rc == 0
    nt = ((c["W"] + 15) // 16) * ((c["H"] + 15) // 16)
    used = buf[(buf[:, 3] != 0)]
    # the buffer is indexed by workgroup and survives launches: keep the stamps of the LAST launch only (earlier launches of another grid size leave theirs behind)
    last = used[:, 3].astype(np.int64).max()
    keep = used[:, 2].astype(np.int64) >= last - 200000          # started within 2 ms of the last end (100 MHz ticks)
    buf = np.where(((buf[:, 3] != 0) & (buf[:, 2].astype(np.int64) >= last - 200000))[:, None], buf, 0)
    used = used[keep]
    hw, xcc, t0, t1 = used[:, 0], used[:, 1] & 0xF, used[:, 2].astype(np.int64), used[:, 3].astype(np.int64)
    # HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    span0, span1 = t0.min(), t1.max()
    span = float(span1 - span0)
    ids, inv = np.unique(key, return_inverse=True)
    nw = np.bincount(inv)
    fin = np.zeros(len(ids)); np.maximum.at(fin, inv, (t1 - span0).astype(np.float64))
    start_last = np.zeros(len(ids)); np.maximum.at(start_last, inv, (t0 - span0).astype(np.float64))
    dur = (t1 - t0).astype(np.float64)
    print("%s (opacity=%s): %d stamped workgroups on %d SIMDs; kernel span %.1f us" % (which, opacity, len(used), len(ids), span / 100.0))
    print("   waves per SIMD: min %d mean %.2f max %d; all waves started within %.1f us" % (nw.min(), nw.mean(), nw.max(), start_last.max() / 100.0))
    print("   SIMD finish time / span: mean %.3f, percentiles 5/25/50/75/95: %s  => perfectly balanced SIMDs would end at ~%.0f %% of today's span"
          % ((fin / span).mean(), np.round(np.percentile(fin / span, [5, 25, 50, 75, 95]), 3), 100.0 * (fin / span).mean()))
    print("   wave duration / span: mean %.3f, p5 %.3f, p95 %.3f" % ((dur / span).mean(), np.percentile(dur / span, 5), np.percentile(dur / span, 95)))
    real = dur > 0.02 * span          # (workgroups past the tile grid leave at once)
    st = (t0 - span0).astype(np.float64)[real] / 100.0
    nreal = np.bincount(inv[real], minlength=len(ids))
    print("   real tiles %d; per SIMD min %d max %d; started later than 5 us: %d, later than 50 us: %d (start percentiles 90/99/100: %s us)"
          % (real.sum(), nreal.min(), nreal.max(), (st > 5).sum(), (st > 50).sum(), np.round(np.percentile(st, [90, 99, 100]), 1)))
    conc = np.zeros(len(ids)); np.add.at(conc, inv[real], ((t0 - span0)[real] < 500).astype(np.float64))
    print("   real tiles resident per SIMD 5 us after the launch: min %d mean %.2f max %d" % (conc.min(), conc.mean(), conc.max()))
    if "--slowest" in sys.argv:          # who ends last: a work outlier or a hardware one?
        o = np.argsort(-fin)[:12]
        for k in o:
            sel = inv == k
            print("   SIMD xcc %d se %d sh %d cu %2d simd %d: finish %.3f, waves %d, sum of wave durations %.2f x span, first 4 workgroups %s"
                  % ((ids[k] >> 9) & 15, (ids[k] >> 6) & 7, (ids[k] >> 5) & 1, (ids[k] >> 2) & 15 if False else cu[sel][0], ids[k] & 3, fin[k] / span, sel.sum(), dur[sel].sum() / span,
                     np.sort(np.nonzero(buf[:, 3] != 0)[0][sel])[:4]))
        cuk = key >> 2
        cids, cinv = np.unique(cuk, return_inverse=True)
        cfin = np.zeros(len(cids)); np.maximum.at(cfin, cinv, (t1 - span0).astype(np.float64))
        print("   per-CU finish / span percentiles 5/50/95/100:", np.round(np.percentile(cfin / span, [5, 50, 95, 100]), 3), " CUs: %d" % len(cids))
    if "--placement" in sys.argv:          # which workgroup indices share a SIMD?  (is the dispatcher's placement regular enough to plan for?)
        wg = np.nonzero(buf[:, 3] != 0)[0]
        for k in list(range(3)) + [len(ids) // 2, len(ids) - 1]:
            print("   SIMD key %d (xcc %d): workgroups %s" % (ids[k], ids[k] >> 9, np.sort(wg[inv == k])[:12]))
        print("   xcc of workgroups 0..15:", xcc[np.argsort(wg)][:16], " (workgroup %% 8 == xcc for %.1f %%)" % (100.0 * np.mean((wg % 8) == xcc[:])))
        # within an XCD: position k = workgroup // 8; does SIMD = f(k % 128)?
        kpos = wg // 8
        same = 0; tot = 0
        for x in range(8):
            m = xcc == x
            d = {}
            for kp, ky in zip(kpos[m] % 128, key[m]):
                d.setdefault(kp, set()).add(ky)
            same += sum(1 for v in d.values() if len(v) == 1); tot += len(d)
        print("   residues (workgroup // 8) %% 128 that always land on one SIMD: %d of %d" % (same, tot))
    byx = np.zeros(8); np.maximum.at(byx, xcc, (t1 - span0).astype(np.float64))
    print("   per-XCD finish / span:", np.round(byx / span, 3))
