"""The trainer's optimiser step on the MI355X in one launch.

`FusedAdam` is a drop-in for the `torch.optim.Adam(l, lr=0.0, eps=1e-15)` of `GaussianModel.training_setup`
(reference scene/gaussian_model.py:227-241): same constructor, same `param_groups`, same per-parameter state
(`step`, `exp_avg`, `exp_avg_sq`), so the reference's densification surgery on the optimiser state
(`cat_tensors_to_optimizer`, `_prune_optimizer`, `replace_tensor_to_optimizer`, gaussian_model.py:423-497) keeps
working.  Only `step()` differs: all parameter tensors are updated by ONE HIP kernel (C ABI `ibgs_adam_step`,
ibgs_amd/csrc/adam.hip) instead of one kernel (fused=True) or ~10 (foreach) per tensor.

Round 6 -- the SH coefficients without their dense gradient.  For one view dL/dsh is the outer product basis(dir) x dL/dRGB; written out it is 192 B per Gaussian that the
backward stores and the optimiser reads back.  With

    with rasterizer.capture_sh_factors() as factors:
        loss.backward()
    optimizer.step(sh_factors=factors, sh_params=(gaussians._features_dc, gaussians._features_rest), means3D=gaussians._xyz)

the backward leaves dL/dsh unwritten (`.grad` of the two tensors stays None) and `step` updates them straight from the factors (C ABI `ibgs_adam_step_sh`): parameters
and moments bit-identical to expanding the factors (`shgrad.sh_grad_from_views`) and stepping densely (tests/test_gpu_adam.py); against the plain backward's own dL/dsh
the gradient agrees to an ulp (the view direction is normalised in two places).  The iteration is ~0.06-0.13 ms shorter at 1 M Gaussians."""
import ctypes
import math

import torch

from . import _lib


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        if weight_decay != 0 or amsgrad or kw.get("maximize", False):
            raise NotImplementedError("FusedAdam covers the reference's use of Adam: no weight decay, amsgrad or maximize")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)

    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _step_sh(self, lib, sh_factors, sh_params, means3D):
        """The coefficient tensors `sh_params` (in coefficient order: (f_dc, f_rest) or one combined tensor) from the factors of this step's backward(s)."""
        items = list(sh_factors)
        if not items:
            return set()
        params = [p for p in sh_params if p is not None]
        if not 1 <= len(params) <= 2:
            raise ValueError("FusedAdam.step: sh_params holds the one or two SH coefficient tensors")
        dev = params[0].device
        P = int(params[0].shape[0])
        if P == 0:
            return set(id(p) for p in params)
        degree, M = int(items[0]["degree"]), int(items[0]["M"])
        if any(int(it["degree"]) != degree or int(it["M"]) != M or tuple(it["dcolor"].shape) != (P, 3) for it in items):
            raise ValueError("FusedAdam.step: the factors of one step must share degree, M and P")
        if sum(int(p.shape[1]) for p in params) != M or M > 16:
            raise ValueError("FusedAdam.step: sh_params hold %d coefficients per Gaussian, the factors were recorded for %d" % (sum(int(p.shape[1]) for p in params), M))
        dcolor = items[0]["dcolor"].contiguous() if len(items) == 1 else torch.stack([it["dcolor"] for it in items]).contiguous()
        cams = torch.stack([it["campos"].to(dev).reshape(3) for it in items]).contiguous()
        m3 = means3D.detach()
        if not (m3.is_cuda and m3.dtype == torch.float32 and m3.is_contiguous() and tuple(m3.shape) == (P, 3)):
            raise RuntimeError("FusedAdam.step: means3D must be the contiguous fp32 (P, 3) positions on the MI355X")
        group_of = {id(p): g for g in self.param_groups for p in g["params"]}
        ds, k0s, Ks, k0 = [], [], [], 0
        for p in params:
            if p.grad is not None:
                raise RuntimeError("FusedAdam.step: an SH tensor has a dense .grad beside the factors (a loss term outside the rasterizer?): step without sh_factors")
            if id(p) not in group_of:
                raise ValueError("FusedAdam.step: sh_params must be parameters of this optimiser")
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.dim() == 3 and int(p.shape[0]) == P and int(p.shape[2]) == 3):
                raise RuntimeError("FusedAdam: SH tensors must be contiguous fp32 (P, K, 3) tensors on the MI355X")
            group = group_of[id(p)]
            b1, b2 = group["betas"]
            st = self._state_of(p)
            st["step"] += 1
            t = float(st["step"])
            m, v = st["exp_avg"], st["exp_avg_sq"]
            if not (m.is_contiguous() and v.is_contiguous()):
                m = st["exp_avg"] = m.contiguous(); v = st["exp_avg_sq"] = v.contiguous()
            d = _lib.AdamTensor()
            d.param, d.grad, d.exp_avg, d.exp_avg_sq = p.data_ptr(), None, m.data_ptr(), v.data_ptr()
            d.numel = p.numel()
            d.lr, d.beta1, d.beta2, d.eps = float(group["lr"]), float(b1), float(b2), float(group["eps"])
            d.bias_correction1 = 1.0 - math.pow(b1, t); d.bias_correction2 = 1.0 - math.pow(b2, t)
            K = int(p.shape[1])
            ds.append(d); k0s.append(k0); Ks.append(K); k0 += K
        arr = (_lib.AdamTensor * len(ds))(*ds)
        a0, aK = (ctypes.c_int32 * len(ds))(*k0s), (ctypes.c_int32 * len(ds))(*Ks)
        with torch.cuda.device(dev):
            rc = lib.ibgs_adam_step_sh(torch.cuda.current_stream(dev).cuda_stream, P, degree, len(items), m3.data_ptr(), cams.data_ptr(), dcolor.data_ptr(), 3 * P,
                                       len(ds), ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(a0, ctypes.c_void_p), ctypes.cast(aK, ctypes.c_void_p))
        if rc < 0:
            raise RuntimeError("ibgs_adam_step_sh failed (%d): %s" % (rc, _lib.last_error()))
        return set(id(p) for p in params)

    @torch.no_grad()
    def step(self, closure=None, sh_factors=None, sh_params=None, means3D=None):
        """sh_factors / sh_params / means3D: see the module docstring (the SH coefficients from `rasterizer.capture_sh_factors()` records instead of a dense gradient)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        if sh_factors is not None:
            if sh_params is None or means3D is None:
                raise ValueError("FusedAdam.step(sh_factors=...) needs sh_params and means3D")
            self._step_sh(lib, sh_factors, sh_params, means3D)          # (before the dense tensors: it reads the positions this step is about to move)
        batch, keep = [], []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and not p.grad.is_sparse):
                    raise RuntimeError("FusedAdam: parameters must be contiguous fp32 tensors on the MI355X")
                st = self._state_of(p)
                st["step"] += 1
                t = float(st["step"])
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous()):
                    m = st["exp_avg"] = m.contiguous(); v = st["exp_avg_sq"] = v.contiguous()
                d = _lib.AdamTensor()
                d.param, d.grad, d.exp_avg, d.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
                d.numel = p.numel()
                d.lr, d.beta1, d.beta2, d.eps = float(group["lr"]), float(b1), float(b2), float(group["eps"])
                d.bias_correction1 = 1.0 - math.pow(b1, t); d.bias_correction2 = 1.0 - math.pow(b2, t)
                batch.append((p.device, d)); keep.append(g)
        by_dev = {}
        for dev, d in batch:
            by_dev.setdefault(dev, []).append(d)
        for dev, ds in by_dev.items():
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                for i in range(0, len(ds), 16):
                    part = ds[i:i + 16]
                    arr = (_lib.AdamTensor * len(part))(*part)
                    rc = lib.ibgs_adam_step(stream, len(part), ctypes.cast(arr, ctypes.c_void_p))
                    if rc < 0:
                        raise RuntimeError("ibgs_adam_step failed (%d): %s" % (rc, _lib.last_error()))
        return loss
