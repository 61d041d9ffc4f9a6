"""The trainer's optimiser step on the MI355X in one launch.

`FusedAdam` is a drop-in for the `torch.optim.Adam(l, lr=0.0, eps=1e-15)` of `GaussianModel.training_setup`
(reference scene/gaussian_model.py:227-241): same constructor, same `param_groups`, same per-parameter state
(`step`, `exp_avg`, `exp_avg_sq`), so the reference's densification surgery on the optimiser state
(`cat_tensors_to_optimizer`, `_prune_optimizer`, `replace_tensor_to_optimizer`, gaussian_model.py:423-497) keeps
working.  Only `step()` differs: all parameter tensors are updated by ONE HIP kernel (C ABI `ibgs_adam_step`,
ibgs_amd/csrc/adam.hip) instead of one kernel (fused=True) or ~10 (foreach) per tensor."""
import ctypes
import math

import torch

from . import _lib


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        if weight_decay != 0 or amsgrad or kw.get("maximize", False):
            raise NotImplementedError("FusedAdam covers the reference's use of Adam: no weight decay, amsgrad or maximize")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        batch, keep = [], []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and not p.grad.is_sparse):
                    raise RuntimeError("FusedAdam: parameters must be contiguous fp32 tensors on the MI355X")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
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
