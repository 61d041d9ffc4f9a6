"""The model's three activations as ONE kernel each way (model side; C ABI `ibgs_activate_forward / _backward`, ibgs_amd/csrc/activate.hip).

`fused_activations(raw_scaling, raw_rotation, raw_opacity)` = `(torch.exp(raw_scaling), F.normalize(raw_rotation), torch.sigmoid(raw_opacity))` -- what
`GaussianModel.get_scaling / get_rotation / get_opacity` (scene/gaussian_model.py:44-52, 128-147) compute on every render() call -- differentiable in all
three.  HIP only.  `renderer.FUSED_ACTIVATIONS = False` keeps the model's own properties (the behavioural definition; tests compare)."""
import torch

from . import _lib


def _ptr(t):
    return None if t is None else t.data_ptr()


class _Activate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw_scaling, raw_rotation, raw_opacity):
        if not raw_scaling.is_cuda:
            raise RuntimeError("fused_activations runs on the MI355X only (no CPU path)")
        lib = _lib.load()
        rs, rr, ro = (t.detach().float().contiguous() for t in (raw_scaling, raw_rotation, raw_opacity))
        P = int(rs.shape[0])
        if tuple(rs.shape) != (P, 3) or tuple(rr.shape) != (P, 4) or ro.numel() != P:
            raise RuntimeError("fused_activations: scaling (P, 3), rotation (P, 4), opacity (P, 1) expected")
        s, r, o = torch.empty_like(rs), torch.empty_like(rr), torch.empty_like(ro)
        with torch.cuda.device(rs.device):
            rc = lib.ibgs_activate_forward(torch.cuda.current_stream(rs.device).cuda_stream, P, rs.data_ptr(), rr.data_ptr(), ro.data_ptr(), s.data_ptr(), r.data_ptr(), o.data_ptr())
        if rc < 0:
            raise RuntimeError("ibgs_activate_forward failed (%d): %s" % (rc, _lib.last_error()))
        ctx.save_for_backward(rs, rr, ro)
        return s, r, o

    @staticmethod
    def backward(ctx, g_s, g_r, g_o):
        rs, rr, ro = ctx.saved_tensors
        lib = _lib.load()
        P = int(rs.shape[0])
        want = ctx.needs_input_grad
        gs = g_s.detach().float().contiguous() if (want[0] and g_s is not None) else None
        gr = g_r.detach().float().contiguous() if (want[1] and g_r is not None) else None
        go = g_o.detach().float().contiguous() if (want[2] and g_o is not None) else None
        ds = torch.empty_like(rs) if gs is not None else None
        dr = torch.empty_like(rr) if gr is not None else None
        do = torch.empty_like(ro) if go is not None else None
        if P and (ds is not None or dr is not None or do is not None):
            with torch.cuda.device(rs.device):
                rc = lib.ibgs_activate_backward(torch.cuda.current_stream(rs.device).cuda_stream, P, rs.data_ptr(), rr.data_ptr(), ro.data_ptr(),
                                                _ptr(gs), _ptr(gr), _ptr(go), _ptr(ds), _ptr(dr), _ptr(do))
            if rc < 0:
                raise RuntimeError("ibgs_activate_backward failed (%d): %s" % (rc, _lib.last_error()))
        return ds, dr, do


def fused_activations(raw_scaling, raw_rotation, raw_opacity):
    return _Activate.apply(raw_scaling, raw_rotation, raw_opacity)


def standard_model(pc):
    """True when `pc` keeps its parameters and activations the way the reference's GaussianModel does (scene/gaussian_model.py:44-52): raw `_scaling`,
    `_rotation`, `_opacity` on the device, activated by exp / F.normalize / sigmoid.  A model without the `*_activation` attributes (ibgs_amd.simple_scene)
    declares the same through `standard_activations = True`."""
    raw = [getattr(pc, n, None) for n in ("_scaling", "_rotation", "_opacity")]
    if not all(torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 for t in raw):
        return False
    if getattr(pc, "standard_activations", False) is True:
        return True
    return (getattr(pc, "scaling_activation", None) is torch.exp and getattr(pc, "opacity_activation", None) is torch.sigmoid
            and getattr(pc, "rotation_activation", None) is torch.nn.functional.normalize)
