"""dL/dsh of a view-parallel step from the exchanged per-view factors (C ABI ``ibgs_sh_grad_from_views``,
ibgs_amd/csrc/preprocess_bwd.hip).  HIP only: raises when the library or a GPU tensor is missing."""
import torch

from . import _lib


def sh_grad_from_views(means3D, camposes, dcolor, degree, M):
    """means3D (P,3), camposes (V,3), dcolor (V,P,3) [row stride may exceed 3P], returns (P,M,3):
    sum_v basis(normalise(means3D - camposes[v]))[:, :, None] * dcolor[v][:, None, :], zero above the active degree."""
    if not means3D.is_cuda:
        raise RuntimeError("sh_grad_from_views runs on the MI355X only (no CPU path)")
    lib = _lib.load()
    dev = means3D.device
    P, V = int(means3D.shape[0]), int(camposes.shape[0])
    m = means3D.detach().float().contiguous()
    c = camposes.detach().to(dev).float().contiguous()
    d = dcolor.detach().float()
    assert d.shape == (V, P, 3) and d.stride(2) == 1 and d.stride(1) == 3 and (V <= 1 or d.stride(0) >= 3 * P)
    stride = int(d.stride(0)) if V > 1 else 3 * P
    out = torch.empty(P, M, 3, dtype=torch.float32, device=dev)
    if P == 0 or M == 0:
        return out
    with torch.cuda.device(dev):
        rc = lib.ibgs_sh_grad_from_views(torch.cuda.current_stream(dev).cuda_stream, P, int(degree), int(M), V, m.data_ptr(),
                                         c.data_ptr(), d.data_ptr(), stride, out.data_ptr())
    if rc < 0:
        raise RuntimeError("ibgs_sh_grad_from_views failed (%d): %s" % (rc, _lib.last_error()))
    return out
