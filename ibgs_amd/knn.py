"""distCUDA2 replacement: mean squared distance to the 3 nearest neighbours,
used once by GaussianModel.create_from_pcd to initialise the scales (reference scene/gaussian_model.py:195,
submodules/simple-knn/spatial.cu:15-26)."""
import torch

from . import _lib


def distCUDA2(points):
    lib = _lib.load()
    if points.ndimension() != 2 or points.size(1) != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    if not points.is_cuda:
        raise RuntimeError("points must live on a HIP device (libibgs_rast.so has no CPU path)")
    P = int(points.size(0))
    dev = points.device
    pts = points.detach().float().contiguous()
    out = torch.zeros(P, dtype=torch.float32, device=dev)
    if P:
        with torch.cuda.device(dev):
            scratch = torch.empty(lib.ibgs_required_knn(P), dtype=torch.uint8, device=dev)
            rc = lib.ibgs_knn_mean_dist2(torch.cuda.current_stream(dev).cuda_stream, P, pts.data_ptr(), out.data_ptr(),
                                         scratch.data_ptr(), scratch.numel())
            if rc < 0:
                raise RuntimeError("ibgs_knn_mean_dist2 failed (%d): %s" % (rc, _lib.last_error()))
    return out
