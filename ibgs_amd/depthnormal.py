"""Depth -> normal map of the render glue as ONE kernel each way instead of ~25 torch kernels each way.

`depth_normal(viewpoint_cam, depth)` = `render_normal(viewpoint_cam, depth)` of the reference (gaussian_renderer/__init__.py:16-26: finite
differences of the back-projected points, utils/graphics_utils.py:17-83) followed by the normalisation `render()` applies to it (:338-342):
returns the (3, H, W) unit normals, differentiable in `depth`.  HIP only (C ABI `ibgs_depth_normal_forward / _backward`,
ibgs_amd/csrc/depth_normal.hip); `renderer.FUSED_DEPTH_NORMAL = False` keeps the torch formulation (the behavioural definition, tests compare)."""
import torch

from . import _lib


class _DepthNormal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, fx, fy, cx, cy):
        if not depth.is_cuda:
            raise RuntimeError("depth_normal runs on the MI355X only (no CPU path)")
        lib = _lib.load()
        d = depth.detach().float().contiguous()
        H, W = int(d.shape[0]), int(d.shape[1])
        out = torch.empty(3, H, W, dtype=torch.float32, device=d.device)
        with torch.cuda.device(d.device):
            rc = lib.ibgs_depth_normal_forward(torch.cuda.current_stream(d.device).cuda_stream, W, H, fx, fy, cx, cy, d.data_ptr(), out.data_ptr())
        if rc < 0:
            raise RuntimeError("ibgs_depth_normal_forward failed (%d): %s" % (rc, _lib.last_error()))
        ctx.save_for_backward(d)
        ctx.k = (fx, fy, cx, cy)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        if not ctx.needs_input_grad[0]:
            return None, None, None, None, None
        (d,) = ctx.saved_tensors
        lib = _lib.load()
        H, W = int(d.shape[0]), int(d.shape[1])
        g = grad_out.detach().float().contiguous()
        gd = torch.empty(H, W, dtype=torch.float32, device=d.device)
        with torch.cuda.device(d.device):
            rc = lib.ibgs_depth_normal_backward(torch.cuda.current_stream(d.device).cuda_stream, W, H, *ctx.k, d.data_ptr(), g.data_ptr(), gd.data_ptr())
        if rc < 0:
            raise RuntimeError("ibgs_depth_normal_backward failed (%d): %s" % (rc, _lib.last_error()))
        return gd, None, None, None, None


def depth_normal(viewpoint_cam, depth):
    """depth: (H, W) on the device.  Intrinsics as `Camera.get_calib_matrix_nerf(scale=1)` builds them (scene/cameras.py:118-121): Fx, Fy, Cx, Cy."""
    return _DepthNormal.apply(depth, float(viewpoint_cam.Fx), float(viewpoint_cam.Fy), float(viewpoint_cam.Cx), float(viewpoint_cam.Cy))
