"""The photometric L1 term of the reference's training step (`l1_loss`, utils/loss_utils.py:23-24; train.py:302) as one pass over the
image: value and gradient together (C ABI `ibgs_l1_loss`, csrc/loss.hip).  Same number as `torch.abs(a - b).mean()`, same gradient
`sign(a - b) / N`; torch needs six small kernels for the pair, ~70 us on a 1080p image.

HIP only: raises when the library or a GPU tensor is missing (no torch fallback in the product path)."""
import torch

from . import _lib

_scratch = {}


class _L1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, target):
        if not image.is_cuda:
            raise RuntimeError("ibgs_amd.losses.l1_loss runs on the MI355X only (no CPU path)")
        if image.shape != target.shape:
            raise ValueError("l1_loss: shapes differ: %s vs %s" % (tuple(image.shape), tuple(target.shape)))
        lib = _lib.load()
        x = image.detach().float().contiguous()
        y = target.detach().to(x.device).float().contiguous()
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream(x.device).cuda_stream
            key = (x.device.index, stream)
            sc = _scratch.get(key)
            if sc is None:
                if len(_scratch) > 8:
                    _scratch.clear()
                sc = _scratch[key] = torch.empty(lib.ibgs_required_l1(), dtype=torch.uint8, device=x.device)
            # value AND gradient sign(x - y) / N in the one pass over x and y (when a gradient will be asked for): the backward then only has to
            # scale it by the incoming gradient -- and not even that when the term enters the total with weight one
            grad = torch.empty_like(x) if (ctx.needs_input_grad[0] and torch.is_grad_enabled()) else None          # (needs_input_grad ignores no_grad())
            rc = lib.ibgs_l1_loss(stream, x.numel(), x.data_ptr(), y.data_ptr(), None if grad is None else grad.data_ptr(), loss.data_ptr(), sc.data_ptr(), sc.numel())
        if rc < 0:
            raise RuntimeError("ibgs_l1_loss failed (%d): %s" % (rc, _lib.last_error()))
        ctx.save_for_backward(x, y)
        ctx.unit_grad = grad          # consumed (scaled in place) by the first backward
        ctx.shape = image.shape
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        if not ctx.needs_input_grad[0]:
            return None, None
        x, y = ctx.saved_tensors
        lib = _lib.load()
        go = grad_out.detach().to(x.device).float().contiguous()
        grad, ctx.unit_grad = ctx.unit_grad, None
        if grad is not None:
            with torch.cuda.device(x.device):
                rc = lib.ibgs_l1_rescale(torch.cuda.current_stream(x.device).cuda_stream, grad.numel(), grad.data_ptr(), go.data_ptr())
            if rc < 0:
                raise RuntimeError("ibgs_l1_rescale failed (%d): %s" % (rc, _lib.last_error()))
            return grad.view(ctx.shape), None
        grad = torch.empty_like(x)          # a second backward through the same node (retain_graph): from x and y again
        with torch.cuda.device(x.device):
            # sign(x - y) * grad_out / N in ONE pass: the incoming gradient is read on the device (no host sync, no separate multiply)
            rc = lib.ibgs_l1_grad(torch.cuda.current_stream(x.device).cuda_stream, x.numel(), x.data_ptr(), y.data_ptr(), go.data_ptr(), grad.data_ptr())
        if rc < 0:
            raise RuntimeError("ibgs_l1_grad failed (%d): %s" % (rc, _lib.last_error()))
        return grad.view(ctx.shape), None      # (the target's gradient is never asked for by the trainer)


def l1_loss(network_output, gt):
    """Drop-in for the reference's `l1_loss(network_output, gt)`: mean absolute difference, differentiable in `network_output`."""
    if network_output.numel() == 0:
        return torch.abs(network_output - gt).mean()
    return _L1.apply(network_output, gt)
