"""`ibgs_amd.losses.l1_loss` (one pass: value + gradient, csrc/loss.hip) against the reference's formulation
`torch.abs(network_output - gt).mean()` (utils/loss_utils.py:23-24) and its autograd gradient."""
import pytest
import torch

from ibgs_amd.losses import l1_loss

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(3, 1080, 1920), (3, 17, 33), (1, 5), (3, 64, 64)])
def test_l1_loss_value_and_gradient(shape):
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(sum(shape))
    a = torch.rand(shape, device=dev, generator=g)
    b = torch.rand(shape, device=dev, generator=g)
    a.view(-1)[::7] = b.view(-1)[::7]                     # exact ties: sign(0) = 0
    a1 = a.clone().requires_grad_(True); a2 = a.clone().requires_grad_(True)
    ref = torch.abs(a1 - b).mean()
    got = l1_loss(a2, b)
    assert abs(float(got.detach()) - float(ref.detach())) <= 2e-6 * abs(float(ref.detach()))
    (ref * 0.75).backward(); (got * 0.75).backward()
    assert torch.equal(a2.grad == 0, a1.grad == 0)
    assert torch.allclose(a2.grad, a1.grad, rtol=1e-6, atol=0.0)


def test_l1_loss_is_reproducible_and_value_only_without_grad():
    dev = torch.device("cuda")
    a = torch.rand(3, 300, 500, device=dev); b = torch.rand(3, 300, 500, device=dev)
    v = [float(l1_loss(a, b)) for _ in range(3)]           # no gradient asked for: the gradient store is skipped
    assert v[0] == v[1] == v[2]
    assert abs(v[0] - float(torch.abs(a - b).mean())) <= 2e-6 * v[0]
    with pytest.raises(ValueError):
        l1_loss(a, b[:, :10])


def test_unweighted_term_and_repeated_backward():
    """The gradient is stored by the loss pass and only scaled in the backward (not at all for a scale of exactly one); a second backward through the
    same node (retain_graph) falls back to the two-input kernel.  Bits must equal the autograd gradient of the reference's expression in every case."""
    dev = torch.device("cuda")
    a = torch.rand(3, 200, 300, device=dev); b = torch.rand(3, 200, 300, device=dev)
    for w in (1.0, 0.8, 3.0):
        a1 = a.clone().requires_grad_(True); a2 = a.clone().requires_grad_(True)
        (torch.abs(a1 - b).mean() * w + 0.0).backward()
        l = l1_loss(a2, b)
        (l * w + 0.0).backward(retain_graph=True)
        assert torch.allclose(a2.grad, a1.grad, rtol=1e-6, atol=0.0), w
        g1 = a2.grad.clone(); a2.grad = None
        (l * w).backward()                                # the node's stored gradient is spent: recomputed from x and y
        assert torch.equal(a2.grad, g1), w
