"""Parity metrics with the reference's definitions (utils/image_utils.py:18-20, utils/loss_utils.py:24-26, 34-64);
pinned against the reference's own functions by tests/test_oracle_golden.py::test_metric_definitions."""
import numpy as np


def psnr(a, b):
    """per image: 20 log10(1 / sqrt(mse)); a, b (N, C, H, W) or (C, H, W)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if a.ndim == 3:
        a, b = a[None], b[None]
    mse = ((a - b) ** 2).reshape(a.shape[0], -1).mean(1)
    return 20.0 * np.log10(1.0 / np.sqrt(mse))


def l1(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).mean())


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def ssim(a, b, window_size=11, sigma=1.5, size_average=True):
    """utils/loss_utils.py:34-64: per-channel 11x11 Gaussian window (sigma 1.5, normalised 1-D kernel, outer product),
    zero-padded 'same' correlation, C1 = 0.01^2, C2 = 0.03^2, mean over everything (or per image).  a, b (N, C, H, W)."""
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    if a.ndim == 3:
        a, b = a[None], b[None]
    x = np.arange(window_size) - window_size // 2
    g = np.exp(-(x ** 2) / (2.0 * sigma ** 2)); g /= g.sum()
    half = window_size // 2

    def blur(im):     # separable zero-padded correlation along H then W
        p = np.pad(im, ((0, 0), (0, 0), (half, half), (half, half)))
        H, W = im.shape[-2:]
        t = sum(g[k] * p[:, :, k:k + H, :] for k in range(window_size))
        return sum(g[k] * t[:, :, :, k:k + W] for k in range(window_size))

    mu1, mu2 = blur(a), blur(b)
    s1, s2, s12 = blur(a * a) - mu1 * mu1, blur(b * b) - mu2 * mu2, blur(a * b) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))
    return float(m.mean()) if size_average else m.reshape(m.shape[0], -1).mean(1)
