"""Parity metrics with the reference's definitions (utils/image_utils.py:18-20, utils/loss_utils.py:24-26);
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
