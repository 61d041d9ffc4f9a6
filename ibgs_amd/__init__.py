"""ibgs_amd -- MI355X (gfx950) native differentiable plane rasterizer for IBGS.

Only the hot path of the reference (its ``diff_plane_rasterization`` CUDA extension) lives here:
``csrc/`` holds the hand-written HIP kernels and the C ABI (``include/ibgs_rast.h``), the Python
modules mirror the reference's operator interface on top of it.
"""
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians,  # noqa: F401
                         _RasterizeGaussians, _C)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
