"""Drop-in for the reference's ``diff_plane_rasterization`` package.

``gaussian_renderer/__init__.py:5-6`` of the reference does
``from diff_plane_rasterization import GaussianRasterizationSettings, GaussianRasterizer``; putting this
repository on ``sys.path`` makes those imports resolve to the MI355X implementation unchanged.
"""
from ibgs_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians,  # noqa: F401
                                 _RasterizeGaussians, _C, cpu_deep_copy_tuple)
