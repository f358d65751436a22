"""Drop-in package name: the reference imports
`from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`
(gaussian_renderer/__init__.py:14, train.py:35, utils/norminit_utils.py:7).  Everything is
implemented in vegs_amd (HIP kernels behind the C ABI of include/vegs_rast.h)."""
from vegs_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                 _RasterizeGaussians, rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
