"""vegs_amd -- MI355X-native differentiable Gaussian-splatting rasterizer for VEGS.

The product is libvegsrast.so (hand-written HIP for gfx950, C ABI in include/vegs_rast.h);
this package is the host-side mirror of the reference's operator interface
(`diff_gaussian_rasterization`), plus the view-sharded multi-GPU helper and the synthetic
scene generator used by tests and bench.py.
"""
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
