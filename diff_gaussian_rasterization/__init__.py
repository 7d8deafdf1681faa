"""Drop-in for the external `diff_gaussian_rasterization` package the reference imports at
gs_renderer.py:10-13 -- put this repository on sys.path and `gs_renderer.py` runs unmodified
on PyTorch-ROCm. Everything is implemented in dreamgaussian_amd (hand-written HIP, gfx950)."""
from dreamgaussian_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,
                                          rasterize_gaussians)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
