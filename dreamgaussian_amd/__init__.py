"""MI355X-native differentiable Gaussian-splat rasterizer for DreamGaussian's stage-1 path.

Public surface (mirrors what gs_renderer.py imports, gs_renderer.py:10-14):
    GaussianRasterizationSettings, GaussianRasterizer   (package `diff_gaussian_rasterization`)
    distCUDA2                                           (package `simple_knn._C`)
Beside the drop-in names: rasterize_views (several cameras in flight), rasterize_gaussians_raw
(fused activations), rasterize_gaussians_split (fused activations + features_dc / features_rest read in place),
extract_fields (GaussianModel.extract_fields, gs_renderer.py:218-294), FusedAdam (torch.optim.Adam with a
one-launch step), add_densification_stats / compact_mask / gather_rows / prune_points / densification_postfix /
densify_and_clone / densify_and_split / morton_order / reorder_gaussians (densification without per-tensor nonzero() synchronisations, boolean-mask
indexings and `torch.cat`s).
"""
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,
                         rasterize_gaussians, rasterize_gaussians_raw, rasterize_gaussians_split, last_stats,
                         set_async_forward, set_deterministic, use_cpp_binding, binding_loaded, peek_stats)
from .knn import distCUDA2
from .batched import rasterize_views
from .fields import extract_fields
from .densify import (add_densification_stats, compact_mask, gather_rows, prune_points, densification_postfix,
                      densify_and_clone, densify_and_split, morton_order, reorder_gaussians)
from .optim import FusedAdam

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians", "rasterize_gaussians_raw", "rasterize_gaussians_split",
           "last_stats", "peek_stats", "set_async_forward", "set_deterministic", "use_cpp_binding", "binding_loaded", "distCUDA2", "rasterize_views", "extract_fields", "add_densification_stats", "compact_mask", "gather_rows", "prune_points",
           "densification_postfix", "densify_and_clone", "densify_and_split", "morton_order", "reorder_gaussians", "FusedAdam"]
