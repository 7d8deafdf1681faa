"""`from simple_knn._C import distCUDA2` (gs_renderer.py:14) -> dreamgaussian_amd.knn."""
from dreamgaussian_amd.knn import distCUDA2

__all__ = ["distCUDA2"]
