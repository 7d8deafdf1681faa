"""Per-step densification bookkeeping on the MI355X: the consumer of the rasterizer's `means2D`
gradient holder and `radii` (main.py:276-281, `GaussianModel.add_densification_stats`
gs_renderer.py:625-627) as ONE launch with no host synchronisation -- in torch each of the three
boolean-mask updates is a `nonzero()` (device->host sync) plus gathers and scatters."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


@torch.no_grad()
def add_densification_stats(viewspace_grad: torch.Tensor, radii: torch.Tensor, xyz_gradient_accum: torch.Tensor,
                            denom: torch.Tensor, max_radii2D: torch.Tensor) -> None:
    """In place, for `visibility_filter = radii > 0` (gs_renderer.py:813):

        max_radii2D[vis] = max(max_radii2D[vis], radii[vis])                    main.py:280
        xyz_gradient_accum[vis] += norm(viewspace_grad[vis, :2], dim=-1)        gs_renderer.py:626
        denom[vis] += 1                                                         gs_renderer.py:627

    `viewspace_grad` is `viewspace_points.grad` [N,3]; the accumulators are the model's own tensors
    ([N,1] / [N] float32, contiguous, on the GPU) and are updated in place."""
    dev = viewspace_grad.device
    if dev.type != "cuda":
        raise RuntimeError("add_densification_stats runs on the GPU only (no CPU fallback); got " + str(dev))
    N = int(radii.shape[0])
    if tuple(viewspace_grad.shape) != (N, 3):
        raise RuntimeError("viewspace_grad must have dimensions (num_points, 3)")
    for name, t in (("xyz_gradient_accum", xyz_gradient_accum), ("denom", denom), ("max_radii2D", max_radii2D)):
        if t.device != dev or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != N:
            raise RuntimeError(f"{name} must be a contiguous float32 tensor with num_points elements on {dev}")
    g = viewspace_grad.detach().to(torch.float32).contiguous()
    r = radii.detach().to(torch.int32).contiguous()
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = _lib.load().gsr_densify_stats(N, _lib.ptr(g), _lib.ptr(r), _lib.ptr(xyz_gradient_accum), _lib.ptr(denom),
                                           _lib.ptr(max_radii2D), stream)
    _lib.check(rc, "gsr_densify_stats")
