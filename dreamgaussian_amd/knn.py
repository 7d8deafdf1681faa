"""distCUDA2 on the MI355X: host-side mirror of `simple_knn._C.distCUDA2`
(simple-knn/ext.cpp:15-17 -> spatial.cu:15-26 -> simple_knn.cu:185-221; caller
gs_renderer.py:341). points [P,3] float32 on the GPU -> [P] float32: mean of the squared
distances to the 3 nearest neighbours. Runs in libgsr.so's HIP kernels on torch's current
stream, with no host synchronisation."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if points.device.type != "cuda":
        raise RuntimeError("distCUDA2 runs on the GPU only (no CPU fallback); got " + str(points.device))
    if points.dim() != 2 or points.shape[1] != 3:
        raise RuntimeError("points must have dimensions (num_points, 3)")
    lib = _lib.load()
    dev = points.device
    pts = points.detach().to(torch.float32).contiguous()
    P = int(pts.shape[0])
    out = torch.zeros(P, dtype=torch.float32, device=dev)
    tmp = _lib.Scratch(dev)
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = lib.gsr_dist2(P, _lib.ptr(pts), _lib.ptr(out), tmp.alloc, stream)
    tmp.release()
    _lib.check(rc, "gsr_dist2")
    return out
