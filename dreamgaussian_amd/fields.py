"""Density grid of the Gaussians on the MI355X: host-side mirror of
`GaussianModel.extract_fields` (gs_renderer.py:218-294), the first step of mesh export
(`extract_mesh`, gs_renderer.py:296-323; callers main.py:455-460 via `save_model`).

The reference loops over 16^3 blocks in Python and builds `[M,L,3]` / `[M,L,6]` temporaries per
block; here the whole grid is three HIP launches (libgsr.so: `gsr_extract_fields`). The host keeps
what the reference computes on the host -- the block geometry, from the same torch calls -- so the
block membership of every Gaussian is decided on bit-identical numbers."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def block_geometry(resolution: int, num_blocks: int, relax_ratio: float):
    """(axis [R], split_size, box_lo [nc], box_hi [nc]) exactly as gs_renderer.py:221-225, 251-262
    builds them: `linspace(-1,1,R).split(R // num_blocks)`, every chunk's [min, max] grown by
    `block_size * relax_ratio`."""
    block_size = 2 / num_blocks
    assert resolution % block_size == 0                       # the reference's own check (:223)
    split_size = resolution // num_blocks
    axis = torch.linspace(-1, 1, resolution)
    lo, hi = [], []
    for c in axis.split(split_size):
        vmin, vmax = c.amin(0), c.amax(0)
        vmin -= block_size * relax_ratio
        vmax += block_size * relax_ratio
        lo.append(vmin); hi.append(vmax)
    return axis, split_size, torch.stack(lo), torch.stack(hi)


@torch.no_grad()
def extract_fields(xyz: torch.Tensor, opacity: torch.Tensor, scaling: torch.Tensor, rotation: torch.Tensor,
                   resolution: int = 128, num_blocks: int = 16, relax_ratio: float = 1.5):
    """-> (occ [R,R,R] float32 on the GPU, center [3] tensor, scale float).

    Arguments are what the reference method reads from the model: `get_xyz` [N,3], `get_opacity`
    [N,1] and `get_scaling` [N,3] (activated), and the RAW `_rotation` [N,4] (gs_renderer.py:228-246).
    `center` and `scale` are the reference's `self.center` / `self.scale` (used by `extract_mesh`
    to map marching-cubes vertices back, gs_renderer.py:305-308)."""
    if xyz.device.type != "cuda":
        raise RuntimeError("extract_fields runs on the GPU only (no CPU fallback); got " + str(xyz.device))
    if xyz.dim() != 2 or xyz.shape[1] != 3:
        raise RuntimeError("xyz must have dimensions (num_points, 3)")
    dev = xyz.device
    N = int(xyz.shape[0])
    f = lambda t, shape: t.detach().to(device=dev, dtype=torch.float32).reshape(shape).contiguous()
    xyz_, op_, sc_, rot_ = f(xyz, (N, 3)), f(opacity, (N,)), f(scaling, (N, 3)), f(rotation, (N, 4))
    if N == 0 or not bool((op_ > 0.005).any()):
        # the reference reduces an empty tensor here (gs_renderer.py:238) and torch raises
        raise RuntimeError("extract_fields: no Gaussian passes the opacity > 0.005 pre-filter")
    axis, split_size, lo, hi = block_geometry(resolution, num_blocks, relax_ratio)
    nc = int(lo.shape[0])
    axis_d, lo_d, hi_d = axis.to(dev), lo.to(dev), hi.to(dev)
    occ = torch.empty((resolution,) * 3, dtype=torch.float32, device=dev)
    norm = torch.empty(4, dtype=torch.float32, device=dev)
    lib = _lib.load()
    tmp = _lib.Scratch(dev)
    P = _lib.ptr
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = lib.gsr_extract_fields(N, P(xyz_), P(op_), P(sc_), P(rot_), int(resolution), int(split_size), nc,
                                    P(axis_d), P(lo_d), P(hi_d), P(occ), P(norm), tmp.alloc, stream)
    tmp.release()
    _lib.check(rc, "gsr_extract_fields")
    host = norm.cpu()                                         # one sync, as `.item()` at gs_renderer.py:240
    center = norm[:3].clone()
    scale = 1.8 / host[3].item()
    return occ, center, scale
