"""Per-step densification bookkeeping on the MI355X: the consumer of the rasterizer's `means2D`
gradient holder and `radii` (main.py:276-281, `GaussianModel.add_densification_stats`
gs_renderer.py:625-627) as ONE launch with no host synchronisation -- in torch each of the three
boolean-mask updates is a `nonzero()` (device->host sync) plus gathers and scatters."""
from __future__ import annotations

import ctypes as C

import torch

from typing import List, Sequence

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


@torch.no_grad()
def compact_mask(mask: torch.Tensor):
    """Stable compaction of a boolean mask [N] on the GPU: returns (idx uint32-as-int32 view [count], count).
    `idx[j]` is the index of the j-th True element -- what `mask.nonzero()` gives, with ONE host synchronisation
    (the count, needed to size tensors) however many tensors are gathered with it afterwards."""
    dev = mask.device
    if dev.type != "cuda":
        raise RuntimeError("compact_mask runs on the GPU only (no CPU fallback); got " + str(dev))
    N = int(mask.numel())
    m8 = mask.reshape(-1).to(torch.uint8).contiguous()
    idx = torch.empty(max(N, 1), dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    tmp = _lib.Scratch(dev)
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = _lib.load().gsr_mask_compact(N, _lib.ptr(m8), _lib.ptr(idx), _lib.ptr(cnt), tmp.alloc, stream)
    tmp.release()
    _lib.check(rc, "gsr_mask_compact")
    count = int(cnt.item())                            # the one synchronisation
    return idx[:count], count


@torch.no_grad()
def gather_rows(idx: torch.Tensor, tensors: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """[t[idx] for t in tensors] (first-dimension gather) for up to 24 float32 tensors per launch: the six parameters,
    their Adam moments and the densification accumulators of prune_points / densification_postfix
    (gs_renderer.py:479-545) move with one kernel instead of one index_select each."""
    if not tensors:
        return []
    dev = idx.device
    rows = int(idx.numel())
    outs, srcs, dsts = [], [], []
    for t in tensors:
        if t.device != dev or t.dtype is not torch.float32:
            raise RuntimeError("gather_rows takes float32 tensors on the index's device")
        o = torch.empty((rows,) + tuple(t.shape[1:]), dtype=torch.float32, device=dev)
        outs.append(o)
        if o.numel() > 0:                              # e.g. _features_rest is [N,0,3] at sh_degree 0: nothing to move
            srcs.append(t.detach().contiguous())
            dsts.append(o)
    if srcs:
        lib = _lib.load()
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            for i0 in range(0, len(srcs), 24):
                chunk = list(zip(srcs[i0:i0 + 24], dsts[i0:i0 + 24]))
                arr = (_lib.GsrGatherTensor * len(chunk))()
                for j, (a, b) in enumerate(chunk):
                    width = 1
                    for d in a.shape[1:]:
                        width *= int(d)
                    arr[j] = _lib.GsrGatherTensor(a.data_ptr(), b.data_ptr(), max(width, 1), 0)
                _lib.check(lib.gsr_gather_rows(len(chunk), arr, rows, _lib.ptr(idx), stream), "gsr_gather_rows")
    return outs


@torch.no_grad()
def prune_points(gaussians, mask: torch.Tensor) -> None:
    """`GaussianModel.prune_points(mask)` (gs_renderer.py:496-511, with `_prune_optimizer` :479-494) on a reference
    GaussianModel instance: removes the rows where `mask` is True from the six parameters, their Adam moments and the
    three accumulators -- one mask compaction (one synchronisation) and one gather launch instead of 21 boolean-mask
    indexings. Opt-in replacement; the reference's own method keeps working unchanged."""
    idx, _ = compact_mask(~mask)
    groups = gaussians.optimizer.param_groups
    tensors, slots = [], []
    for g in groups:
        p = g["params"][0]
        st = gaussians.optimizer.state.get(p, None)
        tensors.append(p.data)
        slots.append((g, "param", st))
        if st is not None:
            tensors.append(st["exp_avg"]); slots.append((g, "exp_avg", st))
            tensors.append(st["exp_avg_sq"]); slots.append((g, "exp_avg_sq", st))
    aux = [gaussians.xyz_gradient_accum, gaussians.denom, gaussians.max_radii2D]
    outs = gather_rows(idx, tensors + aux)
    new_params = {}
    for (g, kind, st), t in zip(slots, outs[:len(slots)]):
        if kind == "param":
            old = g["params"][0]
            newp = torch.nn.Parameter(t.requires_grad_(True))
            if st is not None:
                del gaussians.optimizer.state[old]
                gaussians.optimizer.state[newp] = st
            g["params"][0] = newp
            new_params[g["name"]] = newp
        else:
            st[kind] = t
    gaussians._xyz, gaussians._features_dc, gaussians._features_rest = new_params["xyz"], new_params["f_dc"], new_params["f_rest"]
    gaussians._opacity, gaussians._scaling, gaussians._rotation = new_params["opacity"], new_params["scaling"], new_params["rotation"]
    gaussians.xyz_gradient_accum, gaussians.denom, gaussians.max_radii2D = outs[len(slots)], outs[len(slots) + 1], outs[len(slots) + 2]
