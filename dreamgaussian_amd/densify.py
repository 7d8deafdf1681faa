"""Per-step densification bookkeeping on the MI355X: the consumer of the rasterizer's `means2D`
gradient holder and `radii` (main.py:276-281, `GaussianModel.add_densification_stats`
gs_renderer.py:625-627) as ONE launch with no host synchronisation -- in torch each of the three
boolean-mask updates is a `nonzero()` (device->host sync) plus gathers and scatters -- and the structural half of the
densification (gs_renderer.py:479-609): prune, clone, split and the optimiser-state surgery behind them as one mask
compaction + one gather / concatenation launch instead of one boolean-mask indexing or `torch.cat` per tensor."""
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
def prune_points(gaussians, mask: torch.Tensor, zorder: bool = False) -> None:
    """`GaussianModel.prune_points(mask)` (gs_renderer.py:496-511, with `_prune_optimizer` :479-494) on a reference
    GaussianModel instance: removes the rows where `mask` is True from the six parameters, their Adam moments and the
    three accumulators -- one mask compaction (one synchronisation) and one gather launch instead of 21 boolean-mask
    indexings. Opt-in replacement; the reference's own method keeps working unchanged.

    `zorder=True` (round 5): the surviving rows leave in the order of a 3-D Z-order curve through their positions instead of in
    index order -- the gather moves every row of every tensor anyway, so the spatial order the binning kernels like (the scatter
    writes a third of the bytes, `profiles/r04_bench_order.jsonl`) costs one sort of N keys per densification interval and no
    pass of its own: `densify_and_prune` ends in this call (gs_renderer.py:597-609). The model is the same set of Gaussians, each
    with its own moments and statistics: the reference's result up to a permutation of the rows (tests/test_optim_gpu.py)."""
    idx, _ = compact_mask(~mask)
    if zorder and int(idx.numel()) > 1:
        order = morton_order(gaussians._xyz.detach()[idx.long()])
        idx = idx[order.long()].contiguous()
    slots = _optimizer_slots(gaussians)
    aux = [gaussians.xyz_gradient_accum, gaussians.denom, gaussians.max_radii2D]
    outs = gather_rows(idx, [t for t, _, _, _ in slots] + aux)
    _rebind(gaussians, slots, outs[:len(slots)])
    gaussians.xyz_gradient_accum, gaussians.denom, gaussians.max_radii2D = outs[len(slots):]


def _optimizer_slots(gaussians):
    """(tensor, kind, group, state) of the six parameters and their Adam moments, in the optimiser's group order."""
    out = []
    for g in gaussians.optimizer.param_groups:
        p = g["params"][0]
        st = gaussians.optimizer.state.get(p, None)
        out.append((p.data, "param", g, st))
        if st is not None:
            out.append((st["exp_avg"], "exp_avg", g, st))
            out.append((st["exp_avg_sq"], "exp_avg_sq", g, st))
    return out


def _rebind(gaussians, slots, outs):
    new_params = {}
    for (_, kind, g, st), t in zip(slots, outs):
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


@torch.no_grad()
def densification_postfix(gaussians, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling, new_rotation,
                          zorder: bool = False) -> None:
    """`GaussianModel.densification_postfix` with `cat_tensors_to_optimizer` (gs_renderer.py:513-552) on a reference
    GaussianModel instance: the six parameters extended by the new rows, their Adam moments by zeros, the three
    accumulators reset to zeros of the new length -- ONE launch (`gsr_concat_rows`) instead of 18 `torch.cat` + 12 `zeros_like`.

    `zorder=True` (round 6): the extended model is left along the 3-D Z-order curve of its positions (`reorder_gaussians`: one sort
    of N keys and one gather launch behind the concatenation), so that a trainer that keeps its Gaussians on the curve
    (`prune_points(zorder=True)`) does not leave it when it clones or splits. The same set of Gaussians, each with its own moments:
    the reference's result up to a permutation of the rows (tests/test_optim_gpu.py)."""
    new = {"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest, "opacity": new_opacities,
           "scaling": new_scaling, "rotation": new_rotation}
    slots = _optimizer_slots(gaussians)
    dev = gaussians._xyz.device
    n_old, n_new = int(gaussians._xyz.shape[0]), int(new_xyz.shape[0])
    outs, keep, arr = [], [], []
    for t, kind, g, _ in slots:
        ext = new[g["name"]].detach().to(torch.float32).contiguous() if kind == "param" else None
        if kind == "param" and (tuple(ext.shape[1:]) != tuple(t.shape[1:]) or int(ext.shape[0]) != n_new):
            # (rows as well as trailing dimensions: gsr_concat_rows reads n_new rows of every tensor)
            raise RuntimeError(f"new rows of '{g['name']}' have shape {tuple(ext.shape)}, expected ({n_new}, {', '.join(str(int(d)) for d in t.shape[1:])})")
        o = torch.empty((n_old + n_new,) + tuple(t.shape[1:]), dtype=torch.float32, device=dev)
        outs.append(o)
        width = 1
        for d in t.shape[1:]:
            width *= int(d)
        if o.numel() > 0:
            src = t.detach().contiguous()
            keep += [src, ext]
            arr.append(_lib.GsrConcatTensor(src.data_ptr(), None if ext is None else ext.data_ptr(), o.data_ptr(), width, 0))
    acc = [torch.empty((n_old + n_new, 1), dtype=torch.float32, device=dev), torch.empty((n_old + n_new, 1), dtype=torch.float32, device=dev),
           torch.empty((n_old + n_new,), dtype=torch.float32, device=dev)]
    for o in acc:
        arr.append(_lib.GsrConcatTensor(None, None, o.data_ptr(), 1, 0))
    if n_old + n_new > 0:
        lib = _lib.load()
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            for i0 in range(0, len(arr), 24):
                chunk = (_lib.GsrConcatTensor * len(arr[i0:i0 + 24]))(*arr[i0:i0 + 24])
                _lib.check(lib.gsr_concat_rows(len(chunk), chunk, n_old, n_new, stream), "gsr_concat_rows")
    _rebind(gaussians, slots, outs)
    gaussians.xyz_gradient_accum, gaussians.denom, gaussians.max_radii2D = acc
    if zorder and n_old + n_new > 1:
        reorder_gaussians(gaussians)


@torch.no_grad()
def densify_and_clone(gaussians, grads: torch.Tensor, grad_threshold: float, scene_extent: float, zorder: bool = False) -> None:
    """`GaussianModel.densify_and_clone` (gs_renderer.py:582-595): the selection as the reference writes it, then ONE mask
    compaction, one gather of the six parameters' selected rows and the one-launch postfix (`zorder`: see there)."""
    sel = torch.logical_and(torch.norm(grads, dim=-1) >= grad_threshold,
                            torch.max(gaussians.get_scaling, dim=1).values <= gaussians.percent_dense * scene_extent)
    idx, count = compact_mask(sel)
    rows = gather_rows(idx, [gaussians._xyz, gaussians._features_dc, gaussians._features_rest, gaussians._opacity,
                             gaussians._scaling, gaussians._rotation])
    densification_postfix(gaussians, *rows, zorder=zorder)


@torch.no_grad()
def quaternion_to_matrix(r: torch.Tensor) -> torch.Tensor:
    """[n,4] quaternions (r, x, y, z), normalised here, -> [n,3,3]: the default `build_rotation` of densify_and_split (the
    convention of gs_renderer.py:85-107; the reference's own function can be passed instead)."""
    q = r / torch.norm(r, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.reshape(-1, 3, 3)


@torch.no_grad()
def densify_and_split(gaussians, grads: torch.Tensor, grad_threshold: float, scene_extent: float, N: int = 2, build_rotation=None,
                      zorder: bool = False) -> None:
    """`GaussianModel.densify_and_split` (gs_renderer.py:554-580): selection and the random offsets as the reference computes
    them (same torch calls, same RNG stream), the rows through one compaction + one gather, then the one-launch postfix and
    the one-gather prune of the split originals. `build_rotation` = gs_renderer.build_rotation (quaternion -> matrix; default: the
    same convention, `quaternion_to_matrix`). `zorder=True`: the prune that ends the call leaves the rows along the Z-order curve
    (its gather moves every row anyway: no pass of its own)."""
    n_init = int(gaussians.get_xyz.shape[0])
    dev = gaussians._xyz.device
    padded = torch.zeros(n_init, device=dev)
    padded[:grads.shape[0]] = grads.squeeze()
    sel = torch.logical_and(padded >= grad_threshold,
                            torch.max(gaussians.get_scaling, dim=1).values > gaussians.percent_dense * scene_extent)
    idx, count = compact_mask(sel)
    xyz, f_dc, f_rest, opac, scaling_raw, rot = gather_rows(idx, [gaussians._xyz, gaussians._features_dc, gaussians._features_rest,
                                                                    gaussians._opacity, gaussians._scaling, gaussians._rotation])
    scaling = gaussians.scaling_activation(scaling_raw)
    stds = scaling.repeat(N, 1)
    samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=dev), std=stds)
    rots = (build_rotation or quaternion_to_matrix)(rot).repeat(N, 1, 1)
    new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + xyz.repeat(N, 1)
    new_scaling = gaussians.scaling_inverse_activation(scaling.repeat(N, 1) / (0.8 * N))
    densification_postfix(gaussians, new_xyz, f_dc.repeat(N, 1, 1), f_rest.repeat(N, 1, 1), opac.repeat(N, 1), new_scaling, rot.repeat(N, 1))
    prune_points(gaussians, torch.cat((sel, torch.zeros(N * count, device=dev, dtype=torch.bool))), zorder=zorder)


def _spread3(v: torch.Tensor) -> torch.Tensor:
    """21-bit integers -> the same bits at every third position (int64)."""
    v = v & 0x1FFFFF
    v = (v | (v << 32)) & 0x1F00000000FFFF
    v = (v | (v << 16)) & 0x1F0000FF0000FF
    v = (v | (v << 8)) & 0x100F00F00F00F00F
    v = (v | (v << 4)) & 0x10C30C30C30C30C3
    v = (v | (v << 2)) & 0x1249249249249249
    return v


@torch.no_grad()
def morton_order(xyz: torch.Tensor) -> torch.Tensor:
    """Permutation [N] (int32) that sorts the points along a 3-D Morton (Z-order) curve of their bounding box
    (21 bits per axis, ties by index: `torch.sort(stable=True)`). Gaussians that are close in space are close on the
    screen of any camera, so after `reorder_gaussians` the 256 Gaussians of a preprocess workgroup emit into a
    handful of tiles (the binning kernels write runs instead of single 8-byte entries) and the record gathers of a
    tile's compositing hit the same cache lines. Host-side torch ops (a dozen elementwise launches and one sort):
    meant for the densification interval (main.py:283-289), not for every step."""
    p = xyz.detach().to(torch.float32)
    lo = p.min(dim=0).values
    span = (p.max(dim=0).values - lo).clamp_min(1e-20)
    q = ((p - lo) / span * 2097151.0).to(torch.int64).clamp_(0, 2097151)
    code = _spread3(q[:, 0]) | (_spread3(q[:, 1]) << 1) | (_spread3(q[:, 2]) << 2)
    return torch.sort(code, stable=True).indices.to(torch.int32)


@torch.no_grad()
def reorder_gaussians(gaussians, perm: torch.Tensor = None) -> torch.Tensor:
    """Permute the rows of a reference GaussianModel -- six parameters, their Adam moments, the three densification
    accumulators -- with ONE gather launch; `perm` defaults to `morton_order(gaussians._xyz)`. The model is the same
    set of Gaussians in another order: every per-Gaussian quantity follows its row, images change only through the
    tie-break of equal depths inside a tile (the rasterizer orders equal keys by index, like the reference's stable
    radix sort, rasterizer_impl.cu's `SortPairs`). Opt-in; returns the permutation it applied."""
    if perm is None:
        perm = morton_order(gaussians._xyz)
    perm = perm.to(torch.int32).contiguous()
    slots = _optimizer_slots(gaussians)
    aux = [gaussians.xyz_gradient_accum, gaussians.denom, gaussians.max_radii2D]
    outs = gather_rows(perm, [t for t, _, _, _ in slots] + aux)
    _rebind(gaussians, slots, outs[:len(slots)])
    gaussians.xyz_gradient_accum, gaussians.denom, gaussians.max_radii2D = outs[len(slots):]
    return perm
