"""View-parallel rendering over the GPUs of one node.

The B cameras of one DreamGaussian SDS step are independent given the Gaussians: the
reference renders them in a serial Python loop and `torch.cat`s the images
(main.py:219-255). Here the Gaussians are replicated, rank r renders views r, r+world, ...
and the images meet on one rank through an RCCL gather over xGMI (direct peer->root
transfers: each peer owns a link to the root, so the gather is one hop and per-link bound);
per-Gaussian gradients are summed with one all-reduce per attribute bucket.

Where the loss lives decides what is exchanged (bench.py --step sds --sds-mode ...):
  * "gather": ONE rank evaluates the image-space loss for all views (a guidance model that exists once): gather of the images,
    scatter of dL/dimage, all-reduce of the parameter gradients -- three collectives per step;
  * "local": the loss is a sum over the views and every rank can evaluate its own views' terms (DreamGaussian's known-view MSE,
    main.py:200-216, and an SDS guidance replicated per rank, the usual data-parallel layout): each rank differentiates its own
    images, NO image ever crosses a link, the all-reduce of the parameter gradients is the only collective (`allreduce_grads`,
    `async_op=True`: it travels while the host prepares the optimiser step).

One process per GPU, `torch.distributed` (backend "nccl" = RCCL on ROCm; "gloo" in the CPU
tests). No rendering arithmetic lives here.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist


_force = False
gather_fallbacks = 0        # calls of gather_views_async that had to replace `gather` by `all_gather` (never under "nccl")


def force_collectives(on: bool = True):
    """Run every collective even in a group of ONE rank (default: a single rank short-circuits them). With a 1-rank
    "nccl" group this pushes gather / scatter / all-reduce of device tensors through RCCL on a single-GPU box: the
    same calls, arguments and buffer handling as on 8 GPUs (tests/test_views_gpu.py, bench.py --force-collectives)."""
    global _force
    _force = bool(on)


def _world(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def _single(world: int) -> bool:
    """True when the collectives may be skipped: one rank and nobody asked for them."""
    return world == 1 and not (_force and dist.is_available() and dist.is_initialized())


def shard_views(views: Sequence, group=None) -> List:
    """Round-robin ownership: rank r gets views[r::world] (view i lives on rank i % world)."""
    rank, world = _world(group)
    return list(views[rank::world])


def owner_of(view_index: int, group=None) -> int:
    return view_index % _world(group)[1]


def make_gather_buffer(world: int, channels: int, H: int, W: int, device) -> torch.Tensor:
    """[world, channels, H, W] staging buffer; only the destination rank reads it."""
    return torch.empty(world, channels, H, W, dtype=torch.float32, device=device)


def gather_views_async(color: torch.Tensor, depth: torch.Tensor, alpha: torch.Tensor,
                       buf: torch.Tensor, dst: int = 0, group=None):
    """Start the gather of this rank's (color[3,H,W], depth[1,H,W], alpha[1,H,W]) into
    buf[rank] on `dst`; returns the work handle (wait() before reading buf). The images are
    detached: gradients flow back through `scatter_view_grads`."""
    rank, world = _world(group)
    local = torch.cat([color.detach(), depth.detach(), alpha.detach()], dim=0).contiguous()
    if _single(world):
        buf[0].copy_(local)
        return None
    glist = [buf[i] for i in range(world)] if rank == dst else None
    try:
        return dist.gather(local, glist, dst=dst, group=group, async_op=True)
    except (RuntimeError, NotImplementedError) as e:
        # A backend without gather: every rank receives every view -- world x the bytes. Never silently: counted, logged once per
        # process, and refused outright under RCCL ("nccl" HAS gather: an exception there is a real failure, not a missing feature).
        if dist.get_backend(group) == "nccl":
            raise
        global gather_fallbacks
        gather_fallbacks += 1
        if gather_fallbacks == 1:
            import warnings
            warnings.warn(f"dreamgaussian_amd.views: dist.gather failed on backend {dist.get_backend(group)!r} ({e}); falling back to "
                          f"all_gather_into_tensor -- {world} x the bytes per step (views.gather_fallbacks counts the calls)")
        full = buf if rank == dst else torch.empty_like(buf)
        return dist.all_gather_into_tensor(full.view(-1), local.view(-1), group=group, async_op=True)


def views_on_rank(num_views: int, rank: int, world: int) -> int:
    """How many of `num_views` round-robin views rank `rank` owns."""
    return len(range(rank, num_views, world))


def gather_images(local: torch.Tensor, dst: Optional[int] = 0, group=None,
                  num_views: Optional[int] = None) -> Optional[torch.Tensor]:
    """local [b,C,H,W] (this rank's views) -> [num_views,C,H,W] in VIEW order (view i = rank i % world,
    slot i // world) on `dst` (None = every rank). `num_views` = total number of views; when it is not a
    multiple of the world size the ranks hold unequal counts (rank r: views r, r+world, ...) and the
    shorter ranks are padded for the collective. Default: b * world (equal counts)."""
    rank, world = _world(group)
    if _single(world):
        return local
    b = int(local.shape[0])
    if num_views is None:
        num_views = b * world
    bmax = (num_views + world - 1) // world
    if b != views_on_rank(num_views, rank, world):
        raise RuntimeError(f"rank {rank} holds {b} views, expected {views_on_rank(num_views, rank, world)} of {num_views}")
    local = local.contiguous()
    if b < bmax:
        local = torch.cat([local, local.new_zeros((bmax - b,) + tuple(local.shape[1:]))], 0)
    if dst is None:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.all_gather([out[i] for i in range(world)], local, group=group)
    else:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device) \
            if rank == dst else None
        dist.gather(local, [out[i] for i in range(world)] if rank == dst else None, dst=dst, group=group)
        if rank != dst:
            return None
    # [world, bmax, ...] -> view order: index = slot * world + rank; padded slots are exactly the indices >= num_views
    return out.transpose(0, 1).reshape((-1,) + tuple(local.shape[1:]))[:num_views]


def scatter_view_grads(grad_all: Optional[torch.Tensor], like: torch.Tensor, src: int = 0,
                       group=None, num_views: Optional[int] = None) -> torch.Tensor:
    """Inverse of gather_images for the backward: `grad_all` [num_views,C,H,W] in view order on
    `src` -> this rank's [b,C,H,W] slice (`like` gives its shape)."""
    rank, world = _world(group)
    if _single(world):
        return grad_all
    b = int(like.shape[0])
    if num_views is None:
        num_views = b * world
    bmax = (num_views + world - 1) // world
    out = like.new_empty((bmax,) + tuple(like.shape[1:]))
    if rank == src:
        g = grad_all
        if num_views < bmax * world:
            g = torch.cat([g, g.new_zeros((bmax * world - num_views,) + tuple(g.shape[1:]))], 0)
        g = g.reshape((bmax, world) + tuple(like.shape[1:])).transpose(0, 1).contiguous()
        dist.scatter(out, [g[i] for i in range(world)], src=src, group=group)
    else:
        dist.scatter(out, None, src=src, group=group)
    return out[:b]


class GradSync:
    """Handle of `allreduce_grads(..., async_op=True)`: the collectives are in flight on the backend's stream; `wait()` makes the
    CURRENT stream wait for them (no host block under "nccl") and copies bucketed gradients back. Call it before anything reads or
    frees the gradients -- the optimiser step, or `p.grad = None` at the top of the next iteration."""

    def __init__(self):
        self._works = []
        self._copies = []          # (flat bucket, its gradients): copied back in wait()

    def wait(self):
        for w in self._works:
            if w is not None:
                w.wait()
        for flat, bucket in self._copies:
            off = 0
            for g in bucket:
                n = g.numel()
                g.copy_(flat[off:off + n].view_as(g))
                off += n
        self._works, self._copies = [], []


def allreduce_grads(params: Iterable[torch.Tensor], group=None, bucket_bytes: int = 64 << 20, async_op: bool = False):
    """Sum `.grad` of the replicated Gaussian parameters over the ranks. `async_op=True`: returns a `GradSync` at once (None when
    there is nothing to do): the caller's next host work -- the optimiser's Python, the next iteration's camera set-up -- runs
    while the ranks exchange; `wait()` before the gradients are read or released.

    The rasterizer's backward carves every gradient out of ONE allocation (rasterizer.py: the parameter gradients first,
    the per-view means2D gradient last) and autograd keeps those views as `.grad`: gradients that tile one storage are
    reduced IN PLACE with a single all-reduce over their span -- no `cat`, no copy back (62 MB at 1M Gaussians / SH 3:
    two passes over HBM and a launch per tensor saved). Anything else (gradients that went through torch ops, e.g. the
    activations of gs_renderer.py:196-216) goes through flat buckets of ~bucket_bytes: few, large collectives -- xGMI
    rings are per-link bound, launch latency dominates small ones."""
    rank, world = _world(group)
    if _single(world):
        return None
    sync = GradSync()
    grads = [p.grad for p in params if p.grad is not None]
    by_storage = {}
    for g in grads:
        by_storage.setdefault((g.untyped_storage().data_ptr(), g.dtype, g.device), []).append(g)
    rest = []
    for (_, dtype, dev), gs in by_storage.items():
        if len(gs) > 1 and all(g.is_contiguous() for g in gs):
            # The span [lo, hi) is reduced as ONE buffer, gaps included: safe only if every gap is padding nobody reads -- the
            # rasterizer carves its gradients at 64-element (256-byte) boundaries, so a gap of up to 63 elements is padding; a
            # larger one may be a LIVE gradient that was not passed here (e.g. d_opacity between d_means3D and d_sh when a caller
            # reduces parameter group by parameter group) and would be reduced twice. Then: the bucket path.
            order = sorted(gs, key=lambda g: g.storage_offset())
            lo, hi = order[0].storage_offset(), max(g.storage_offset() + g.numel() for g in gs)
            end, tight = lo, True
            for g in order:
                tight = tight and 0 <= g.storage_offset() - end <= 63
                end = max(end, g.storage_offset() + g.numel())
            if tight:
                span = torch.empty(0, dtype=dtype, device=dev).set_(gs[0].untyped_storage(), lo, (hi - lo,))
                sync._works.append(dist.all_reduce(span, op=dist.ReduceOp.SUM, group=group, async_op=True))
                continue
        rest.extend(gs)
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        if len(bucket) == 1 and bucket[0].is_contiguous():
            sync._works.append(dist.all_reduce(bucket[0], op=dist.ReduceOp.SUM, group=group, async_op=True))
        else:
            flat = torch.cat([g.reshape(-1) for g in bucket])
            sync._works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True))
            sync._copies.append((flat, bucket))
        bucket, size = [], 0

    for g in rest:
        bucket.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            flush()
    flush()
    if async_op:
        return sync
    sync.wait()
    return None
