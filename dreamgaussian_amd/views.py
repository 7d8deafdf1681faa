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


# ---------------------------------------------------------------------------------------------------------------------------------
# The gradient exchange at SH-3 sizes (round 6). `allreduce_grads` is dense: 62 MB per step at 1M Gaussians / SH 3, and a ring
# all-reduce moves 2 S (N - 1) / N bytes over every xGMI link -- ~0.7 ms at 8 GPUs, longer than the 0.55 ms render step. Two ways to
# send fewer bytes, both with the result of the replicated path up to the order of the fp32 sums:
#   * ShardedAdam: reduce-scatter of the gradient span (every rank receives the SUM of its 1/N slice: S (N - 1) / N bytes in, over the
#     N - 1 links of the mesh), each rank takes the Adam step on its slice only (1/N of the optimiser's arithmetic and moment traffic),
#     all-gather of the updated parameters (the same bytes out): 2 S / N per link with a direct algorithm instead of 2 S (N - 1) / N;
#   * allreduce_live_rows: only 27 % (dense blob) / 6 % (trained-like) of the Gaussians receive a gradient from ONE view
#     (profiles/r04_grad_stats.json); the ranks agree on the union of their live rows (N bytes of flags) and all-reduce those rows only.
# main.py:219-275, gs_renderer.py:356-374.
# ---------------------------------------------------------------------------------------------------------------------------------
def _flat_span(grads: Sequence[torch.Tensor], world: int):
    """The gradients as ONE flat buffer whose length is a multiple of `world`: in place when they tile one storage at the
    rasterizer's 64-element boundaries (rasterizer.carve_gradients: no copy), else a packed copy. Returns (flat, offsets, in_place):
    gradient i = flat[offsets[i] : offsets[i] + grads[i].numel()]."""
    same = len({(g.untyped_storage().data_ptr(), g.dtype, g.device) for g in grads}) == 1 and all(g.is_contiguous() for g in grads)
    if same and len(grads) > 1:
        order = sorted(grads, key=lambda g: g.storage_offset())
        lo, end, tight = order[0].storage_offset(), order[0].storage_offset(), True
        for g in order:
            tight = tight and 0 <= g.storage_offset() - end <= 63
            end = max(end, g.storage_offset() + g.numel())
        n = -(-(end - lo) // world) * world
        total = grads[0].untyped_storage().nbytes() // grads[0].element_size()
        if tight and lo + n <= total:                 # (the tail beyond `end` is padding or the per-view gradient: only READ by a reduce-scatter)
            flat = torch.empty(0, dtype=grads[0].dtype, device=grads[0].device).set_(grads[0].untyped_storage(), lo, (n,))
            return flat, [g.storage_offset() - lo for g in grads], True
    offs, tot = [], 0
    for g in grads:
        offs.append(tot)
        tot += g.numel()
    n = -(-tot // world) * world
    flat = torch.zeros(n, dtype=grads[0].dtype, device=grads[0].device)
    for g, o in zip(grads, offs):
        flat[o:o + g.numel()].copy_(g.reshape(-1))
    return flat, offs, False


def _adam_slices(p, g, m, v, step: int, lr: float, beta1: float, beta2: float, eps: float):
    """torch.optim.Adam's single-tensor update (amsgrad off, no weight decay) on flat slices, in torch's order of operations."""
    m.lerp_(g, 1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bias1, bias2 = 1 - beta1 ** step, 1 - beta2 ** step
    denom = (v.sqrt() / (bias2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bias1))


class ShardedAdam:
    """Data-parallel Adam step with the optimiser SHARDED behind a reduce-scatter (ZeRO-1 over the view-parallel ranks).

    `optimizer` is the model's Adam (torch.optim.Adam or dreamgaussian_amd.FusedAdam, one parameter per group as in
    GaussianModel.training_setup, gs_renderer.py:356-374); its `state` keeps full-size moment tensors, so the reference's
    optimiser-state surgery (densify / prune) keeps working -- but between two `gather_state()` calls a rank only maintains the
    moments of ITS slice: call `gather_state()` before anything reads or re-indexes them (the densification interval, a checkpoint).

        sync = ShardedAdam(gaussians.optimizer)
        loss.backward(); sync.step()          # instead of allreduce_grads(...) + optimizer.step()
    """

    def __init__(self, optimizer, group=None):
        self.opt, self.group = optimizer, group
        self._layout = None                               # (parameters, their offsets in the flat span, slice length) of the last step

    def _items(self):
        items = []
        for grp in self.opt.param_groups:
            for p in grp["params"]:
                if p.grad is not None:
                    items.append((p, grp))
        return items

    @torch.no_grad()
    def step(self):
        rank, world = _world(self.group)
        items = self._items()
        if not items:
            return
        if _single(world):
            self.opt.step()
            return
        grads = [p.grad for p, _ in items]
        flat, offs, _ = _flat_span(grads, world)
        S = flat.numel() // world
        shard = torch.empty(S, dtype=flat.dtype, device=flat.device)
        dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=self.group)
        s0, s1 = rank * S, (rank + 1) * S
        pshard = torch.zeros(S, dtype=flat.dtype, device=flat.device)
        adam_args = []
        for (p, grp), o in zip(items, offs):
            st = self.opt.state[p]
            if len(st) == 0:                              # torch's own lazy initialisation (adam.py: _init_group)
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] += 1
            a, b = max(o, s0) - o, min(o + p.numel(), s1) - o     # this rank's elements of the parameter
            if a >= b:
                continue
            adam_args.append((p.data.view(-1)[a:b], shard[o + a - s0:o + b - s0], st["exp_avg"].view(-1)[a:b], st["exp_avg_sq"].view(-1)[a:b],
                              int(st["step"].item()), float(grp["lr"]), float(grp["betas"][0]), float(grp["betas"][1]), float(grp["eps"]),
                              pshard[o + a - s0:o + b - s0]))
        self._apply(adam_args)
        for pv, _, _, _, _, _, _, _, _, out in adam_args:
            out.copy_(pv)
        full = torch.empty(S * world, dtype=flat.dtype, device=flat.device)
        dist.all_gather_into_tensor(full, pshard, group=self.group)
        for (p, _), o in zip(items, offs):
            p.data.view(-1).copy_(full[o:o + p.numel()])
        self._layout = ([p for p, _ in items], list(offs), S)

    def _apply(self, adam_args):
        """The Adam update of this rank's slices: one launch of libgsr's fused kernel per <= 8 slices on the GPU (pointer offsets into
        the full tensors), torch's elementwise ops elsewhere (the gloo tests) -- the same arithmetic in the same order."""
        if adam_args and adam_args[0][0].device.type == "cuda":
            import ctypes as C
            from . import _lib
            lib = _lib.load()
            by_key = {}
            for pv, gv, mv, vv, step, lr, b1, b2, eps, _ in adam_args:
                by_key.setdefault((pv.device, step, b1, b2, eps), []).append((pv, gv.contiguous(), mv, vv, lr))
            for (dev, step, b1, b2, eps), lst in by_key.items():
                with torch.cuda.device(dev):
                    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                    for i0 in range(0, len(lst), 8):
                        chunk = lst[i0:i0 + 8]
                        arr = (_lib.GsrAdamTensor * len(chunk))()
                        for j, (pv, gv, mv, vv, lr) in enumerate(chunk):
                            arr[j] = _lib.GsrAdamTensor(pv.data_ptr(), gv.data_ptr(), mv.data_ptr(), vv.data_ptr(), pv.numel(), lr)
                        _lib.check(lib.gsr_adam_step(len(chunk), arr, step, b1, b2, eps, stream), "gsr_adam_step")
            return
        for pv, gv, mv, vv, step, lr, b1, b2, eps, _ in adam_args:
            _adam_slices(pv, gv, mv, vv, step, lr, b1, b2, eps)

    @torch.no_grad()
    def gather_state(self):
        """Make every rank's moment tensors whole again (each rank has only kept ITS slice current since the last call): two
        all-gathers of S / N elements per rank. Call it BEFORE densification / pruning re-index the rows, or before a checkpoint."""
        rank, world = _world(self.group)
        if _single(world) or self._layout is None:
            return
        params, offs, S = self._layout
        s0 = rank * S
        for key in ("exp_avg", "exp_avg_sq"):
            mine = torch.zeros(S, dtype=torch.float32, device=params[0].device)
            for p, o in zip(params, offs):
                a, b = max(o, s0) - o, min(o + p.numel(), s0 + S) - o
                if a < b:
                    mine[o + a - s0:o + b - s0].copy_(self.opt.state[p][key].view(-1)[a:b])
            full = torch.empty(S * world, dtype=torch.float32, device=mine.device)
            dist.all_gather_into_tensor(full, mine, group=self.group)
            for p, o in zip(params, offs):
                self.opt.state[p][key].view(-1).copy_(full[o:o + p.numel()])


@torch.no_grad()
def allreduce_live_rows(params: Sequence[torch.Tensor], group=None, probe: Optional[Sequence[int]] = None):
    """Sum `.grad` of the replicated Gaussian parameters over the ranks, exchanging only the ROWS (Gaussians) that carry a gradient
    on at least one rank. Every parameter must be [N, ...] with the same N. A row is live on a rank when any element of its
    gradients in `probe` (indices into `params`; default: all of them) is non-zero -- for the rasterizer's gradients the small ones
    (positions, opacity, scales, rotations, SH band 0) decide exactly: a Gaussian no pixel gradient reached has exact zeros
    everywhere. Steps: N bytes of flags all-reduced (MAX) -> the union's indices -> the rows gathered into one
    [U, sum of widths] buffer (gather_rows on the GPU) -> ONE all-reduce of U rows -> written back. Returns (U, N)."""
    rank, world = _world(group)
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return 0, 0
    N = int(ps[0].shape[0])
    if any(int(p.shape[0]) != N for p in ps):
        raise RuntimeError("allreduce_live_rows needs parameters with the same number of rows")
    if _single(world):
        return N, N
    grads = [p.grad.reshape(N, -1) for p in ps]
    use = range(len(ps)) if probe is None else probe
    live = torch.zeros(N, dtype=torch.bool, device=grads[0].device)
    for i in use:
        live |= (grads[i] != 0).any(dim=1)
    flags = live.to(torch.uint8)
    dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=group)
    idx = flags.nonzero().squeeze(1)                      # (one host synchronisation: the size of the exchange)
    U = int(idx.numel())
    if U == 0:
        return 0, N
    widths = [g.shape[1] for g in grads]
    if grads[0].device.type == "cuda" and all(g.dtype is torch.float32 for g in grads):
        from .densify import gather_rows
        rows = gather_rows(idx.to(torch.int32), [g.contiguous() for g in grads])       # one launch for all tensors
        buf = torch.cat(rows, dim=1)
    else:
        buf = torch.cat([g[idx] for g in grads], dim=1)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    o = 0
    for p, w in zip(ps, widths):
        p.grad.reshape(N, -1).index_copy_(0, idx, buf[:, o:o + w])
        o += w
    return U, N
