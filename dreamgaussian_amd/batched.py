"""Several views of the same Gaussians on ONE GPU.

The reference renders the cameras of one optimisation step in a serial Python loop
(main.py:219-255: `for _ in range(batch_size): out = renderer.render(cam) ...; torch.cat(images)`),
paying one host round trip (the instance count) and one under-filled set of launches per view --
at 512x512 a view has 1024 tiles, a third of them non-empty, on a 256-CU chip.

`rasterize_views` renders B cameras with ONE launch chain (`gsr_forward_views` / `gsr_backward_views`,
include/gsr.h): the per-Gaussian kernels run with grid.y = view, binning / sort / compositing run over all
B * tiles tiles at once (heaviest tile first across views), the host waits once per batch, and the backward adds
the views' parameter gradients into one [N,...] tensor in the order autograd would (no [B,N,...] + sum).

Same arithmetic, same kernels as `GaussianRasterizer`: images, depth, alpha, radii and the per-view means2D
gradients are bit-identical to B single-view calls; the summed parameter gradients equal the serial loop's up to
the order of the fp32 atomics inside each view.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import torch

from . import _lib
from .rasterizer import (GaussianRasterizationSettings, _f32c, _require_gpu, _view_struct)

class _RasterizeViews(torch.autograd.Function):
    """Up to GSR_MAX_VIEWS cameras through gsr_forward_views / gsr_backward_views."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, settings, grad_mode=True):
        _require_gpu(means3D)
        lib = _lib.load()
        dev = means3D.device
        B = len(settings)
        H, W = int(settings[0].image_height), int(settings[0].image_width)
        for rs in settings:
            if (int(rs.image_height), int(rs.image_width)) != (H, W):
                raise RuntimeError("rasterize_views: all views must share one image size")
        N = int(means3D.shape[0])
        if tuple(means2D.shape) != (B, N, 3):
            raise RuntimeError("means2D must have dimensions (num_views, num_points, 3)")
        m3, shc, col = _f32c(means3D, dev), _f32c(sh, dev), _f32c(colors_precomp, dev)
        op, sc, rot, cov = _f32c(opacities, dev), _f32c(scales, dev), _f32c(rotations, dev), _f32c(cov3Ds_precomp, dev)
        K = int(shc.shape[1]) if shc is not None else 0
        if (sh is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3Ds_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3Ds_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        color = torch.empty(B, 3, H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev)
        alpha = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev)
        radii = torch.empty(B, N, dtype=torch.int32, device=dev)
        views = (_lib.GsrView * B)()
        keeps = []
        for v in range(B):
            views[v], keep = _view_struct(settings[v], dev, no_backward=not (grad_mode and any(ctx.needs_input_grad)))   # (grad_mode: rasterizer.py)
            keeps.append(keep)
        geom, binb, img, st = _lib.Scratch(dev), _lib.Scratch(dev), _lib.Scratch(dev), _lib.GsrStats()
        P = _lib.ptr
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            rc = lib.gsr_forward_views(views, B, N, K, P(m3), P(shc), P(col), P(op), P(sc), P(rot), P(cov),
                                       P(color), P(depth), P(alpha), P(radii), geom.alloc, binb.alloc, img.alloc,
                                       C.byref(st), stream)
        geom_t, bin_t, img_t = geom.release(), binb.release(), img.release()
        _lib.check(rc, "gsr_forward_views")
        ctx.views, ctx.keeps, ctx.stats = views, keeps, st
        ctx.dims = (B, N, K, H, W)
        ctx.present = (shc is not None, col is not None, sc is not None, cov is not None)
        empty = torch.empty(0, device=dev)
        # the camera constants travel as raw pointers: saved too, so that an in-place edit between forward and backward raises
        ctx.save_for_backward(*[t if t is not None else empty for t in (m3, shc, col, op, sc, rot, cov)], radii,
                              geom_t, bin_t, img_t, *[t for keep in keeps for t in keep])
        ctx.shapes = (means3D.shape, None if sh is None else sh.shape,
                      None if colors_precomp is None else colors_precomp.shape, opacities.shape,
                      None if scales is None else scales.shape, None if rotations is None else rotations.shape,
                      None if cov3Ds_precomp is None else cov3Ds_precomp.shape)
        ctx.mark_non_differentiable(radii)
        # no zero tensors for outputs the loss does not use: autograd would otherwise fill an int32 [N] "gradient" of radii
        # (a 4 MB memset per step at 1M Gaussians) before every backward; the backward below handles None
        ctx.set_materialize_grads(False)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth, g_alpha):
        lib = _lib.load()
        B, N, K, H, W = ctx.dims
        m3, shc, col, op, sc, rot, cov, radii, geom, binb, img = ctx.saved_tensors[:11]
        has_sh, has_col, has_sr, has_cov = ctx.present
        dev = radii.device
        z = lambda g: None if g is None else g.to(torch.float32).contiguous()     # (None -> NULL = zeros: include/gsr.h, ABI 6)
        gc, gd, ga = z(g_color), z(g_depth), z(g_alpha)
        f = lambda *s: (torch.empty if N > 0 else torch.zeros)(*s, dtype=torch.float32, device=dev)
        d_m3, d_m2, d_op = f(N, 3), f(B, N, 3), f(N, 1)
        d_sh = f(N, K, 3) if has_sh else None
        d_col = f(N, 3) if has_col else None
        d_sc, d_rot = (f(N, 3), f(N, 4)) if has_sr else (None, None)
        d_cov = f(N, 6) if has_cov else None
        if N > 0:
            P = _lib.ptr
            tmp = _lib.Scratch(dev)
            with torch.cuda.device(dev):
                stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                rc = lib.gsr_backward_views(
                    ctx.views, B, N, K, P(m3), P(shc) if has_sh else None, P(col) if has_col else None,
                    P(op), P(sc) if has_sr else None, P(rot) if has_sr else None, P(cov) if has_cov else None,
                    P(radii), P(gc), P(gd), P(ga), P(geom), P(binb), P(img), C.byref(ctx.stats),
                    P(d_m3), P(d_m2), P(d_sh), P(d_col), P(d_op), P(d_sc), P(d_rot), P(d_cov), tmp.alloc, stream)
            tmp.release()
            ctx.stats.bwd_prepared = 0         # one-shot (see rasterizer.py)
            _lib.check(rc, "gsr_backward_views")
        sh_ = ctx.shapes
        r = lambda g, shape: None if g is None or shape is None else g.reshape(shape)
        return (r(d_m3, sh_[0]), d_m2, r(d_sh, sh_[1]), r(d_col, sh_[2]), r(d_op, sh_[3]),
                r(d_sc, sh_[4]), r(d_rot, sh_[5]), r(d_cov, sh_[6]), None, None)


def rasterize_views(means3D, means2D, opacities, raster_settings: Sequence[GaussianRasterizationSettings],
                    shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
    """Render `len(raster_settings)` cameras of the same Gaussians.

    Arguments as `GaussianRasterizer.forward` (gs_renderer.py:800-809) except `means2D`, the
    screen-space gradient holder, which is `[B, N, 3]` (one `[N,3]` holder per view, as the reference
    creates one per render: gs_renderer.py:727-739). Returns `(color [B,3,H,W], radii [B,N] int32,
    depth [B,1,H,W], alpha [B,1,H,W])`; gradients of the shared inputs are summed over the views.
    More than GSR_MAX_VIEWS (16) cameras are rendered in chunks of 16."""
    settings = tuple(raster_settings)
    if len(settings) <= _lib.GSR_MAX_VIEWS:
        return _RasterizeViews.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                     cov3D_precomp, settings, torch.is_grad_enabled())
    parts = []
    for i0 in range(0, len(settings), _lib.GSR_MAX_VIEWS):
        sl = slice(i0, i0 + _lib.GSR_MAX_VIEWS)
        parts.append(_RasterizeViews.apply(means3D, means2D[sl], shs, colors_precomp, opacities, scales, rotations,
                                           cov3D_precomp, settings[sl], torch.is_grad_enabled()))
    return tuple(torch.cat([p[i] for p in parts], 0) for i in range(4))
