"""Host-side mirror of the `diff_gaussian_rasterization` surface DreamGaussian uses.

Same names, argument meaning and error behaviour as the external package the reference
imports at gs_renderer.py:10-13 and calls at gs_renderer.py:745-760, 800-809:
`GaussianRasterizationSettings` (12-field NamedTuple), `GaussianRasterizer(raster_settings=)`
whose call returns `(color[3,H,W], radii[N] int32, depth[1,H,W], alpha[1,H,W])`, and an
autograd.Function whose backward returns gradients positionally for
`means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp`.

All arithmetic runs in the hand-written HIP kernels of libgsr.so (include/gsr.h) on the
tensors' device and torch's current stream; torch is used for memory, streams and autograd
plumbing only. There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _lib, _testing


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


_last_stats = {}
_last_pending = None       # (GsrStats, dict) of an asynchronous forward whose counts have not been collected yet (last_stats() does)
# OPT-IN: gsr_forward returns without waiting for its instance counters when it could size its scratch from earlier calls of the same
# shape (GSR_VIEW_ASYNC_STATS, include/gsr.h): the ~25 us of host wait per forward overlap with the caller's Python. The price is the
# failure mode written down there (an overflow of the speculative capacity leaves NaN images and raises at the thread's NEXT forward).
_async_forward = False
# OPT-IN: bit-reproducible gradients (GSR_VIEW_DETERMINISTIC: the compositing backward sums in 64-bit fixed point instead of with
# float atomics; SURVEY 5's "deterministic-mode flag")
_deterministic = False
# The backward normally hands the forward's GsrStats back to gsr_backward (no host round trip). Setting this
# to False exercises the ABI's other documented mode (fwd_stats == NULL: the library reads the counters back
# from the device, blocking) -- used by tests/test_parity_gpu.py.
_pass_fwd_stats = True


# The torch binding in C++ (csrc/gsr_torch.cpp -> _gsr_torch.so, dreamgaussian_amd/build.py): the same host logic as the
# autograd.Function below -- checks, allocations, view struct, gradient carving, the two C-ABI calls -- without the Python in between
# (and without the three scratch callbacks that re-enter it): at DreamGaussian's own sizes that Python is most of a step. Used by the
# three entry points when the module has been built; `use_cpp_binding(False)` (or a missing module) selects the ctypes path.
_binding = None
_binding_state = 0          # 0 = not tried, 1 = loaded, -1 = unavailable
_use_binding = True
_last_via_binding = False


def use_cpp_binding(on: bool) -> bool:
    """Route the rasterizer's entry points through the C++ binding (default when it has been built) or through the ctypes path;
    returns the old setting."""
    global _use_binding
    old, _use_binding = _use_binding, bool(on)
    return old


def _get_binding():
    global _binding, _binding_state
    if _binding_state == 0:
        _binding_state = -1
        import importlib.util
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_gsr_torch.so")
        if os.path.exists(path):
            try:
                _lib.load()                                   # (loads libgsr.so and checks its ABI version first)
                spec = importlib.util.spec_from_file_location("dreamgaussian_amd._gsr_torch", path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                if mod.abi_version() == _lib.GSR_ABI_VERSION:
                    mod.init(_lib.LIB_PATH)
                    _binding, _binding_state = mod, 1
            except Exception as e:                            # a stale or unloadable module: the ctypes path is complete on its own
                import warnings
                warnings.warn(f"dreamgaussian_amd: the C++ torch binding could not be loaded ({e}); using the ctypes path")
    return _binding if _binding_state == 1 else None


def binding_loaded() -> bool:
    return _get_binding() is not None


def _via_binding(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs, raw, sh_rest):
    """The call through _gsr_torch.rasterize, or None when the C++ path does not apply (module absent or switched off, settings that are
    not tensors, the tests' fwd_stats = NULL mode)."""
    global _last_via_binding, _last_pending
    if not (_use_binding and _pass_fwd_stats):
        return None
    b = _get_binding()
    if b is None or not all(type(x) is torch.Tensor for x in (rs.bg, rs.viewmatrix, rs.projmatrix, rs.campos)):
        return None
    extra = (_lib.GSR_VIEW_ASYNC_STATS if _async_forward else 0) | (_lib.GSR_VIEW_DETERMINISTIC if _deterministic else 0)
    out = b.rasterize(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, sh_rest,
                      rs.bg, rs.viewmatrix, rs.projmatrix, rs.campos, int(rs.image_height), int(rs.image_width), float(rs.tanfovx),
                      float(rs.tanfovy), float(rs.scale_modifier), int(rs.sh_degree), bool(rs.prefiltered), bool(rs.debug), bool(raw),
                      extra, _testing._current.get("k6_compact") == 1)
    _last_via_binding, _last_pending = True, None
    return tuple(out)


def set_async_forward(on: bool) -> bool:
    """Opt into (or out of) the forward that does not wait for its instance counters (GSR_VIEW_ASYNC_STATS); returns the old setting."""
    global _async_forward
    old, _async_forward = _async_forward, bool(on)
    return old


def peek_stats(complete: bool = False) -> dict:
    """The statistics of the calling thread's most recent forward as they stand (an asynchronous forward's counts are -1 and
    `pending` != 0 until `last_stats()` -- or `complete=True` here -- has collected them)."""
    if _last_via_binding:
        v = _binding.last_stats(bool(complete))
        return dict(M=v[0], M_ref=v[1], V=v[2], max_tile=v[3], seg_shift=v[4], bwd_prepared=v[5], speculated=v[6], pending=v[7],
                    N=v[8], H=v[9], W=v[10], K=v[11])
    return last_stats() if complete else dict(_last_stats)


def set_deterministic(on: bool) -> bool:
    """Opt into (or out of) the bit-reproducible backward (GSR_VIEW_DETERMINISTIC, include/gsr.h); returns the old setting."""
    global _deterministic
    old, _deterministic = _deterministic, bool(on)
    return old


def last_stats() -> dict:
    """Scene statistics of the most recent forward (V, M of SURVEY 8(d)). After an asynchronous forward this collects its counts
    (blocking until they have arrived; call it from the thread that rendered)."""
    global _last_pending
    if _last_via_binding:
        return peek_stats(complete=True)
    if _last_pending is not None:
        stats, _last_pending = _last_pending, None
        rc = _lib.load().gsr_forward_complete(C.byref(stats))
        _last_stats.update(M=stats.num_instances, M_ref=stats.num_instances_ref, V=stats.num_visible, max_tile=stats.max_tile_count,
                           speculated=stats.speculated, pending=0)
        _lib.check(rc, "gsr_forward_complete")
    return dict(_last_stats)


def _f32c(t: Optional[torch.Tensor], device) -> Optional[torch.Tensor]:
    if t is None or t.numel() == 0:
        return None
    if t.device != device:
        raise RuntimeError(f"all rasterizer inputs must be on {device}, got {t.device}")
    if t.dtype is torch.float32 and t.is_contiguous():
        return t                                          # (no detach(): grad mode is off inside forward(), and a tensor op is ~1 us of host)
    return t.detach().to(torch.float32).contiguous()


class _NoGuard:
    def __enter__(self): return None
    def __exit__(self, *a): return False


_NO_GUARD = _NoGuard()


def _on_device(dev):
    """`with torch.cuda.device(dev)` only when dev is not the current device already: the guard is ~4 us of Python per use, and the
    step of a small scene is host-bound."""
    idx = dev.index
    return _NO_GUARD if (idx is None or torch.cuda.current_device() == idx) else torch.cuda.device(dev)


def _view_struct(rs: GaussianRasterizationSettings, device, raw: bool = False, no_backward: bool = False):
    keep = []
    flags = ((_lib.GSR_VIEW_NO_BACKWARD if no_backward else 0) | (_lib.GSR_VIEW_ASYNC_STATS if _async_forward else 0)
             | (_lib.GSR_VIEW_DETERMINISTIC if _deterministic else 0))
    for x, tbit in ((rs.bg, 0), (rs.viewmatrix, _lib.GSR_VIEW_VIEWMATRIX_T), (rs.projmatrix, _lib.GSR_VIEW_PROJMATRIX_T), (rs.campos, 0)):
        ok = type(x) is torch.Tensor and x.dtype is torch.float32 and x.device == device
        if ok and x.is_contiguous():
            keep.append(x)                                # already what the kernels read: no copy, no new tensor object
        elif ok and tbit and x.dim() == 2 and x.shape == (4, 4) and x.stride() == (1, 4):
            # the reference's `world_view_transform`: a .transpose(0, 1) VIEW of the row-major w2c (gs_renderer.py:662-664). Its
            # storage holds the transpose; the kernels read it as such (GSR_VIEW_*_T) -- no .contiguous() copy kernel per render
            keep.append(x)
            flags |= tbit
        else:
            keep.append(torch.as_tensor(x).to(device=device, dtype=torch.float32).contiguous().reshape(-1))
    if keep[0].numel() != 3 or keep[1].numel() != 16 or keep[2].numel() != 16 or keep[3].numel() != 3:
        raise RuntimeError("bg/campos must have 3 elements and viewmatrix/projmatrix 16")
    v = _lib.GsrView(int(rs.image_height), int(rs.image_width), float(rs.tanfovx),
                     float(rs.tanfovy), float(rs.scale_modifier), int(rs.sh_degree),
                     int(bool(rs.prefiltered)), int(bool(rs.debug)),
                     keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(),
                     1 if raw else 0, flags, None, None)
    return v, keep


def _require_gpu(t: torch.Tensor):
    if t.device.type != "cuda":
        raise RuntimeError(
            "dreamgaussian_amd rasterizer runs on an MI355X (torch device 'cuda' on ROCm) only; "
            f"got a tensor on '{t.device}'. There is no CPU fallback.")


def carve_gradients(N: int, widths, device):
    """ONE allocation for all gradients of a backward, gradient i = N x widths[i] floats starting at a 64-element (256-byte) boundary
    (offs[i]); K6 writes every element of every gradient, so nothing is cleared. views.allreduce_grads relies on this layout: the
    parameter gradients tile one storage with gaps of at most 63 elements and are reduced in place over their span."""
    offs, total = [], 0
    for w_ in widths:
        offs.append(total)
        total += (N * w_ + 63) & ~63
    flat = (torch.empty if N > 0 else torch.zeros)(max(total, 1), dtype=torch.float32, device=device)
    return flat, offs


def _gradient_widths(K, k_rest, has_sh, has_col, has_sr, has_cov):
    """Floats per Gaussian of each gradient, in the order they are carved: the parameter gradients first, the per-view means2D
    gradient last -- views.allreduce_grads reduces the span of the parameter gradients in place and must not touch the screen-space one."""
    k_sh = K - k_rest
    return [3, 1, 3 * k_sh if has_sh else 0, 3 if has_col else 0, 3 if has_sr else 0, 4 if has_sr else 0,
            6 if has_cov else 0, 3 * k_rest, 3]


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                cov3Ds_precomp, raster_settings, raw=False, sh_rest=None, grad_mode=True):
        _require_gpu(means3D)
        ctx.raw = bool(raw)
        lib = _lib.load()
        dev = means3D.device
        rs = raster_settings
        H, W = int(rs.image_height), int(rs.image_width)
        m3 = _f32c(means3D, dev)
        N = int(means3D.shape[0])
        if N > 0 and (means3D.dim() != 2 or means3D.shape[1] != 3):
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        shc = _f32c(sh, dev)
        col = _f32c(colors_precomp, dev)
        op = _f32c(opacities, dev)
        sc = _f32c(scales, dev)
        rot = _f32c(rotations, dev)
        cov = _f32c(cov3Ds_precomp, dev)
        K = int(shc.shape[1]) if shc is not None else 0
        if shc is not None and (shc.dim() != 3 or shc.shape[0] != N or shc.shape[2] != 3):
            raise RuntimeError("shs must have dimensions (num_points, num_coeffs, 3)")
        rest = _f32c(sh_rest, dev)                       # split layout: sh = features_dc [N,1,3], sh_rest = features_rest [N,K-1,3]
        if rest is not None:
            if shc is None or K != 1 or rest.dim() != 3 or rest.shape[0] != N or rest.shape[2] != 3:
                raise RuntimeError("split SH input needs features_dc (num_points, 1, 3) and features_rest (num_points, K-1, 3)")
            K = 1 + int(rest.shape[1])

        color = torch.empty(3, H, W, dtype=torch.float32, device=dev)
        depth = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        alpha = torch.empty(1, H, W, dtype=torch.float32, device=dev)
        radii = torch.empty(N, dtype=torch.int32, device=dev)     # every element written by K1
        geom, binb, img = _lib.Scratch(dev), _lib.Scratch(dev), _lib.Scratch(dev)
        stats = _lib.GsrStats()
        with _on_device(dev):
            # inference -- torch.no_grad() around the call (`grad_mode`, taken by the wrappers below BEFORE .apply: inside forward() grad
            # mode is always off, and needs_input_grad mirrors requires_grad whatever the mode), or no input that requires a gradient:
            # the backward's accumulators are not prepared
            want_bwd = bool(grad_mode and any(ctx.needs_input_grad))
            view, keep = _view_struct(rs, dev, ctx.raw, no_backward=not want_bwd)
            if rest is not None:
                view.shs_rest = rest.data_ptr()
                keep.append(rest)
            # the backward's gradients are carved out of ONE allocation; it is made HERE so that the forward's compositing kernel can
            # clear it on the side where the backward would otherwise have to (GsrView.grad_clear: large scenes, whose per-Gaussian
            # backward then writes the rows of the live Gaussians only). stats.bwd_prepared == 2 says it did; the block is handed to
            # the first backward either way (cleared or not: the streaming per-Gaussian backward writes every element)
            # ... only where the library would use it (its rule, gsr_api.hip k6_compact_for: one view, concatenated SH layout, >= 64 MB
            # of gradients): otherwise the block is allocated by the backward, as before round 5 -- a render whose backward never runs
            # does not pay for it, and the loss computation does not hold it (round-5 advisor)
            grad_flat = None
            if want_bwd and N > 0 and rest is None and (N * (3 * (K if shc is not None else 1) + 14) * 4 >= (64 << 20)
                                                         or _testing._current.get("k6_compact") == 1):      # (test hook: the small scenes of the suite)
                widths = _gradient_widths(K, 0 if rest is None else int(rest.shape[1]), shc is not None, col is not None,
                                          sc is not None, cov is not None)
                grad_flat = carve_gradients(N, widths, dev)
                view.grad_clear = grad_flat[0].data_ptr()
                view.grad_clear_floats = grad_flat[0].numel()
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            rc = lib.gsr_forward(C.byref(view), N, K, _lib.ptr(m3), _lib.ptr(shc), _lib.ptr(col),
                                 _lib.ptr(op), _lib.ptr(sc), _lib.ptr(rot), _lib.ptr(cov),
                                 _lib.ptr(color), _lib.ptr(depth), _lib.ptr(alpha), _lib.ptr(radii),
                                 geom.alloc, binb.alloc, img.alloc, C.byref(stats), stream)
        geom_t, bin_t, img_t = geom.release(), binb.release(), img.release()
        _lib.check(rc, "gsr_forward")
        global _last_pending, _last_via_binding
        _last_via_binding = False
        _last_pending = stats if stats.pending else None   # (an asynchronous forward: the counts are -1 until last_stats() collects them)
        _last_stats.update(M=stats.num_instances, M_ref=stats.num_instances_ref,
                           V=stats.num_visible, max_tile=stats.max_tile_count, N=N, H=H, W=W, K=K,
                           seg_shift=stats.seg_shift, bwd_prepared=stats.bwd_prepared, speculated=stats.speculated, pending=stats.pending)
        ctx.raster_settings = rs
        ctx.view = (view, keep)            # the backward reuses the struct (and keeps its device constants alive)
        ctx.dims = (N, K)
        ctx.fwd_stats = stats
        ctx.grad_flat = grad_flat          # (flat, offsets): taken by the first backward
        ctx.present = (shc is not None, col is not None, sc is not None, cov is not None)
        ctx.k_rest = 0 if rest is None else int(rest.shape[1])
        ctx.rest_shape = None if rest is None else sh_rest.shape
        empty = torch.empty(0, device=dev)
        ctx.save_for_backward(m3 if m3 is not None else empty, shc if shc is not None else empty,
                              col if col is not None else empty, op if op is not None else empty, sc if sc is not None else empty,
                              rot if rot is not None else empty, cov if cov is not None else empty,
                              radii, geom_t, bin_t, img_t,
                              *keep)       # the camera constants travel as raw pointers: saved, so that an in-place edit before the backward raises
        ctx.shapes = (means3D.shape, means2D.shape, None if sh is None else sh.shape,
                      None if colors_precomp is None else colors_precomp.shape, opacities.shape,
                      None if scales is None else scales.shape,
                      None if rotations is None else rotations.shape,
                      None if cov3Ds_precomp is None else cov3Ds_precomp.shape)
        ctx.mark_non_differentiable(radii)
        # no zero tensors for outputs the loss does not use: autograd would otherwise fill an int32 [N] "gradient" of radii
        # (a 4 MB memset per step at 1M Gaussians) before every backward; the backward below handles None
        ctx.set_materialize_grads(False)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        lib = _lib.load()
        (m3, shc, col, op, sc, rot, cov, radii, geom, binb, img) = ctx.saved_tensors[:11]
        has_sh, has_col, has_sr, has_cov = ctx.present
        N, K = ctx.dims
        rs = ctx.raster_settings
        dev = radii.device                 # (m3 is an empty placeholder when N == 0)
        H, W = int(rs.image_height), int(rs.image_width)
        # (every avoided tensor op is ~1 us of host time, and at DreamGaussian's sizes the step is host-bound: an incoming gradient
        # that is already fp32 and contiguous is taken as it is, a gradient view is ONE as_strided, a reshape to the shape it has is skipped)
        # an output the loss never used arrives as None (set_materialize_grads(False)) and goes to the library as NULL = zeros (ABI 6):
        # no zero image, no fill kernel -- stage 1 never differentiates depth (main.py:198-275)
        z = lambda g: (None if g is None
                       else (g if g.dtype is torch.float32 and g.is_contiguous() else g.to(torch.float32).contiguous()))
        gc, gd, ga = z(grad_color), z(grad_depth), z(grad_alpha)
        # K6 writes every element of every gradient (exact zeros for culled Gaussians): no memset
        # one allocation for all gradients, each carved out at a 256-byte boundary
        k_rest = ctx.k_rest                              # split SH: dL/dfeatures_dc and dL/dfeatures_rest are separate tensors
        k_sh = K - k_rest
        widths = _gradient_widths(K, k_rest, has_sh, has_col, has_sr, has_cov)
        if ctx.grad_flat is not None:      # the forward's allocation (cleared there when fwd_stats.bwd_prepared == 2); one-shot like bwd_prepared
            flat, offs = ctx.grad_flat
            ctx.grad_flat = None
        else:                              # a second backward of this forward (retain_graph), or N == 0
            flat, offs = carve_gradients(N, widths, dev)
            if ctx.fwd_stats.bwd_prepared == 2:
                ctx.fwd_stats.bwd_prepared = 1
        def part(i, *shape):                             # gradient i: N x widths[i] floats at offs[i], contiguous
            if len(shape) == 2:
                return flat.as_strided(shape, (shape[1], 1), offs[i])
            return flat.as_strided(shape, (shape[1] * shape[2], shape[2], 1), offs[i])
        d_m3, d_op, d_m2 = part(0, N, 3), part(1, N, 1), part(8, N, 3)
        d_sh = part(2, N, k_sh, 3) if has_sh else None
        d_rest = part(7, N, k_rest, 3) if k_rest else None
        d_col = part(3, N, 3) if has_col else None
        d_sc = part(4, N, 3) if has_sr else None
        d_rot = part(5, N, 4) if has_sr else None
        d_cov = part(6, N, 6) if has_cov else None
        if N > 0:
            tmp = _lib.Scratch(dev)
            with _on_device(dev):
                view, keep = ctx.view
                if d_rest is not None:
                    view.dL_dshs_rest = d_rest.data_ptr()
                stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
                P = _lib.ptr
                rc = lib.gsr_backward(
                    C.byref(view), N, K, P(m3), P(shc) if has_sh else None, P(col) if has_col else None,
                    P(op), P(sc) if has_sr else None, P(rot) if has_sr else None,
                    P(cov) if has_cov else None, P(radii), P(gc), P(gd), P(ga),
                    P(geom), P(binb), P(img), C.byref(ctx.fwd_stats) if _pass_fwd_stats else None, P(d_m3), P(d_m2), P(d_sh), P(d_col), P(d_op),
                    P(d_sc), P(d_rot), P(d_cov), tmp.alloc, stream)
            tmp.release()
            ctx.fwd_stats.bwd_prepared = 0     # one-shot: a second backward of this forward (retain_graph) clears its own accumulators
            _lib.check(rc, "gsr_backward")
        s = ctx.shapes
        rs_ = lambda g, shape: None if g is None or shape is None else (g if g.shape == shape else g.reshape(shape))
        return (rs_(d_m3, s[0]), rs_(d_m2, s[1]) if tuple(s[1]) == (N, 3) else None, rs_(d_sh, s[2]),
                rs_(d_col, s[3]), rs_(d_op, s[4]), rs_(d_sc, s[5]), rs_(d_rot, s[6]), rs_(d_cov, s[7]),
                None, None, rs_(d_rest, ctx.rest_shape), None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                        cov3Ds_precomp, raster_settings):
    out = _via_binding(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, False, None)
    if out is not None:
        return out
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales,
                                     rotations, cov3Ds_precomp, raster_settings, False, None, torch.is_grad_enabled())


def rasterize_gaussians_raw(means3D, means2D, sh, opacity_raw, scaling_raw, rotation_raw, raster_settings):
    """Opt-in entry beside the drop-in one: takes DreamGaussian's RAW parameters
    (`_opacity`, `_scaling`, `_rotation`, gs_renderer.py:144-160) and runs sigmoid / exp / normalise
    (gs_renderer.py:134-142, 196-216) and their backward inside the per-Gaussian kernels -- five
    elementwise launches and their five backward launches per render disappear. Same outputs as
    `rasterize_gaussians(means3D, means2D, sh, None, sigmoid(o), exp(s), normalize(q), None, settings)`."""
    out = _via_binding(means3D, means2D, sh, None, opacity_raw, scaling_raw, rotation_raw, None, raster_settings, True, None)
    if out is not None:
        return out
    return _RasterizeGaussians.apply(means3D, means2D, sh, None, opacity_raw, scaling_raw, rotation_raw,
                                     None, raster_settings, True, None, torch.is_grad_enabled())


def rasterize_gaussians_split(means3D, means2D, features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw,
                              raster_settings):
    """The whole opt-in front end of SURVEY 8(f) rank 2: RAW parameters as `rasterize_gaussians_raw` AND the two SH
    tensors DreamGaussian keeps (`_features_dc` [N,1,3], `_features_rest` [N,K-1,3]) read where they are -- no
    `torch.cat` copy per render (gs_renderer.py:209-212), gradients written straight into two tensors of the same
    shapes. Same outputs as `rasterize_gaussians_raw(means3D, means2D, cat((features_dc, features_rest), 1), ...)`."""
    out = _via_binding(means3D, means2D, features_dc, None, opacity_raw, scaling_raw, rotation_raw, None, raster_settings, True, features_rest)
    if out is not None:
        return out
    return _RasterizeGaussians.apply(means3D, means2D, features_dc, None, opacity_raw, scaling_raw, rotation_raw,
                                     None, raster_settings, True, features_rest, torch.is_grad_enabled())


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        """Boolean [N]: passes the frustum rule (view-space z > 0.2)."""
        _require_gpu(positions)
        lib = _lib.load()
        dev = positions.device
        pos = _f32c(positions, dev)
        N = int(positions.shape[0])
        vis = torch.zeros(N, dtype=torch.uint8, device=dev)
        with torch.no_grad(), torch.cuda.device(dev):
            view, keep = _view_struct(self.raster_settings, dev)
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            rc = lib.gsr_mark_visible(C.byref(view), N, _lib.ptr(pos), _lib.ptr(vis), stream)
        _lib.check(rc, "gsr_mark_visible")
        return vis.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = torch.Tensor([]).to(means3D.device)
        if shs is None:
            shs = empty
        if colors_precomp is None:
            colors_precomp = empty
        if scales is None:
            scales = empty
        if rotations is None:
            rotations = empty
        if cov3D_precomp is None:
            cov3D_precomp = empty
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales,
                                   rotations, cov3D_precomp, raster_settings)
