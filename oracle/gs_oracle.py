"""CPU oracle for the differentiable Gaussian-splat rasterizer (TEST INFRASTRUCTURE ONLY).

This file is the parity checker for the hand-written HIP path in
``dreamgaussian_amd/csrc``.  It is a pure-PyTorch, device-agnostic, float32/float64
restatement of the algorithm that DreamGaussian reaches through
``GaussianRasterizer(raster_settings)(means3D, means2D, shs, ...)``
(reference call site: gs_renderer.py:745-809).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the
product path (``dreamgaussian_amd``) never does.

PARITY UNPINNED: the arithmetic of this path lives in the un-vendored, un-pinned pip
package ``diff_gaussian_rasterization`` (ashawkey's depth/alpha fork of
graphdeco-inria/diff-gaussian-rasterization; readme.md:30-32) which is absent from
/root/reference, and the reference ships no tests or golden vectors.  What IS pinned to
in-tree reference code (checked by tests/test_oracle_golden.py against fixtures produced
by importing the reference's own Python) is:
  * rotation / covariance convention   gs_renderer.py:85-117, 128-132, 50-62
  * SH basis, sign and layout          sh_utils.py:26-112, gs_renderer.py:209-212, 793
  * camera / projection convention     gs_renderer.py:629-671, cam_utils.py:44-63
  * output order and shapes            gs_renderer.py:800, 815-822
Everything else follows the published 3DGS algorithm (Kerbl et al. 2023, EWA splatting)
as written down in SURVEY.md Appendix A.3-A.7.

Gradients come from torch autograd over the forward restatement, with the two
straight-through conventions of the published backward (the min(0.99, .) clamp passes
gradient; the alpha<1/255, power>0 and T<1e-4 tests are constants), so they are an
independent check of the hand-derived HIP backward.
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import numpy as np
import torch

import time

TIMERS = {"composite_fwd": 0.0, "composite_bwd": 0.0}   # seconds spent compositing (bench.py's cpu_baseline)
# Ambiguity bands of the discrete per-pair decisions. Any fp32 implementation (this one, the CUDA
# reference) projects the mean to pixels in fp32: an absolute position error of a few ulp of the
# image size (FRAGILE_PX_PER_PIXEL * max(W,H) pixels; 2e-4 px at 800). Through the exponent's
# gradient |Sigma'^-1 d| that becomes a RELATIVE alpha error of up to ~2.5e-4 for small splats, on top
# of the ~1e-6 of exp itself; the transmittance inherits the (partly cancelling) alpha errors of
# every Gaussian in front. A decision inside its band may legitimately fall either way.
FRAGILE_ALPHA_REL = 2e-5         # floor of the alpha >= 1/255 band (exp, conic rounding)
FRAGILE_PX_PER_PIXEL = 2.5e-7    # position uncertainty per pixel of image extent
FRAGILE_T_REL_AT_800 = 2.5e-4    # T(1-alpha) >= 1e-4 band at an 800-pixel frame (scales with the extent)
FRAGILE_PX = [2e-4]              # set per render by rasterize()
FRAGILE_T_REL = [2.5e-4]
COMPOSITE_INFO = {}
LAST_FRAGILE = {}           # of the last full forward: "pixels" [H,W] bool, "gaussians" [N] bool         # side channel of composite_tile (last call): fragile pixel / Gaussian masks

BLOCK = 16  # binning granularity in pixels (part of the numerics: SURVEY §7 hard part 2)

# SH constants: identical values to sh_utils.py:26-48
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
      -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
      -0.4570457994644658, 1.445305721320277, -0.5900435899266435)


class Settings(NamedTuple):
    """Same 12 fields, same order as the reference builds them (gs_renderer.py:745-758)."""
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
    prefiltered: bool = False
    debug: bool = False


# --------------------------------------------------------------------------------------
# per-Gaussian stage  (Python twins gs_renderer.py:85-132)
# --------------------------------------------------------------------------------------

def rotation_matrix(q: torch.Tensor) -> torch.Tensor:
    """Quaternion (r,x,y,z) -> R, NOT normalised here (the caller normalises,
    gs_renderer.py:142,201-202); element formulas as gs_renderer.py:97-105."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y),
    ], dim=-1).reshape(-1, 3, 3)
    return R


def covariance3d(scales: torch.Tensor, mod: float, rotations: torch.Tensor) -> torch.Tensor:
    """Sigma = (R diag(mod*s)) (R diag(mod*s))^T  (gs_renderer.py:108-117,128-132) -> [N,3,3]."""
    L = rotation_matrix(rotations) * (mod * scales)[:, None, :]
    return L @ L.transpose(1, 2)


def sym_from6(c6: torch.Tensor) -> torch.Tensor:
    """6-vector (S00,S01,S02,S11,S12,S22) (gs_renderer.py:50-59) -> [N,3,3]."""
    i = torch.tensor([0, 1, 2, 1, 3, 4, 2, 4, 5], device=c6.device)
    return c6[:, i].reshape(-1, 3, 3)


def eval_sh_color(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """sh [N,K,3] coefficient-major (gs_renderer.py:209-212); same polynomial as
    sh_utils.py:74-100 (which takes [...,C,K]); returns [N,3] BEFORE the +0.5/clamp."""
    res = C0 * sh[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5]
                   + C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                   + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + C3[0] * y * (3 * xx - yy) * sh[:, 9]
                       + C3[1] * xy * z * sh[:, 10]
                       + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11]
                       + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
                       + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13]
                       + C3[5] * z * (xx - yy) * sh[:, 14]
                       + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return res


def preprocess(means3D, means2D, opacities, shs, colors_precomp, scales, rotations,
               cov3D_precomp, S: Settings) -> dict:
    """Per-Gaussian projection. All outputs are autograd-connected torch tensors except the
    integer/boolean decisions (radius, rect, valid), which are constants."""
    dt, dev = means3D.dtype, means3D.device
    H, W = int(S.image_height), int(S.image_width)
    V = S.viewmatrix.to(dt)
    Pm = S.projmatrix.to(dt)
    N = means3D.shape[0]
    gx, gy = (W + BLOCK - 1) // BLOCK, (H + BLOCK - 1) // BLOCK

    # row-vector convention: p' = [p,1] @ M  (gs_renderer.py:662-670; flat m[4i+j]=M[i][j])
    p_view = means3D @ V[:3, :3] + V[3, :3]
    p_hom = means3D @ Pm[:3, :] + Pm[3, :]
    p_w = 1.0 / (p_hom[:, 3] + 1e-7)
    ndc = p_hom[:, :2] * p_w[:, None]
    if means2D is not None:
        # the screen-space grad holder (gs_renderer.py:727-739): an additive NDC offset whose
        # gradient is dL/d(ndc) = dL/d(pixel) * 0.5*(W,H)   
        ndc = ndc + means2D[:, :2]
    tz = p_view[:, 2]
    with torch.no_grad():
        # The depth SORT KEY is pinned to one fp32 evaluation order so that the discrete
        # compositing order is reproducible bit for bit (A.4): key = ((V02*x + V12*y) + V22*z) + V32,
        # every product and sum rounded to fp32 separately (no fma). The kernels use the same chain.
        m32, V32 = means3D.detach().to(torch.float32), V.detach().to(torch.float32)
        depth_key = ((V32[0, 2] * m32[:, 0] + V32[1, 2] * m32[:, 1]) + V32[2, 2] * m32[:, 2]) + V32[3, 2]
        valid = depth_key > 0.2          # frustum rule (A.3), decided on the same fp32 value

    if cov3D_precomp is not None and cov3D_precomp.numel() > 0:
        Sigma = sym_from6(cov3D_precomp)
    else:
        Sigma = covariance3d(scales, float(S.scale_modifier), rotations)

    fx = W / (2.0 * S.tanfovx)
    fy = H / (2.0 * S.tanfovy)
    limx, limy = 1.3 * S.tanfovx, 1.3 * S.tanfovy
    tzs = torch.where(valid, tz, torch.ones_like(tz))  # keep culled rows finite
    txtz = p_view[:, 0] / tzs
    tytz = p_view[:, 1] / tzs
    inx = (txtz >= -limx) & (txtz <= limx)
    iny = (tytz >= -limy) & (tytz <= limy)
    vx = (txtz.clamp(-limx, limx) * tzs).detach()
    vy = (tytz.clamp(-limy, limy) * tzs).detach()
    # value = clamp(tx/tz)*tz ; gradient = identity when unclamped, none when clamped (A.7)
    tx = torch.where(inx, p_view[:, 0] - p_view[:, 0].detach() + vx, vx)
    ty = torch.where(iny, p_view[:, 1] - p_view[:, 1].detach() + vy, vy)
    zero = torch.zeros_like(tzs)
    J = torch.stack([fx / tzs, zero, -(fx * tx) / (tzs * tzs),
                     zero, fy / tzs, -(fy * ty) / (tzs * tzs)], dim=-1).reshape(N, 2, 3)
    Wr = V[:3, :3].t()
    T = J @ Wr
    cov2 = T @ Sigma @ T.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    valid = valid & (det != 0)
    dets = torch.where(det != 0, det, torch.ones_like(det))
    det_inv = 1.0 / dets
    conic = torch.stack([c * det_inv, -b * det_inv, a * det_inv], dim=-1)
    with torch.no_grad():
        mid = 0.5 * (a + c)
        lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
        radius = torch.ceil(3.0 * torch.sqrt(lam))
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    with torch.no_grad():
        def tile(v, g):
            v = torch.nan_to_num(v, nan=0.0, posinf=1e9, neginf=-1e9)
            return torch.clamp(torch.trunc(v / BLOCK), 0, g).to(torch.int64)
        rx0, rx1 = tile(px - radius, gx), tile(px + radius + (BLOCK - 1), gx)
        ry0, ry1 = tile(py - radius, gy), tile(py + radius + (BLOCK - 1), gy)
        tiles = (rx1 - rx0) * (ry1 - ry0)
        valid = valid & (tiles > 0)
        # how far the two discontinuous integer decisions sit from flipping (for the tests: a radius / tile-count mismatch of an
        # fp32 implementation must BE such a boundary case): 3 sqrt(lambda) against the next integer, the four rect edges against
        # the next tile boundary (in tiles)
        radius_raw = 3.0 * torch.sqrt(lam)
        edges = torch.stack([px - radius, px + radius + (BLOCK - 1), py - radius, py + radius + (BLOCK - 1)], -1) / BLOCK
        edge_margin = (edges - torch.round(edges)).abs().amin(-1)

    if colors_precomp is not None and colors_precomp.numel() > 0:
        color = colors_precomp
        clamped = torch.zeros(N, 3, dtype=torch.bool, device=dev)
    else:
        d = means3D - S.campos.to(dt)
        d = d / d.norm(dim=1, keepdim=True)
        raw = eval_sh_color(int(S.sh_degree), shs, d) + 0.5
        clamped = raw < 0
        color = torch.clamp_min(raw, 0.0)  # gs_renderer.py:793

    return dict(valid=valid, xy=torch.stack([px, py], -1), depth=tz, depth_key=depth_key, conic=conic,
                opacity=opacities.reshape(-1), color=color, clamped=clamped,
                radius=torch.where(valid, radius, torch.zeros_like(radius)).to(torch.int32),
                rect=torch.stack([rx0, ry0, rx1, ry1], -1), tiles=torch.where(valid, tiles, 0),
                cov2d=torch.stack([a, b, c], -1), grid=(gx, gy), radius_raw=radius_raw, edge_margin=edge_margin)


# --------------------------------------------------------------------------------------
# binning + per-tile compositing  
# --------------------------------------------------------------------------------------

def build_tile_lists(pre: dict):
    """Instance list sorted by (tile, depth bits, Gaussian index) -> (ids[M], ranges[T+1])."""
    gx, gy = pre["grid"]
    valid = pre["valid"].cpu().numpy()
    rect = pre["rect"].cpu().numpy()
    depth = pre["depth_key"].cpu().numpy()
    idx = np.nonzero(valid)[0]
    r = rect[idx]
    w = (r[:, 2] - r[:, 0]).astype(np.int64)
    h = (r[:, 3] - r[:, 1]).astype(np.int64)
    cnt = w * h
    M = int(cnt.sum())
    gid = np.repeat(idx, cnt)
    start = np.repeat(np.cumsum(cnt) - cnt, cnt)
    k = np.arange(M, dtype=np.int64) - start
    wrep = np.repeat(w, cnt)
    tx = np.repeat(r[:, 0], cnt) + k % np.maximum(wrep, 1)
    ty = np.repeat(r[:, 1], cnt) + k // np.maximum(wrep, 1)
    tile = ty * gx + tx
    dbits = depth[gid].view(np.uint32).astype(np.int64)  # depth>0.2 so bit order = numeric
    order = np.lexsort((gid, dbits, tile))
    tile_s = tile[order]
    ranges = np.searchsorted(tile_s, np.arange(gx * gy + 1), side="left")
    return gid[order], ranges, M


ALLOW_DEVICE = [False]      # bench.py --naive-gpu sets it: the oracle as a stock-PyTorch GPU program (a baseline, not a checker)


def composite_tile(pix, xy, conic, opac, color, depth, bg):
    """Front-to-back compositing of one sorted list over a set of pixels.
    pix [P,2] pixel centres (integer coords as float); per-Gaussian tensors [n,...]."""
    dx = xy[:, 0:1] - pix[None, :, 0]
    dy = xy[:, 1:2] - pix[None, :, 1]
    power = -0.5 * (conic[:, 0:1] * dx * dx + conic[:, 2:3] * dy * dy) - conic[:, 1:2] * dx * dy
    G = torch.exp(torch.clamp_max(power, 0.0))
    alpha_raw = opac[:, None] * G
    alpha = alpha_raw + (torch.clamp_max(alpha_raw, 0.99) - alpha_raw).detach()
    with torch.no_grad():
        ok = (power <= 0) & (alpha >= 1.0 / 255.0)
    one = torch.ones_like(alpha)
    a_eff = torch.where(ok, alpha, torch.zeros_like(alpha))
    om = one - a_eff
    with torch.no_grad():
        T_after = torch.cumprod(om, dim=0)          # sequential product, same order as the loop
        keep = ok & (T_after >= 1e-4)               # monotone => equals the "done" rule
        n_contrib = torch.where(keep.any(0), keep.shape[0] - torch.flip(keep, [0]).to(torch.int8).argmax(0), 0)
        # AMBIGUOUS discrete decisions: pairs whose alpha>=1/255, T(1-alpha)>=1e-4 or power<=0 test
        # sits within fp32 evaluation noise of its threshold. Any two fp32 implementations may
        # decide these differently; the parity tests compare those pixels / Gaussians with the
        # documented relaxed bound instead of the strict one (tests/util.py).
        alive = torch.cat([torch.ones_like(T_after[:1]), T_after[:-1]], 0) >= 1e-4 * (1 - 1e-3)
        gpow = torch.sqrt((conic[:, 0:1] * dx + conic[:, 1:2] * dy) ** 2 + (conic[:, 2:3] * dy + conic[:, 1:2] * dx) ** 2)
        band = FRAGILE_ALPHA_REL + gpow * FRAGILE_PX[0]          # relative alpha uncertainty of this pair
        frag = (power <= 1e-6) & ((alpha_raw.clamp_max(0.99) * 255.0 - 1.0).abs() < band)
        # the transmittance inherits the alpha errors of every Gaussian in front: d(1 - a) / (1 - a) = a / (1 - a) * da / a, i.e.
        # amplified ~99x behind a near-opaque (alpha -> 0.99) Gaussian -- the trained stage-1 model is full of those, the synthetic
        # scenes have none. Independent per-Gaussian errors add in quadrature; three sigma, never below the calibrated floor.
        amp = torch.where(ok, a_eff / (one - a_eff), torch.zeros_like(alpha)) * band
        t_band = torch.clamp_min(3.0 * torch.sqrt(torch.cumsum(amp * amp, dim=0)), FRAGILE_T_REL[0])
        frag = frag | (ok & ((T_after * 1e4 - 1.0).abs() < t_band)) | (power.abs() < 1e-6)
        frag = frag & alive
        COMPOSITE_INFO["fragile_pix"] = frag.any(0)
        COMPOSITE_INFO["fragile_gauss"] = frag.any(1)
    a_k = torch.where(keep, alpha, torch.zeros_like(alpha))
    om_k = one - a_k
    T_incl = torch.cumprod(om_k, dim=0)
    T_before = torch.cat([torch.ones_like(T_incl[:1]), T_incl[:-1]], 0)
    w = a_k * T_before
    C = (w[:, :, None] * color[:, None, :]).sum(0)   # [P,3]
    D = (w * depth[:, None]).sum(0)
    A = w.sum(0)
    T_final = T_incl[-1] if T_incl.shape[0] > 0 else torch.ones(pix.shape[0], dtype=xy.dtype, device=xy.device)
    return C + T_final[:, None] * bg[None, :], D, A, T_final, n_contrib


class _Composite(torch.autograd.Function):
    """All tiles; memory-bounded: the backward re-runs each tile under autograd (so the
    per-tile gradient is torch's, not hand-derived)."""

    @staticmethod
    def forward(ctx, xy, conic, opac, color, depth, bg, ids, ranges, H, W, gx, tiles=None):
        dt, dv = xy.dtype, xy.device
        _t0 = time.perf_counter()
        out_c = torch.zeros(H, W, 3, dtype=dt, device=dv)
        out_d = torch.zeros(H, W, dtype=dt, device=dv)
        out_a = torch.zeros(H, W, dtype=dt, device=dv)
        out_T = torch.ones(H, W, dtype=dt, device=dv)
        out_n = torch.zeros(H, W, dtype=torch.int64, device=dv)
        out_c[:] = bg
        frag_pix = torch.zeros(H, W, dtype=torch.bool, device=dv)
        frag_gauss = torch.zeros(xy.shape[0], dtype=torch.bool, device=dv)
        tiles = range(len(ranges) - 1) if tiles is None else tiles
        for t in tiles:
            s, e = int(ranges[t]), int(ranges[t + 1])
            if e == s:
                continue
            y0, x0 = (t // gx) * BLOCK, (t % gx) * BLOCK
            y1, x1 = min(y0 + BLOCK, H), min(x0 + BLOCK, W)
            ys, xs = torch.meshgrid(torch.arange(y0, y1, device=dv), torch.arange(x0, x1, device=dv), indexing="ij")
            pix = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1).to(dt)
            g = torch.as_tensor(ids[s:e]).to(dv)
            with torch.no_grad():
                c, d, a, T, n = composite_tile(pix, xy[g], conic[g], opac[g], color[g], depth[g], bg)
            out_c[y0:y1, x0:x1] = c.reshape(y1 - y0, x1 - x0, 3)
            out_d[y0:y1, x0:x1] = d.reshape(y1 - y0, x1 - x0)
            out_a[y0:y1, x0:x1] = a.reshape(y1 - y0, x1 - x0)
            out_T[y0:y1, x0:x1] = T.reshape(y1 - y0, x1 - x0)
            out_n[y0:y1, x0:x1] = n.reshape(y1 - y0, x1 - x0)
            frag_pix[y0:y1, x0:x1] = COMPOSITE_INFO["fragile_pix"].reshape(y1 - y0, x1 - x0)
            frag_gauss[g[COMPOSITE_INFO["fragile_gauss"]]] = True
        LAST_FRAGILE["pixels"], LAST_FRAGILE["gaussians"] = frag_pix, frag_gauss
        TIMERS["composite_fwd"] += time.perf_counter() - _t0
        ctx.save_for_backward(xy, conic, opac, color, depth, bg)
        ctx.misc = (ids, ranges, H, W, gx, tiles)
        ctx.mark_non_differentiable(out_T, out_n)
        return out_c.permute(2, 0, 1).contiguous(), out_d[None], out_a[None], out_T, out_n

    @staticmethod
    def backward(ctx, g_c, g_d, g_a, _gT, _gn):
        xy, conic, opac, color, depth, bg = ctx.saved_tensors
        ids, ranges, H, W, gx, tiles = ctx.misc
        _t0 = time.perf_counter()
        grads = [torch.zeros_like(t) for t in (xy, conic, opac, color, depth)]
        g_c = g_c.permute(1, 2, 0)
        for t in tiles:
            s, e = int(ranges[t]), int(ranges[t + 1])
            if e == s:
                continue
            y0, x0 = (t // gx) * BLOCK, (t % gx) * BLOCK
            y1, x1 = min(y0 + BLOCK, H), min(x0 + BLOCK, W)
            ys, xs = torch.meshgrid(torch.arange(y0, y1, device=xy.device), torch.arange(x0, x1, device=xy.device), indexing="ij")
            pix = torch.stack([xs.reshape(-1), ys.reshape(-1)], -1).to(xy.dtype)
            g = torch.as_tensor(ids[s:e]).to(xy.device)
            leaves = [v[g].detach().requires_grad_(True) for v in (xy, conic, opac, color, depth)]
            with torch.enable_grad():
                c, d, a, _, _ = composite_tile(pix, *leaves, bg)
                loss = ((c * g_c[y0:y1, x0:x1].reshape(-1, 3)).sum()
                        + (d * g_d[0, y0:y1, x0:x1].reshape(-1)).sum()
                        + (a * g_a[0, y0:y1, x0:x1].reshape(-1)).sum())
            gl = torch.autograd.grad(loss, leaves, allow_unused=True)
            for acc, gi in zip(grads, gl):
                if gi is not None:
                    acc.index_add_(0, g, gi)
        TIMERS["composite_bwd"] += time.perf_counter() - _t0
        return (*grads, None, None, None, None, None, None, None)


def rasterize(means3D, means2D, opacities, S: Settings, shs=None, colors_precomp=None,
              scales=None, rotations=None, cov3D_precomp=None, return_aux: bool = False,
              tiles=None):
    """Full differentiable render. Returns (color[3,H,W], radii[N] i32, depth[1,H,W],
    alpha[1,H,W]) in the order the reference unpacks (gs_renderer.py:800).
    `tiles` (optional list of tile ids) restricts compositing to those 16x16 tiles (all other
    pixels keep the background): bench.py's bounded CPU-baseline sample."""
    # The oracle is a CPU checker. On a GPU it runs ONLY on request, as the "naive GPU" middle point of BASELINE.md section 3 (the same
    # stock-PyTorch program on the device: bench.py --naive-gpu) -- never as a checker and never inside the product.
    assert means3D.device.type == "cpu" or ALLOW_DEVICE[0], "the oracle is a CPU checker (gs_oracle.ALLOW_DEVICE: bench.py --naive-gpu)"
    extent = float(max(int(S.image_height), int(S.image_width)))
    FRAGILE_PX[0] = FRAGILE_PX_PER_PIXEL * extent
    FRAGILE_T_REL[0] = max(1e-4, FRAGILE_T_REL_AT_800 * extent / 800.0)
    pre = preprocess(means3D, means2D, opacities, shs, colors_precomp, scales, rotations,
                     cov3D_precomp, S)
    ids, ranges, M = build_tile_lists(pre)
    H, W = int(S.image_height), int(S.image_width)
    bg = S.bg.to(dtype=means3D.dtype, device=means3D.device)
    color, depth, alpha, T_final, n_contrib = _Composite.apply(
        pre["xy"], pre["conic"], pre["opacity"], pre["color"], pre["depth"], bg,
        ids, ranges, H, W, pre["grid"][0], tiles)
    if return_aux:
        aux = dict(pre=pre, ids=ids, ranges=ranges, M=M, V=int(pre["valid"].sum()),
                   T_final=T_final, n_contrib=n_contrib,
                   fragile_pixels=LAST_FRAGILE["pixels"].clone(), fragile_gaussians=LAST_FRAGILE["gaussians"].clone())
        return color, pre["radius"], depth, alpha, aux
    return color, pre["radius"], depth, alpha


def mark_visible(means3D, S: Settings) -> torch.Tensor:
    """Frustum test only (view-space z > 0.2), the rule the preprocess uses."""
    m32, V32 = means3D.detach().to(torch.float32), S.viewmatrix.to(torch.float32)
    return (((V32[0, 2] * m32[:, 0] + V32[1, 2] * m32[:, 1]) + V32[2, 2] * m32[:, 2]) + V32[3, 2]) > 0.2


# --------------------------------------------------------------------------------------
# cameras / synthetic scenes: shared with bench.py, so they live in the package
# (dreamgaussian_amd/synthetic.py, host-side input generators only)
# --------------------------------------------------------------------------------------
from dreamgaussian_amd.synthetic import (look_at_opengl, orbit_pose, nn3_mean_sqdist,  # noqa: E402,F401
                                         make_scene)
from dreamgaussian_amd import synthetic as _syn  # noqa: E402


def make_settings(*a, **k) -> Settings:
    return Settings(*_syn.make_settings(*a, **k))
