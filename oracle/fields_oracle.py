"""CPU oracle for the density-grid row: a restatement of the reference's
`GaussianModel.extract_fields` (gs_renderer.py:218-294) and `gaussian_3d_coeff` (gs_renderer.py:64-83).

TEST INFRASTRUCTURE ONLY -- imported by tests/, never by the product (dreamgaussian_amd/).

PARITY PINNED: unlike the rasterizer, this path exists as Python in the reference, so the oracle is
checked against grids produced by the reference's own unmodified function run on the CPU
(tests/golden/reference_fields.npz, written by tests/golden/make_golden.py part 3;
tests/test_fields_oracle.py).

What is restated exactly (same fp32 operations in the same order, one rounding per operation, as
the chain of torch elementwise kernels the reference runs):
  * the opacity pre-filter `> 0.005`                                   gs_renderer.py:230-231
  * centre / scale normalisation to ~[-1,1]                            gs_renderer.py:237-243
  * rotation matrix, L = R diag(s), Sigma = L L^T (3-term sums in k order)  gs_renderer.py:85-117,126-131
  * the closed-form inverse with the +1e-24 determinant guard, the quadratic form, the
    `power > 0 -> weight 0` rule                                        gs_renderer.py:70-83
  * grid coordinates `linspace(-1,1,R)` split into num_blocks chunks, and block membership:
    strictly inside the chunk's box grown by block_size*relax_ratio   gs_renderer.py:251-270
What is NOT reproducible bit-for-bit: the order in which one grid point's contributions are
added (the reference adds 1024-wide `.sum(-1)` batches). The oracle therefore adds the fp32 terms
in float64; reference and HIP results must agree with it to fp32 summation error."""
from __future__ import annotations

import numpy as np
import torch

f32 = np.float32


def normalisation(xyz: np.ndarray, keep: np.ndarray):
    """centre (fp32 [3]), extent (fp32 scalar), scale (python float = 1.8 / extent) -- gs_renderer.py:237-240."""
    p = xyz[keep]
    mn, mx = p.min(0), p.max(0)
    center = (mn + mx) / f32(2)
    extent = (mx - mn).max()
    return center.astype(f32), f32(extent), 1.8 / float(extent)


def covariance6(stds: np.ndarray, rot_raw: np.ndarray) -> np.ndarray:
    """[n,6] upper triangle of R S S^T R^T in the reference's operation order (gs_renderer.py:85-131)."""
    r0, r1, r2, r3 = (rot_raw[:, i] for i in range(4))
    norm = np.sqrt(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3)
    q = rot_raw / norm[:, None]
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    one, two = f32(1), f32(2)
    R = np.empty((len(r), 3, 3), f32)
    R[:, 0, 0] = one - two * (y * y + z * z)
    R[:, 0, 1] = two * (x * y - r * z)
    R[:, 0, 2] = two * (x * z + r * y)
    R[:, 1, 0] = two * (x * y + r * z)
    R[:, 1, 1] = one - two * (x * x + z * z)
    R[:, 1, 2] = two * (y * z - r * x)
    R[:, 2, 0] = two * (x * z - r * y)
    R[:, 2, 1] = two * (y * z + r * x)
    R[:, 2, 2] = one - two * (x * x + y * y)
    L = R * stds[:, None, :]                                # R @ diag(s): one rounding per entry
    cov = np.empty((len(r), 6), f32)
    for n, (i, j) in enumerate(((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))):
        cov[:, n] = (L[:, i, 0] * L[:, j, 0] + L[:, i, 1] * L[:, j, 1]) + L[:, i, 2] * L[:, j, 2]
    return cov


def precision6(cov: np.ndarray) -> np.ndarray:
    """inv_a .. inv_f of gaussian_3d_coeff (gs_renderer.py:70-77), same order of operations."""
    a, b, c, d, e, f = (cov[:, i] for i in range(6))
    two = f32(2)
    det = a * d * f + two * e * c * b - e * e * a - c * c * d - b * b * f + f32(1e-24)
    inv_det = f32(1) / det
    return np.stack([(d * f - e * e) * inv_det, (e * c - b * f) * inv_det, (e * b - c * d) * inv_det,
                     (a * f - c * c) * inv_det, (b * c - e * a) * inv_det, (a * d - b * b) * inv_det], 1).astype(f32)


def block_boxes(resolution: int, num_blocks: int, relax_ratio: float):
    """Grid axis, chunk starts/lengths and the grown box [lo, hi] of every chunk (gs_renderer.py:221-225, 251-264)."""
    block_size = 2 / num_blocks
    split = resolution // num_blocks
    axis = torch.linspace(-1, 1, resolution)
    chunks = axis.split(split)
    grow = torch.tensor(block_size * relax_ratio, dtype=torch.float32)
    lo = torch.stack([c.amin() - grow for c in chunks]).numpy()
    hi = torch.stack([c.amax() + grow for c in chunks]).numpy()
    starts = np.cumsum([0] + [len(c) for c in chunks[:-1]])
    lens = np.array([len(c) for c in chunks])
    return axis.numpy(), starts, lens, lo, hi


def weights(d: np.ndarray, prec: np.ndarray) -> np.ndarray:
    """exp(power) for offsets d[...,3] and precisions prec[...,6]; power > 0 -> 0 (gs_renderer.py:79-83)."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    ia, ib, ic, id_, ie, if_ = (prec[..., i] for i in range(6))
    power = f32(-0.5) * (x * x * ia + y * y * id_ + z * z * if_) - x * y * ib - x * z * ic - y * z * ie
    power = np.where(power > 0, f32(-1e10), power)
    with np.errstate(under="ignore"):
        return np.exp(power.astype(f32))


def extract_fields(xyz, opacity, scaling, rotation_raw, resolution=128, num_blocks=16, relax_ratio=1.5,
                   acc=np.float64):
    """-> (occ [R,R,R] in `acc` precision, center fp32[3], scale float). Inputs are what the reference
    reads: `get_xyz`, `get_opacity` [N,1], `get_scaling` [N,3] (activated) and the RAW `_rotation` [N,4]."""
    xyz, opacity = np.asarray(xyz, f32), np.asarray(opacity, f32).reshape(-1)
    scaling, rotation_raw = np.asarray(scaling, f32), np.asarray(rotation_raw, f32)
    keep = opacity > f32(0.005)
    center, _, scale = normalisation(xyz, keep)
    s32 = f32(scale)
    p = (xyz[keep] - center) * s32
    prec = precision6(covariance6(scaling[keep] * s32, rotation_raw[keep]))
    op = opacity[keep]
    axis, starts, lens, lo, hi = block_boxes(resolution, num_blocks, relax_ratio)
    nb = len(starts)
    inx = (p[:, 0:1] > lo[None]) & (p[:, 0:1] < hi[None])      # [n, nb] membership per axis
    iny = (p[:, 1:2] > lo[None]) & (p[:, 1:2] < hi[None])
    inz = (p[:, 2:3] > lo[None]) & (p[:, 2:3] < hi[None])
    occ = np.zeros((resolution,) * 3, acc)
    for xi in range(nb):
        mx = inx[:, xi]
        if not mx.any():
            continue
        xs = axis[starts[xi]:starts[xi] + lens[xi]]
        for yi in range(nb):
            mxy = mx & iny[:, yi]
            if not mxy.any():
                continue
            ys = axis[starts[yi]:starts[yi] + lens[yi]]
            for zi in range(nb):
                m = mxy & inz[:, zi]
                if not m.any():
                    continue
                zs = axis[starts[zi]:starts[zi] + lens[zi]]
                pts = np.stack(np.meshgrid(xs, ys, zs, indexing="ij"), -1).reshape(-1, 3)        # [M,3]
                w = weights(pts[:, None, :] - p[m][None], prec[m][None])                          # [M,L]
                val = (op[m][None] * w).astype(acc).sum(-1)
                occ[starts[xi]:starts[xi] + lens[xi], starts[yi]:starts[yi] + lens[yi],
                    starts[zi]:starts[zi] + lens[zi]] = val.reshape(len(xs), len(ys), len(zs))
    return occ, center, scale
