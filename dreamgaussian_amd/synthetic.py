"""Synthetic inputs for tests and bench.py: the scenes and cameras of SURVEY 8(d).

Host-side input generators only (numpy); no rendering arithmetic lives here. They restate
how the reference creates its inputs: the random init blob of Renderer.initialize
(gs_renderer.py:689-709) with create_from_pcd's 3-NN scale rule (gs_renderer.py:341-346),
orbit cameras (cam_utils.py:21-63) and MiniCam + Renderer.render's settings assembly
(gs_renderer.py:629-671, 742-758)."""
from __future__ import annotations

import math

import numpy as np
import torch

from .rasterizer import GaussianRasterizationSettings

SH_C0 = 0.28209479177387814


def look_at_opengl(campos):
    """cam_utils.py:21-41 (opengl=True, target=0)."""
    def nrm(v):
        return v / np.sqrt(max(float(np.sum(v * v)), 1e-20))
    fwd = nrm(campos.astype(np.float64))
    up = np.array([0, 1, 0], dtype=np.float64)
    right = nrm(np.cross(up, fwd))
    up = nrm(np.cross(fwd, right))
    return np.stack([right, up, fwd], axis=1)


def orbit_pose(elevation, azimuth, radius):
    """cam_utils.py:44-63 (degrees, target at the origin) -> c2w [4,4] float32."""
    el, az = np.deg2rad(elevation), np.deg2rad(azimuth)
    campos = np.array([radius * np.cos(el) * np.sin(az), -radius * np.sin(el),
                       radius * np.cos(el) * np.cos(az)])
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = look_at_opengl(campos)
    T[:3, 3] = campos
    return T


def make_settings(c2w, W, H, fovy_deg=49.1, znear=0.01, zfar=100.0, sh_degree=0, bg=(1, 1, 1),
                  scale_modifier=1.0, dtype=torch.float32, device="cpu") -> GaussianRasterizationSettings:
    """MiniCam + Renderer.render's settings assembly (gs_renderer.py:645-671, 742-758)."""
    fovy = np.deg2rad(fovy_deg)
    fovx = 2 * np.arctan(np.tan(fovy / 2) * W / H)
    w2c = np.linalg.inv(c2w)
    w2c[1:3, :3] *= -1
    w2c[:3, 3] *= -1
    view = torch.tensor(w2c).transpose(0, 1).to(torch.float32)
    P = torch.zeros(4, 4)
    P[0, 0] = 1 / math.tan(fovx / 2)
    P[1, 1] = 1 / math.tan(fovy / 2)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    proj = view @ P.transpose(0, 1)
    campos = -torch.tensor(c2w[:3, 3]).to(torch.float32)
    to = lambda t: t.to(dtype=dtype, device=device)
    return GaussianRasterizationSettings(H, W, math.tan(fovx * 0.5), math.tan(fovy * 0.5),
                                         to(torch.tensor(bg, dtype=torch.float32)), scale_modifier,
                                         to(view), to(proj), sh_degree, to(campos), False, False)


def nn3_mean_sqdist(xyz: np.ndarray) -> np.ndarray:
    """Host stand-in for distCUDA2 when building scenes (exact 3-NN mean squared distance)."""
    from scipy.spatial import cKDTree
    d, _ = cKDTree(xyz).query(xyz, k=4, workers=-1)      # all host cores: 8 s -> ~1 s at 1M points
    return (d[:, 1:] ** 2).mean(1)


def make_scene(N, sh_degree=0, seed=0, kind="blob"):
    """Activated tensors exactly as Renderer.render hands them to the rasterizer
    (gs_renderer.py:762-797): means3D [N,3], shs [N,K,3], opacities [N,1], scales [N,3],
    rotations [N,4].  kind 'blob' = the reference's init; 'trained' = random rotations,
    anisotropic scales, opacity U[0.05,0.95]."""
    rs = np.random.RandomState(seed)
    phis = rs.random_sample(N) * 2 * np.pi
    costheta = rs.random_sample(N) * 2 - 1
    thetas = np.arccos(costheta)
    mu = rs.random_sample(N)
    r = 0.5 * np.cbrt(mu)
    xyz = np.stack([r * np.sin(thetas) * np.cos(phis), r * np.sin(thetas) * np.sin(phis),
                    r * np.cos(thetas)], 1).astype(np.float32)
    K = (sh_degree + 1) ** 2
    sh = np.zeros((N, K, 3), np.float32)
    sh[:, 0] = rs.random_sample((N, 3)) / 255.0  # SH2RGB->RGB2SH round trip (gs_renderer.py:705-707,334)
    if K > 1:
        sh[:, 1:] = rs.normal(0, 0.1, (N, K - 1, 3))
    if N >= 4:
        d2 = np.maximum(nn3_mean_sqdist(xyz.astype(np.float64)), 1e-7).astype(np.float32)
    else:
        d2 = np.full(N, 1e-2, np.float32)
    sigma = np.sqrt(d2)
    if kind == "blob":
        scales = np.repeat(sigma[:, None], 3, 1)
        rots = np.zeros((N, 4), np.float32)
        rots[:, 0] = 1
        opac = np.full((N, 1), 0.1, np.float32)
    elif kind == "trained":
        scales = sigma[:, None] * np.exp(rs.uniform(np.log(0.3), np.log(3.0), (N, 3)))
        q = rs.normal(size=(N, 4))
        rots = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
        opac = rs.uniform(0.05, 0.95, (N, 1)).astype(np.float32)
        sh[:, 0] = (rs.random_sample((N, 3)) - 0.5) / SH_C0
    else:
        raise ValueError(kind)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return dict(means3D=t(xyz), shs=t(sh), opacities=t(opac), scales=t(scales), rotations=t(rots))
