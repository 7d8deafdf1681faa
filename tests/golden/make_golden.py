"""Regenerates tests/golden/*.npz.  Runs ONLY in the dev container (needs /root/reference).

Part 1 (reference_twins.npz) imports the reference's OWN Python -- sh_utils.py, cam_utils.py
and gs_renderer.py (with stub modules for its unrelated, uninstalled imports and `.cuda()`
mapped to the CPU) -- and records the outputs of the in-tree twins of the rasterizer's
per-Gaussian math and of the camera/settings assembly. These pin the oracle's conventions.

Part 3 (reference_fields.npz) runs the reference's OWN `GaussianModel.extract_fields`
(gs_renderer.py:218-294, unmodified, on the CPU) on small seeded scenes: the density grids the
HIP `extract_fields` and its oracle are pinned to.

Part 2 (oracle_render_small.npz) is NOT reference output: it is our float64 oracle's render
+ gradients of a small scene, committed so that the GPU tests also compare the HIP path with
a fixed vector (the reference ships no rasterizer source, tests or golden data: SURVEY 0.1/0.3).

    python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def import_reference():
    for name in ("plyfile", "mesh", "mesh_utils", "kiui"):
        m = types.ModuleType(name)
        sys.modules[name] = m
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    sys.modules["mesh"].Mesh = object
    sys.modules["mesh_utils"].decimate_mesh = sys.modules["mesh_utils"].clean_mesh = None
    sys.modules["kiui"].lo = lambda *a, **k: None
    torch.Tensor.cuda = lambda self, *a, **k: self
    _zeros = torch.zeros

    def zeros(*a, **k):
        if str(k.get("device", "")).startswith("cuda"):
            k.pop("device")
        return _zeros(*a, **k)
    torch.zeros = zeros
    sys.path.insert(0, REF)
    import sh_utils, cam_utils, gs_renderer   # noqa: E401
    return sh_utils, cam_utils, gs_renderer


def part1():
    sh_utils, cam_utils, gsr = import_reference()
    rs = np.random.RandomState(1234)
    out = {}
    # --- SH evaluation: reference layout is [..., C, K]; ours is [N, K, C] -----------------
    n = 64
    sh = rs.normal(0, 0.5, (n, 16, 3)).astype(np.float32)
    d = rs.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    out["sh_coeffs"] = sh
    out["sh_dirs"] = d
    for deg in range(4):
        res = sh_utils.eval_sh(deg, torch.from_numpy(sh).transpose(1, 2), torch.from_numpy(d))
        out[f"sh_eval_deg{deg}"] = res.numpy()
        out[f"sh_color_deg{deg}"] = torch.clamp_min(res + 0.5, 0.0).numpy()   # gs_renderer.py:793
    out["C0"] = np.float64(sh_utils.C0)
    out["rgb2sh"] = sh_utils.RGB2SH(torch.from_numpy(d)).numpy()
    # --- rotation / covariance twins ---------------------------------------------------------
    q = rs.normal(size=(n, 4)).astype(np.float32)
    s = np.exp(rs.uniform(-3, 0, (n, 3))).astype(np.float32)
    out["quat_raw"] = q
    out["scales"] = s
    out["build_rotation"] = gsr.build_rotation(torch.from_numpy(q)).numpy()          # normalises inside
    out["build_scaling_rotation"] = gsr.build_scaling_rotation(torch.from_numpy(s), torch.from_numpy(q)).numpy()
    gm = gsr.GaussianModel(3)
    for mod in (1.0, 0.7):
        out[f"covariance6_mod{mod}"] = gm.covariance_activation(torch.from_numpy(s), mod, torch.from_numpy(q)).numpy()
    # --- cameras ----------------------------------------------------------------------------------
    poses, views, projs, fulls, centers, Ps = [], [], [], [], [], []
    cams = [(0, 0, 2.0, 256, 256, 49.1), (-30, 45, 2.0, 800, 800, 49.1), (20, -120, 3.5, 250, 190, 60.0),
            (80, 170, 1.5, 512, 384, 35.0)]
    for el, az, r, W, H, fovy_deg in cams:
        pose = cam_utils.orbit_camera(el, az, r)
        fovy = np.deg2rad(fovy_deg)
        fovx = 2 * np.arctan(np.tan(fovy / 2) * W / H)
        mc = gsr.MiniCam(pose, W, H, fovy, fovx, 0.01, 100.0)
        poses.append(pose)
        views.append(mc.world_view_transform.numpy())
        projs.append(mc.projection_matrix.numpy())
        fulls.append(mc.full_proj_transform.numpy())
        centers.append(mc.camera_center.numpy())
        Ps.append(gsr.getProjectionMatrix(0.01, 100.0, fovx, fovy).numpy())
    out["cam_params"] = np.array(cams, dtype=np.float64)
    out["cam_pose"] = np.stack(poses)
    out["cam_world_view_transform"] = np.stack(views)
    out["cam_projection_matrix_T"] = np.stack(projs)
    out["cam_full_proj_transform"] = np.stack(fulls)
    out["cam_camera_center"] = np.stack(centers)
    out["cam_getProjectionMatrix"] = np.stack(Ps)
    np.savez_compressed(os.path.join(HERE, "reference_twins.npz"), **out)
    print("wrote reference_twins.npz:", sorted(out))
    return gsr


def part2():
    from oracle import gs_oracle as O
    N, deg, W, H = 300, 3, 56, 40
    sc = O.make_scene(N, deg, 7, "trained")
    S = O.make_settings(O.orbit_pose(-15.0, 40.0, 2.0), W, H, sh_degree=deg, dtype=torch.float64)
    g = torch.Generator().manual_seed(3)
    w = [torch.rand(3, H, W, generator=g, dtype=torch.float64), torch.rand(1, H, W, generator=g, dtype=torch.float64),
         torch.rand(1, H, W, generator=g, dtype=torch.float64)]
    t = {k: v.double().requires_grad_(True) for k, v in sc.items()}
    m2d = torch.zeros(N, 3, dtype=torch.float64, requires_grad=True)
    c, r, d, a, aux = O.rasterize(t["means3D"], m2d, t["opacities"], S, shs=t["shs"], scales=t["scales"],
                                  rotations=t["rotations"], return_aux=True)
    torch.autograd.backward([c, d, a], w)
    out = {f"in_{k}": v.numpy() for k, v in sc.items()}
    out.update(pose=O.orbit_pose(-15.0, 40.0, 2.0), W=W, H=H, deg=deg,
               w_color=w[0].numpy(), w_depth=w[1].numpy(), w_alpha=w[2].numpy(),
               color=c.detach().numpy(), radii=r.numpy(), depth=d.detach().numpy(), alpha=a.detach().numpy(),
               grad_means2D=m2d.grad.numpy(), fragile_pixels=aux["fragile_pixels"].numpy(),
               fragile_gaussians=aux["fragile_gaussians"].numpy())
    out.update({f"grad_{k}": v.grad.numpy() for k, v in t.items()})
    np.savez_compressed(os.path.join(HERE, "oracle_render_small.npz"), **out)
    print("wrote oracle_render_small.npz")


def fields_scene(N, seed):
    """Raw GaussianModel parameters (what the reference stores): anisotropic, randomly rotated,
    un-normalised quaternions, 6% of the opacities below the 0.005 pre-filter."""
    from dreamgaussian_amd import synthetic
    sc = synthetic.make_scene(N, 0, seed, "trained")
    rs = np.random.RandomState(seed + 100)
    op = sc["opacities"].numpy().copy()
    low = rs.rand(N) < 0.06
    op[low, 0] = rs.uniform(1e-4, 0.0099, int(low.sum()))
    q = sc["rotations"].numpy() * rs.uniform(0.4, 2.5, (N, 1))
    return dict(xyz=sc["means3D"].numpy().astype(np.float32),
                opacity_raw=np.log(op / (1 - op)).astype(np.float32),
                scaling_raw=np.log(sc["scales"].numpy()).astype(np.float32),
                rotation_raw=q.astype(np.float32))


def part3(gsr):
    import time
    out = {}
    cases = [("r32", 700, 5, 32, 16, 1.5), ("r64", 3000, 6, 64, 16, 1.5), ("r48nb8", 1500, 7, 48, 8, 1.0),
             ("r128", 6000, 8, 128, 16, 1.5)]
    out["cases"] = np.array([c[0] for c in cases])
    for name, N, seed, R, nb, relax in cases:
        raw = fields_scene(N, seed)
        gm = gsr.GaussianModel(0)
        gm._xyz = torch.from_numpy(raw["xyz"])
        gm._opacity = torch.from_numpy(raw["opacity_raw"])
        gm._scaling = torch.from_numpy(raw["scaling_raw"])
        gm._rotation = torch.from_numpy(raw["rotation_raw"])
        t0 = time.time()
        occ = gm.extract_fields(resolution=R, num_blocks=nb, relax_ratio=relax)
        dt = time.time() - t0
        # the exact fp32 activated tensors the reference fed its field evaluation
        out[f"{name}_xyz"] = raw["xyz"]
        out[f"{name}_opacity"] = gm.get_opacity.numpy()
        out[f"{name}_scaling"] = gm.get_scaling.numpy()
        out[f"{name}_rotation_raw"] = raw["rotation_raw"]
        out[f"{name}_params"] = np.array([R, nb, relax], dtype=np.float64)
        out[f"{name}_center"] = gm.center.numpy()
        out[f"{name}_scale"] = np.float64(gm.scale)
        o = occ.numpy()
        if R <= 64:
            out[f"{name}_occ"] = o
        else:                        # 8 MB: keep a strided sample and slab sums instead
            out[f"{name}_occ_stride3"] = o[::3, ::3, ::3].copy()
            out[f"{name}_occ_slab_sums"] = o.astype(np.float64).sum(axis=(1, 2))
            out[f"{name}_occ_max"] = np.float64(o.max())
        print(f"  {name}: N={N} R={R} nb={nb} relax={relax}  {dt:.1f}s  max={o.max():.4f} nonzero={np.count_nonzero(o) / o.size:.3f}")
    np.savez_compressed(os.path.join(HERE, "reference_fields.npz"), **out)
    print("wrote reference_fields.npz")


if __name__ == "__main__":
    if "--fields-only" not in sys.argv:
        part2()      # before part1: part1 monkey-patches torch
    gsr = part1() if "--fields-only" not in sys.argv else import_reference()[2]
    part3(gsr)
