"""Self-pinning tests of the CPU oracle: invariants of the published algorithm
that hold whatever the absent CUDA extension's exact constants are."""
import numpy as np
import pytest
import torch

from oracle import gs_oracle as O
from util import run_oracle, weights_for


def scene(N=400, deg=1, kind="trained", W=64, H=48, seed=0, **kw):
    sc = O.make_scene(N, deg, seed, kind)
    S = O.make_settings(O.orbit_pose(kw.get("el", -10.0), kw.get("az", 25.0), 2.0), W, H, sh_degree=deg,
                        bg=kw.get("bg", (1, 1, 1)))
    return sc, S


def test_alpha_is_one_minus_final_T_and_bg_independent():
    sc, S = scene()
    out_w, _, aux = run_oracle(sc, S)
    assert (out_w[3][0] - (1 - aux["T_final"])).abs().max() < 1e-6
    out_b, _, aux_b = run_oracle(sc, S._replace(bg=torch.zeros(3)))
    # colour minus T*bg does not depend on the background
    assert (out_w[0] - aux["T_final"][None] - out_b[0]).abs().max() < 1e-6
    assert torch.equal(out_w[2], out_b[2]) and torch.equal(out_w[3], out_b[3])


def test_float64_and_float32_agree():
    sc, S = scene(deg=3)
    w = weights_for(48, 64)
    o32, g32, _ = run_oracle(sc, S, w, torch.float32)
    o64, g64, _ = run_oracle(sc, S, w, torch.float64)
    for i in (0, 2, 3):
        assert (o32[i].double() - o64[i]).abs().max() < 5e-5
    for k in g64:
        assert (g32[k].double() - g64[k]).abs().max() <= 2e-3 * g64[k].abs().max() + 1e-9, k


def test_fragile_decisions_explain_every_fp32_vs_fp64_difference():
    """The oracle flags (pixel, Gaussian) pairs whose discrete tests sit within fp32 noise of their
    threshold. Outside the flagged pixels / Gaussians the float32 and float64 oracles must agree
    strictly -- the same rule the GPU parity tests apply (this scene has one real flip)."""
    from util import assert_forward_close, assert_grads_close, grad_floors
    sc = O.make_scene(2000, 2, 0, "blob")
    S = O.make_settings(O.orbit_pose(0.0, 180.0, 2.0), 320, 96, sh_degree=2)
    w = weights_for(96, 320)
    o64, g64, aux = run_oracle(sc, S, w, torch.float64)
    o32, g32, _ = run_oracle(sc, S, w, torch.float32)
    assert 0 < int(aux["fragile_pixels"].sum()) < 50 and int(aux["fragile_gaussians"].sum()) < 50
    assert_forward_close(o32, o64, aux)
    assert_grads_close(g32, g64, aux, floors=grad_floors(sc, g64))


def test_culled_gaussians_have_zero_radius_and_zero_grads():
    sc, S = scene(N=300)
    sc["means3D"][:50, 2] += 5.0          # behind the camera at z=+2 looking down -z(world)
    w = weights_for(48, 64)
    out, g, aux = run_oracle(sc, S, w)
    radii = out[1]
    assert (radii[:50] == 0).all() and (radii[50:] > 0).any()
    for k in ("means3D", "scales", "rotations", "opacities", "shs", "means2D"):
        assert g[k][:50].abs().max() == 0, k


def test_permutation_invariance():
    sc, S = scene(N=200, deg=0)
    perm = torch.randperm(200, generator=torch.Generator().manual_seed(0))
    out1, _, _ = run_oracle(sc, S)
    out2, _, _ = run_oracle({k: v[perm] for k, v in sc.items()}, S)
    assert (out1[0] - out2[0]).abs().max() < 1e-5
    assert torch.equal(out1[1][perm], out2[1])


def test_depth_ties_resolve_in_index_order():
    # two coincident Gaussians with different colours: the lower index composites first
    S = O.make_settings(O.orbit_pose(0, 0, 2.0), 32, 32, sh_degree=0, bg=(0, 0, 0))
    m = torch.zeros(2, 3)
    sh = torch.zeros(2, 1, 3)
    sh[0, 0, 0] = 0.5 / O.C0       # red (+0.5 offset -> 1.0)
    sh[1, 0, 2] = 0.5 / O.C0       # blue
    sc = dict(means3D=m, shs=sh, opacities=torch.full((2, 1), 0.9), scales=torch.full((2, 3), 0.1),
              rotations=torch.tensor([[1.0, 0, 0, 0]] * 2))
    out, _, _ = run_oracle(sc, S)
    c = out[0][:, 16, 16]
    assert c[0] > c[2] > 0         # red in front: weight .9 vs .09


def test_empty_and_tiny_inputs():
    S = O.make_settings(O.orbit_pose(0, 0, 2.0), 20, 12, sh_degree=0)
    z = torch.zeros
    c, r, d, a = O.rasterize(z(0, 3), None, z(0, 1), S, shs=z(0, 1, 3), scales=z(0, 3), rotations=z(0, 4))
    assert c.shape == (3, 12, 20) and (c == 1).all() and r.numel() == 0 and (a == 0).all() and (d == 0).all()
    sc = O.make_scene(1, 0, 0, "blob")
    c, r, d, a = O.rasterize(sc["means3D"], None, sc["opacities"], S, shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
    assert r.shape == (1,) and torch.isfinite(c).all()


def test_means2D_grad_is_ndc_scaled_pixel_grad():
    """means2D.grad[:, :2] must be dL/d(pixel mean) * 0.5*(W,H) (consumer
    gs_renderer.py:625-627): shifting every mean by eps in NDC changes the loss accordingly."""
    sc, S = scene(N=150, deg=0, W=40, H=24)
    w = [x.double() for x in weights_for(24, 40)]
    _, g, _ = run_oracle(sc, S, w, torch.float64)
    S64 = O.Settings(*[x.double() if torch.is_tensor(x) else x for x in S])
    def loss(eps):
        t = {k: v.double() for k, v in sc.items()}
        m2d = torch.zeros(150, 3, dtype=torch.float64)
        m2d[:, 0] = eps
        c, r, d, a = O.rasterize(t["means3D"], m2d, t["opacities"], S64, shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
        return float((w[0] * c).sum() + (w[1] * d).sum() + (w[2] * a).sum())
    eps = 1e-6
    fd = (loss(eps) - loss(-eps)) / (2 * eps)
    assert abs(fd - g["means2D"][:, 0].sum().item()) < 1e-3 * max(1.0, abs(fd))


def test_tile_subset_matches_full_render_on_those_tiles():
    sc, S = scene(N=300, deg=0, W=64, H=48)
    full, _, aux = run_oracle(sc, S)
    t = {k: v for k, v in sc.items()}
    c, r, d, a = O.rasterize(t["means3D"], None, t["opacities"], S, shs=t["shs"], scales=t["scales"], rotations=t["rotations"], tiles=[5])
    y0, x0 = (5 // 4) * 16, (5 % 4) * 16
    assert torch.equal(c[:, y0:y0 + 16, x0:x0 + 16], full[0][:, y0:y0 + 16, x0:x0 + 16])
    assert (a[0, :16, :16] == 0).all()


def test_dist2_oracle_definition():
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3], [10, 10, 10]], np.float64)
    got = O.nn3_mean_sqdist(pts)
    assert abs(got[0] - (1 + 4 + 9) / 3) < 1e-12
