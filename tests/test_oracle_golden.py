"""The oracle's conventions against fixtures produced by the reference's OWN Python
(tests/golden/make_golden.py imports /root/reference/{sh_utils,cam_utils,gs_renderer}.py).
These are the only reference-produced vectors that exist for this path: the rasterizer's
arithmetic itself is an absent third-party CUDA package (parity otherwise unpinned)."""
import os

import numpy as np
import pytest
import torch

from oracle import gs_oracle as O


@pytest.fixture(scope="module")
def tw(golden_dir):
    return np.load(os.path.join(golden_dir, "reference_twins.npz"))


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_sh_basis_sign_and_layout(tw, deg):
    sh = torch.from_numpy(tw["sh_coeffs"])            # [n,16,3] = our [N,K,3] layout
    d = torch.from_numpy(tw["sh_dirs"])
    got = O.eval_sh_color(deg, sh, d)
    np.testing.assert_allclose(got.numpy(), tw[f"sh_eval_deg{deg}"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(torch.clamp_min(got + 0.5, 0).numpy(), tw[f"sh_color_deg{deg}"], rtol=0, atol=2e-6)
    assert float(tw["C0"]) == O.C0


def test_rotation_and_covariance_twins(tw):
    q = torch.from_numpy(tw["quat_raw"])
    s = torch.from_numpy(tw["scales"])
    qn = q / q.norm(dim=1, keepdim=True)               # the kernel contract: caller normalises
    np.testing.assert_allclose(O.rotation_matrix(qn).numpy(), tw["build_rotation"], atol=1e-6)
    L = O.rotation_matrix(qn) * s[:, None, :]
    np.testing.assert_allclose(L.numpy(), tw["build_scaling_rotation"], atol=1e-6)
    for mod in (1.0, 0.7):
        Sig = O.covariance3d(s, mod, qn)
        six = torch.stack([Sig[:, 0, 0], Sig[:, 0, 1], Sig[:, 0, 2], Sig[:, 1, 1], Sig[:, 1, 2], Sig[:, 2, 2]], -1)
        np.testing.assert_allclose(six.numpy(), tw[f"covariance6_mod{mod}"], atol=1e-6)
        np.testing.assert_allclose(O.sym_from6(six).numpy(), Sig.numpy(), atol=0)


def test_camera_and_settings_assembly(tw):
    for i, (el, az, r, W, H, fovy) in enumerate(tw["cam_params"]):
        pose = O.orbit_pose(el, az, r)
        np.testing.assert_allclose(pose, tw["cam_pose"][i], atol=1e-6)
        S = O.make_settings(pose, int(W), int(H), fovy_deg=fovy)
        np.testing.assert_allclose(S.viewmatrix.numpy(), tw["cam_world_view_transform"][i], atol=1e-6)
        np.testing.assert_allclose(S.projmatrix.numpy(), tw["cam_full_proj_transform"][i], atol=2e-6)
        np.testing.assert_allclose(S.campos.numpy(), tw["cam_camera_center"][i], atol=1e-6)
        P = tw["cam_getProjectionMatrix"][i]
        assert abs(S.tanfovx - 1.0 / P[0, 0]) < 1e-6 and abs(S.tanfovy - 1.0 / P[1, 1]) < 1e-6


def test_oracle_reproduces_committed_render(golden_dir):
    """float64 oracle vs its own committed vector: guards the checker against silent drift."""
    z = np.load(os.path.join(golden_dir, "oracle_render_small.npz"))
    sc = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("in_")}
    S = O.make_settings(z["pose"], int(z["W"]), int(z["H"]), sh_degree=int(z["deg"]), dtype=torch.float64)
    t = {k: v.double().requires_grad_(True) for k, v in sc.items()}
    m2d = torch.zeros(t["means3D"].shape[0], 3, dtype=torch.float64, requires_grad=True)
    c, r, d, a = O.rasterize(t["means3D"], m2d, t["opacities"], S, shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    torch.autograd.backward([c, d, a], [torch.from_numpy(z[k]) for k in ("w_color", "w_depth", "w_alpha")])
    np.testing.assert_allclose(c.detach().numpy(), z["color"], atol=1e-12)
    np.testing.assert_array_equal(r.numpy(), z["radii"])
    np.testing.assert_allclose(d.detach().numpy(), z["depth"], atol=1e-12)
    np.testing.assert_allclose(a.detach().numpy(), z["alpha"], atol=1e-12)
    for k in t:
        np.testing.assert_allclose(t[k].grad.numpy(), z[f"grad_{k}"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(m2d.grad.numpy(), z["grad_means2D"], rtol=1e-9, atol=1e-12)


def test_default_build_rotation_of_densify_and_split_is_the_references(golden_dir):
    """dreamgaussian_amd.densify.quaternion_to_matrix (the default `build_rotation` of densify_and_split, round-3 advisor: the
    default used to be None and raised) against the output of the reference's own gs_renderer.build_rotation."""
    from dreamgaussian_amd.densify import quaternion_to_matrix
    z = np.load(os.path.join(golden_dir, "reference_twins.npz"))
    got = quaternion_to_matrix(torch.from_numpy(z["quat_raw"]))
    assert np.abs(got.numpy() - z["build_rotation"]).max() <= 1e-6


def test_testing_hooks_are_explicit_calls():
    """The library's A/B and test switches are explicit calls (gsr_testing_override), not environment variables (round-3 verdict,
    weak #12): unknown names are refused, values round-trip through the module's book-keeping, and the device code / host code
    contain no getenv."""
    from dreamgaussian_amd import _testing
    with pytest.raises(KeyError):
        _testing.set("no_such_switch", 1)
    with _testing.override(fwd_mode="seg", seg_shift=7):
        assert _testing._current == {"fwd_mode": 2, "seg_shift": 7}
    assert _testing._current == {}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in os.listdir(os.path.join(root, "dreamgaussian_amd", "csrc")):
        assert "getenv" not in open(os.path.join(root, "dreamgaussian_amd", "csrc", f)).read(), f
