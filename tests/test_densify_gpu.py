"""add_densification_stats (C ABI gsr_densify_stats) against the reference's own three torch lines
(main.py:280, gs_renderer.py:626-627) executed on the CPU -- pinned: the checker IS the reference code."""
import numpy as np
import pytest
import torch

import dreamgaussian_amd as D

pytestmark = pytest.mark.gpu


def reference_lines(viewspace_grad, radii, xyz_gradient_accum, denom, max_radii2D):
    visibility_filter = radii > 0                                                                  # gs_renderer.py:813
    max_radii2D[visibility_filter] = torch.max(max_radii2D[visibility_filter], radii[visibility_filter])   # main.py:280
    xyz_gradient_accum[visibility_filter] += torch.norm(viewspace_grad[visibility_filter, :2], dim=-1, keepdim=True)  # gs_renderer.py:626
    denom[visibility_filter] += 1                                                                  # gs_renderer.py:627


@pytest.mark.parametrize("N", [1, 257, 50_000])
def test_three_steps_match_the_reference_lines(gpu, N):
    g = torch.Generator().manual_seed(N)
    acc, den, mr = torch.zeros(N, 1), torch.zeros(N, 1), torch.zeros(N)
    acc_d, den_d, mr_d = acc.to(gpu), den.to(gpu), mr.to(gpu)
    for step in range(3):
        grad = torch.randn(N, 3, generator=g) * 10.0 ** float(torch.randint(-6, 2, (1,), generator=g))
        radii = torch.randint(-1, 40, (N,), generator=g, dtype=torch.int32).clamp_min(0)
        radii[torch.rand(N, generator=g) < 0.3] = 0
        reference_lines(grad, radii, acc, den, mr)
        D.add_densification_stats(grad.to(gpu), radii.to(gpu), acc_d, den_d, mr_d)
    assert torch.equal(den_d.cpu(), den) and torch.equal(mr_d.cpu(), mr)
    np.testing.assert_allclose(acc_d.cpu().numpy(), acc.numpy(), rtol=3e-7, atol=0)


def test_with_the_rasterizer_outputs_and_errors(gpu):
    from oracle import gs_oracle as O
    from util import settings_to
    sc = O.make_scene(3000, 0, 2, "trained")
    S = O.make_settings(O.orbit_pose(0.0, 20.0, 2.0), 96, 96, sh_degree=0)
    t = {k: v.to(gpu).requires_grad_(True) for k, v in sc.items()}
    holder = torch.zeros(3000, 3, device=gpu, requires_grad=True)
    color, radii, depth, alpha = D.GaussianRasterizer(raster_settings=settings_to(S, gpu))(
        means3D=t["means3D"], means2D=holder, shs=t["shs"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    color.sum().backward()
    acc, den, mr = torch.zeros(3000, 1, device=gpu), torch.zeros(3000, 1, device=gpu), torch.zeros(3000, device=gpu)
    D.add_densification_stats(holder.grad, radii, acc, den, mr)
    a2, d2, m2 = torch.zeros(3000, 1), torch.zeros(3000, 1), torch.zeros(3000)
    reference_lines(holder.grad.cpu(), radii.cpu(), a2, d2, m2)
    assert torch.equal(den.cpu(), d2) and torch.equal(mr.cpu(), m2)
    np.testing.assert_allclose(acc.cpu().numpy(), a2.numpy(), rtol=3e-7)
    assert (den.cpu().squeeze(1) > 0).equal(radii.cpu() > 0)
    with pytest.raises(RuntimeError, match="GPU only"):
        D.add_densification_stats(torch.zeros(4, 3), torch.zeros(4, dtype=torch.int32), torch.zeros(4, 1), torch.zeros(4, 1), torch.zeros(4))
    with pytest.raises(RuntimeError, match="contiguous float32"):
        D.add_densification_stats(holder.grad, radii, acc.double(), den, mr)
