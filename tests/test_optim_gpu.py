"""Fused Adam and the densification gathers against torch's own code on the same data:
torch.optim.Adam (the optimiser GaussianModel.training_setup builds, gs_renderer.py:356-374) and boolean-mask
indexing (prune_points / _prune_optimizer, gs_renderer.py:479-511) are the checkers."""
import types

import pytest
import torch

import dreamgaussian_amd as D

pytestmark = pytest.mark.gpu

GROUPS = [("xyz", (3,), 1e-3), ("f_dc", (1, 3), 1e-2), ("f_rest", (15, 3), 5e-4), ("opacity", (1,), 5e-2),
          ("scaling", (3,), 5e-3), ("rotation", (4,), 5e-3)]


def _make(N, dev, cls, seed=0):
    g = torch.Generator().manual_seed(seed)
    params = [torch.nn.Parameter(torch.randn((N,) + shp, generator=g).to(dev)) for _, shp, _ in GROUPS]
    groups = [{"params": [p], "lr": lr, "name": name} for p, (name, _, lr) in zip(params, GROUPS)]
    return params, cls(groups, lr=0.0, eps=1e-15)            # gs_renderer.py:373


@pytest.mark.parametrize("N", [1, 777, 50_000])
def test_fused_adam_matches_torch_adam(gpu, N):
    pa, oa = _make(N, gpu, D.FusedAdam)
    pb, ob = _make(N, gpu, torch.optim.Adam)
    g = torch.Generator().manual_seed(1)
    for step in range(6):
        for x, y in zip(pa, pb):
            gr = (torch.randn(x.shape, generator=g) * (10.0 ** (step - 3))).to(gpu)      # gradients over six decades
            x.grad, y.grad = gr.clone(), gr.clone()
        for grp_a, grp_b in zip(oa.param_groups, ob.param_groups):                        # update_learning_rate (gs_renderer.py:376-382)
            if grp_a["name"] == "xyz":
                grp_a["lr"] = grp_b["lr"] = 1e-3 * 0.9 ** step
        oa.step(); ob.step()
    for x, y, (name, _, _) in zip(pa, pb, GROUPS):
        sa, sb = oa.state[x], ob.state[y]
        assert float(sa["step"]) == float(sb["step"]) == 6
        for a, b, what in ((x, y, "param"), (sa["exp_avg"], sb["exp_avg"], "exp_avg"), (sa["exp_avg_sq"], sb["exp_avg_sq"], "exp_avg_sq")):
            err = (a.detach() - b.detach()).abs().max().item()
            assert err == 0.0, (name, what, err)          # the same operations in the same order as torch's foreach path: bit-identical


def test_mask_compaction_and_multi_tensor_gather(gpu):
    g = torch.Generator().manual_seed(2)
    for N in (1, 63, 1024, 1025, 200_003):
        mask = (torch.rand(N, generator=g) < 0.37).to(gpu)
        idx, count = D.compact_mask(mask)
        ref = mask.nonzero().squeeze(1)
        assert count == ref.numel() and torch.equal(idx.long(), ref)
        tensors = [torch.randn(N, 3, generator=g).to(gpu), torch.randn(N, 15, 3, generator=g).to(gpu),
                   torch.randn(N, generator=g).to(gpu), torch.randn(N, 1, generator=g).to(gpu)]
        outs = D.gather_rows(idx, tensors)
        for t, o in zip(tensors, outs):
            assert torch.equal(o, t[mask])
    idx, count = D.compact_mask(torch.zeros(500, dtype=torch.bool, device=gpu))
    assert count == 0 and idx.numel() == 0
    assert [tuple(o.shape) for o in D.gather_rows(idx, [torch.zeros(500, 4, device=gpu)])] == [(0, 4)]


def test_prune_points_equals_reference_algorithm(gpu):
    """dreamgaussian_amd.prune_points on an object with GaussianModel's attributes against the reference's own
    algorithm (boolean-mask indexing of parameters, Adam moments and accumulators, gs_renderer.py:479-511)."""
    N = 4096
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")

    def build(cls):
        params, opt = _make(N, gpu, cls, seed=5)
        for p in params:
            p.grad = torch.ones_like(p) * 0.1
        opt.step()                                              # populate exp_avg / exp_avg_sq
        ga = torch.Generator().manual_seed(7)
        gm = types.SimpleNamespace(optimizer=opt, xyz_gradient_accum=torch.rand(N, 1, generator=ga).to(gpu),
                                   denom=torch.rand(N, 1, generator=ga).to(gpu), max_radii2D=torch.rand(N, generator=ga).to(gpu))
        for n, p in zip(names, params):
            setattr(gm, n, p)
        return gm
    a, b = build(D.FusedAdam), build(D.FusedAdam)          # identical states; what is compared is the pruning
    mask = (torch.rand(N, generator=torch.Generator().manual_seed(6)) < 0.3).to(gpu)
    D.prune_points(a, mask)
    keep = ~mask                                                # the reference's algorithm on b
    for grp in b.optimizer.param_groups:
        p = grp["params"][0]
        st = b.optimizer.state[p]
        st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][keep], st["exp_avg_sq"][keep]
        del b.optimizer.state[p]
        grp["params"][0] = torch.nn.Parameter(p[keep].requires_grad_(True))
        b.optimizer.state[grp["params"][0]] = st
    n_keep = int(keep.sum())
    for ga, gb, n in zip(a.optimizer.param_groups, b.optimizer.param_groups, names):
        pa, pb = ga["params"][0], gb["params"][0]
        assert pa is getattr(a, n) and pa.requires_grad and pa.shape[0] == n_keep
        assert torch.equal(pa.detach(), pb.detach())
        assert torch.equal(a.optimizer.state[pa]["exp_avg"], b.optimizer.state[pb]["exp_avg"])
        assert torch.equal(a.optimizer.state[pa]["exp_avg_sq"], b.optimizer.state[pb]["exp_avg_sq"])
    assert torch.equal(a.xyz_gradient_accum, b.xyz_gradient_accum[keep]) and torch.equal(a.denom, b.denom[keep])
    assert torch.equal(a.max_radii2D, b.max_radii2D[keep])
    for p in (g["params"][0] for g in a.optimizer.param_groups):   # the optimiser keeps stepping on the new tensors
        p.grad = torch.ones_like(p)
    a.optimizer.step()


def test_gather_rows_with_an_empty_feature_tensor(gpu):
    """`_features_rest` is [N,0,3] at sh_degree 0 (configs/image.yaml): nothing to move, shape preserved."""
    g = torch.Generator().manual_seed(5)
    a, b = torch.rand(100, 3, generator=g).to(gpu), torch.zeros(100, 0, 3, device=gpu)
    idx, count = D.compact_mask(torch.rand(100, generator=g).to(gpu) > 0.5)
    oa, ob = D.gather_rows(idx, [a, b])
    assert ob.shape == (count, 0, 3) and torch.equal(oa, a[idx.long()])
