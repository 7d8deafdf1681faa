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


def test_prune_points_in_z_order_is_the_reference_result_up_to_a_row_permutation(gpu):
    """`prune_points(..., zorder=True)`: the same surviving Gaussians, each with its own Adam moments and densification statistics,
    as the reference's boolean-mask indexing leaves them -- in the order of a Z-order curve through their positions."""
    N = 5000
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
    params, opt = _make(N, gpu, D.FusedAdam, seed=11)
    for p_ in params:
        p_.grad = torch.ones_like(p_) * 0.1
    opt.step()
    ga = torch.Generator().manual_seed(7)
    gm = types.SimpleNamespace(optimizer=opt, xyz_gradient_accum=torch.rand(N, 1, generator=ga).to(gpu),
                               denom=torch.rand(N, 1, generator=ga).to(gpu), max_radii2D=torch.rand(N, generator=ga).to(gpu))
    for n, p_ in zip(names, params):
        setattr(gm, n, p_)
    before = {n: getattr(gm, n).detach().clone() for n in names}
    mom = {n: (opt.state[getattr(gm, n)]["exp_avg"].clone(), opt.state[getattr(gm, n)]["exp_avg_sq"].clone()) for n in names}
    acc = (gm.xyz_gradient_accum.clone(), gm.denom.clone(), gm.max_radii2D.clone())
    mask = (torch.rand(N, generator=torch.Generator().manual_seed(6)) < 0.3).to(gpu)
    D.prune_points(gm, mask, zorder=True)
    keep = (~mask).nonzero().squeeze(1)
    # which original row sits where: the positions are distinct random numbers
    xyz_new = gm._xyz.detach()
    perm = D.morton_order(before["_xyz"][keep]).long()
    src = keep[perm]
    assert xyz_new.shape[0] == keep.numel() and torch.equal(xyz_new, before["_xyz"][src])
    code_sorted = D.morton_order(xyz_new).long()
    assert torch.equal(code_sorted, torch.arange(xyz_new.shape[0], device=gpu))      # the result IS in Z-order (stable sort: identity)
    for n in names:
        p_ = getattr(gm, n)
        assert p_.requires_grad and torch.equal(p_.detach(), before[n][src]), n
        st = gm.optimizer.state[p_]
        assert torch.equal(st["exp_avg"], mom[n][0][src]) and torch.equal(st["exp_avg_sq"], mom[n][1][src]), n
    assert torch.equal(gm.xyz_gradient_accum, acc[0][src]) and torch.equal(gm.denom, acc[1][src]) and torch.equal(gm.max_radii2D, acc[2][src])
    for p_ in (g["params"][0] for g in gm.optimizer.param_groups):
        p_.grad = torch.ones_like(p_)
    gm.optimizer.step()


def test_gather_rows_with_an_empty_feature_tensor(gpu):
    """`_features_rest` is [N,0,3] at sh_degree 0 (configs/image.yaml): nothing to move, shape preserved."""
    g = torch.Generator().manual_seed(5)
    a, b = torch.rand(100, 3, generator=g).to(gpu), torch.zeros(100, 0, 3, device=gpu)
    idx, count = D.compact_mask(torch.rand(100, generator=g).to(gpu) > 0.5)
    oa, ob = D.gather_rows(idx, [a, b])
    assert ob.shape == (count, 0, 3) and torch.equal(oa, a[idx.long()])


class _Model:
    """The attributes of the reference's GaussianModel that its densification code touches (gs_renderer.py:134-216)."""
    percent_dense = 0.01
    scaling_activation = staticmethod(torch.exp)
    scaling_inverse_activation = staticmethod(torch.log)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)


def _build_rotation(r):                                        # gs_renderer.py:86-106, restated for the checker
    q = r / torch.norm(r, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                        2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                        2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)


def _ref_postfix(m, new):                                      # cat_tensors_to_optimizer + densification_postfix, gs_renderer.py:513-552
    for grp in m.optimizer.param_groups:
        ext = new[grp["name"]]
        p = grp["params"][0]
        st = m.optimizer.state[p]
        st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), 0)
        st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), 0)
        del m.optimizer.state[p]
        grp["params"][0] = torch.nn.Parameter(torch.cat((p, ext), 0).requires_grad_(True))
        m.optimizer.state[grp["params"][0]] = st
    ps = {g["name"]: g["params"][0] for g in m.optimizer.param_groups}
    m._xyz, m._features_dc, m._features_rest, m._opacity, m._scaling, m._rotation = (ps[k] for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"))
    n = m._xyz.shape[0]
    m.xyz_gradient_accum, m.denom, m.max_radii2D = torch.zeros(n, 1, device=m._xyz.device), torch.zeros(n, 1, device=m._xyz.device), torch.zeros(n, device=m._xyz.device)


def test_densify_clone_and_split_equal_reference_algorithm(gpu):
    """dreamgaussian_amd.densify_and_clone / densify_and_split (one compaction + one gather + one concatenation launch) on an
    object with GaussianModel's attributes against the reference's own sequence (boolean-mask indexing, 18 torch.cat, the same
    torch.normal draw; gs_renderer.py:513-595): parameters, Adam moments and accumulators equal bit for bit."""
    N = 3000
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")

    def build():
        params, opt = _make(N, gpu, D.FusedAdam, seed=11)
        with torch.no_grad():
            params[4].mul_(0.5).sub_(4.0)                       # log-scales around e^-4: both sides of percent_dense * extent
        for p in params:
            p.grad = torch.ones_like(p) * 0.1
        opt.step()
        m = _Model()
        m.optimizer = opt
        ga = torch.Generator().manual_seed(7)
        m.xyz_gradient_accum, m.denom, m.max_radii2D = torch.rand(N, 1, generator=ga).to(gpu), torch.rand(N, 1, generator=ga).to(gpu), torch.rand(N, generator=ga).to(gpu)
        for n, p in zip(names, params):
            setattr(m, n, p)
        return m
    a, b = build(), build()
    grads = torch.rand(N, 1, generator=torch.Generator().manual_seed(3)).to(gpu)
    thr, extent = 0.4, 2.0
    # ---- clone
    D.densify_and_clone(a, grads, thr, extent)
    sel = torch.logical_and(torch.norm(grads, dim=-1) >= thr, torch.max(b.get_scaling, dim=1).values <= b.percent_dense * extent)
    _ref_postfix(b, {"xyz": b._xyz[sel], "f_dc": b._features_dc[sel], "f_rest": b._features_rest[sel], "opacity": b._opacity[sel],
                     "scaling": b._scaling[sel], "rotation": b._rotation[sel]})
    assert 0 < int(sel.sum()) < N

    def same():
        for ga_, gb_, n in zip(a.optimizer.param_groups, b.optimizer.param_groups, names):
            pa, pb = ga_["params"][0], gb_["params"][0]
            assert pa is getattr(a, n) and pa.requires_grad and pa.shape == pb.shape, n
            assert torch.equal(pa.detach(), pb.detach()), n
            assert torch.equal(a.optimizer.state[pa]["exp_avg"], b.optimizer.state[pb]["exp_avg"]), n
            assert torch.equal(a.optimizer.state[pa]["exp_avg_sq"], b.optimizer.state[pb]["exp_avg_sq"]), n
        assert torch.equal(a.xyz_gradient_accum, b.xyz_gradient_accum) and torch.equal(a.denom, b.denom) and torch.equal(a.max_radii2D, b.max_radii2D)
    same()
    # ---- split (the same random draw on both sides)
    torch.manual_seed(5); torch.cuda.manual_seed(5)
    D.densify_and_split(a, grads, thr, extent, N=2, build_rotation=_build_rotation)
    torch.manual_seed(5); torch.cuda.manual_seed(5)
    n_init = b.get_xyz.shape[0]
    padded = torch.zeros(n_init, device=gpu); padded[:grads.shape[0]] = grads.squeeze()
    sel = torch.logical_and(padded >= thr, torch.max(b.get_scaling, dim=1).values > b.percent_dense * extent)
    stds = b.get_scaling[sel].repeat(2, 1)
    samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=gpu), std=stds)
    rots = _build_rotation(b._rotation[sel]).repeat(2, 1, 1)
    new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + b.get_xyz[sel].repeat(2, 1)
    with torch.no_grad():
        _ref_postfix(b, {"xyz": new_xyz, "f_dc": b._features_dc[sel].repeat(2, 1, 1), "f_rest": b._features_rest[sel].repeat(2, 1, 1),
                         "opacity": b._opacity[sel].repeat(2, 1), "scaling": torch.log(b.get_scaling[sel].repeat(2, 1) / (0.8 * 2)),
                         "rotation": b._rotation[sel].repeat(2, 1)})
    keep = ~torch.cat((sel, torch.zeros(2 * int(sel.sum()), device=gpu, dtype=torch.bool)))
    for grp in b.optimizer.param_groups:                        # prune_points, gs_renderer.py:479-511
        p = grp["params"][0]
        st = b.optimizer.state[p]
        st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][keep], st["exp_avg_sq"][keep]
        del b.optimizer.state[p]
        grp["params"][0] = torch.nn.Parameter(p[keep].requires_grad_(True))
        b.optimizer.state[grp["params"][0]] = st
    b.xyz_gradient_accum, b.denom, b.max_radii2D = b.xyz_gradient_accum[keep], b.denom[keep], b.max_radii2D[keep]
    assert 0 < int(sel.sum()) < n_init
    same()
    for p in (g["params"][0] for g in a.optimizer.param_groups):   # the optimiser keeps stepping on the new tensors
        p.grad = torch.ones_like(p)
    a.optimizer.step()


def test_reorder_gaussians_moves_every_row_and_renders_the_same_model(gpu):
    """dreamgaussian_amd.reorder_gaussians (Morton order, one gather launch): parameters, Adam moments and accumulators
    follow their rows bit for bit; the permutation is the Z-order sort of the positions; the rendering of the permuted
    model equals the original's (same Gaussians; only equal-depth ties inside a tile may swap) and per-Gaussian outputs
    (radii, gradients) are the permuted originals."""
    from dreamgaussian_amd import synthetic as syn
    N = 20_000
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
    sc = syn.make_scene(N, 3, seed=2, kind="trained")
    vals = [sc["means3D"], sc["shs"][:, :1].contiguous(), sc["shs"][:, 1:].contiguous(), sc["opacities"], sc["scales"], sc["rotations"]]
    params = [torch.nn.Parameter(v.to(gpu)) for v in vals]
    opt = D.FusedAdam([{"params": [p], "lr": lr, "name": name} for p, (name, _, lr) in zip(params, GROUPS)], lr=0.0, eps=1e-15)
    for p in params:
        p.grad = torch.zeros_like(p)
    opt.step()                                                   # creates the moments
    g = torch.Generator().manual_seed(9)
    for p in params:
        st = opt.state[p]
        st["exp_avg"], st["exp_avg_sq"] = torch.rand(p.shape, generator=g).to(gpu), torch.rand(p.shape, generator=g).to(gpu)
    m = _Model()
    m.optimizer = opt
    m.xyz_gradient_accum, m.denom, m.max_radii2D = torch.rand(N, 1, generator=g).to(gpu), torch.rand(N, 1, generator=g).to(gpu), torch.rand(N, generator=g).to(gpu)
    for n, p in zip(names, params):
        setattr(m, n, p)
    before = {n: getattr(m, n).detach().clone() for n in names}
    mom = {n: (opt.state[getattr(m, n)]["exp_avg"].clone(), opt.state[getattr(m, n)]["exp_avg_sq"].clone()) for n in names}
    acc = (m.xyz_gradient_accum.clone(), m.denom.clone(), m.max_radii2D.clone())

    rs_cpu = syn.make_settings(syn.orbit_pose(-10.0, 30.0, 2.0), 256, 256, sh_degree=3)
    rs = D.GaussianRasterizationSettings(*[x.to(gpu) if torch.is_tensor(x) else x for x in rs_cpu])

    def render(model):
        ps = [getattr(model, n).detach().clone().requires_grad_(True) for n in names]
        shs = torch.cat([ps[1], ps[2]], 1)
        c, r, d, a = D.GaussianRasterizer(rs)(means3D=ps[0], means2D=torch.zeros_like(ps[0], requires_grad=True), opacities=ps[3], shs=shs, scales=ps[4], rotations=ps[5])
        (c.sum() + d.sum() + a.sum()).backward()
        return c.detach(), d.detach(), a.detach(), r, [p.grad for p in ps]
    c0, d0, a0, r0, g0 = render(m)

    perm = D.reorder_gaussians(m).long()
    assert torch.equal(torch.sort(perm).values, torch.arange(N, device=gpu))
    assert torch.equal(perm.cpu(), D.morton_order(before["_xyz"].cpu()).long())          # the same order on either device
    for n in names:
        p = getattr(m, n)
        assert p.requires_grad and any(p is grp["params"][0] for grp in opt.param_groups), n
        assert torch.equal(p.detach(), before[n][perm]), n
        assert torch.equal(opt.state[p]["exp_avg"], mom[n][0][perm]) and torch.equal(opt.state[p]["exp_avg_sq"], mom[n][1][perm]), n
    assert torch.equal(m.xyz_gradient_accum, acc[0][perm]) and torch.equal(m.denom, acc[1][perm]) and torch.equal(m.max_radii2D, acc[2][perm])
    # neighbours on the curve are neighbours in space: the mean distance between consecutive rows collapses
    x = m._xyz.detach()
    assert float((x[1:] - x[:-1]).norm(dim=1).mean()) < 0.25 * float((before["_xyz"][1:] - before["_xyz"][:-1]).norm(dim=1).mean())

    c1, d1, a1, r1, g1 = render(m)
    assert torch.equal(r1, r0[perm])
    for u, v in ((c1, c0), (d1, d0), (a1, a0)):
        assert float((u - v).abs().max()) <= 2e-5, float((u - v).abs().max())    # fp32 order of the gradient/pixel sums does not enter here: ties only
    for u, v in zip(g1, g0):
        ref = v[perm]
        assert float((u - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), float((u - ref).abs().max())
    for p in (grp["params"][0] for grp in opt.param_groups):        # the optimiser keeps stepping on the new tensors
        p.grad = torch.ones_like(p)
    opt.step()


def test_clone_split_postfix_in_z_order_equal_the_plain_result_up_to_a_row_permutation(gpu):
    """`densification_postfix` / `densify_and_clone` / `densify_and_split` with `zorder=True` (round 6): the same Gaussians, each with
    its own Adam moments and (reset) accumulators, as the plain calls -- which equal the reference's sequence bit for bit (test above) --
    leave them; only the order of the rows differs, and it IS the Z-order of the positions."""
    N = 3000
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")

    def build():
        params, opt = _make(N, gpu, D.FusedAdam, seed=11)
        with torch.no_grad():
            params[4].mul_(0.5).sub_(4.0)
        for p in params:
            p.grad = torch.ones_like(p) * 0.1
        opt.step()
        m = _Model()
        m.optimizer = opt
        ga = torch.Generator().manual_seed(7)
        m.xyz_gradient_accum, m.denom, m.max_radii2D = torch.rand(N, 1, generator=ga).to(gpu), torch.rand(N, 1, generator=ga).to(gpu), torch.rand(N, generator=ga).to(gpu)
        for n, p in zip(names, params):
            setattr(m, n, p)
        return m

    def same_up_to_order(a, b, what):
        # b is the plain result; a must be b[perm] with perm = the Z-order of b's positions (distinct random positions: the key is unique)
        perm = D.morton_order(b._xyz.detach()).long()
        assert a._xyz.shape == b._xyz.shape, what
        assert torch.equal(D.morton_order(a._xyz.detach()).long(), torch.arange(a._xyz.shape[0], device=gpu)), what    # a IS on the curve
        for ga_, gb_, n in zip(a.optimizer.param_groups, b.optimizer.param_groups, names):
            pa, pb = ga_["params"][0], gb_["params"][0]
            assert pa is getattr(a, n) and pa.requires_grad, (what, n)
            assert torch.equal(pa.detach(), pb.detach()[perm]), (what, n)
            assert torch.equal(a.optimizer.state[pa]["exp_avg"], b.optimizer.state[pb]["exp_avg"][perm]), (what, n)
            assert torch.equal(a.optimizer.state[pa]["exp_avg_sq"], b.optimizer.state[pb]["exp_avg_sq"][perm]), (what, n)
        assert torch.equal(a.xyz_gradient_accum, b.xyz_gradient_accum[perm]) and torch.equal(a.denom, b.denom[perm]) and torch.equal(a.max_radii2D, b.max_radii2D[perm]), what

    grads = torch.rand(N, 1, generator=torch.Generator().manual_seed(3)).to(gpu)
    thr, extent = 0.4, 2.0
    # ---- postfix on its own
    a, b = build(), build()
    g = torch.Generator().manual_seed(21)
    new = [torch.rand(40, 3, generator=g), torch.rand(40, 1, 3, generator=g), torch.rand(40, 15, 3, generator=g), torch.rand(40, 1, generator=g),
           torch.rand(40, 3, generator=g), torch.rand(40, 4, generator=g)]
    new = [t.to(gpu) for t in new]
    new[2] = new[2][:, :a._features_rest.shape[1]].contiguous()
    D.densification_postfix(a, *new, zorder=True)
    D.densification_postfix(b, *new)
    same_up_to_order(a, b, "postfix")
    # ---- clone, then split (the same random draw on both sides: the selection is made before anything is permuted)
    a, b = build(), build()
    D.densify_and_clone(a, grads, thr, extent, zorder=True)
    D.densify_and_clone(b, grads, thr, extent)
    same_up_to_order(a, b, "clone")
    a, b = build(), build()
    torch.manual_seed(5); torch.cuda.manual_seed(5)
    D.densify_and_split(a, grads, thr, extent, N=2, build_rotation=_build_rotation, zorder=True)
    torch.manual_seed(5); torch.cuda.manual_seed(5)
    D.densify_and_split(b, grads, thr, extent, N=2, build_rotation=_build_rotation)
    same_up_to_order(a, b, "split")
    for p in (g_["params"][0] for g_ in a.optimizer.param_groups):   # the optimiser keeps stepping on the new tensors
        p.grad = torch.ones_like(p)
    a.optimizer.step()
