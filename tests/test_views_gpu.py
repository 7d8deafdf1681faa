"""rasterize_views (B cameras through one launch chain, gsr_forward_views / gsr_backward_views) must equal B
single-view calls: forward bit for bit, gradients of the shared Gaussians = sum over views."""
import pytest
import torch

from oracle import gs_oracle as O
from util import settings_to, weights_for, grad_floors
import dreamgaussian_amd as D

pytestmark = pytest.mark.gpu


# the last case is BASELINE.json configs[3] at full size: 250k Gaussians, SH degree 0, 512x512, 8 orbit cameras (8 x 1024 tiles,
# lists of several thousand entries: every sort class below the HBM fallback, both forward kernels chosen per view)
@pytest.mark.parametrize("deg,N,size,nviews,kind", [(0, 3000, 128, 5, "trained"), (3, 1500, 96, 5, "trained"), (1, 800, 80, 19, "trained"),
                                                     (0, 250_000, 512, 8, "blob"), (3, 100_000, 800, 8, "trained")],
                         ids=["sh0_3000_128_5v", "sh3_1500_96_5v", "sh1_800_80_19v", "cfg3_250k_512_8v", "cfg1x8_100k_sh3_800_8v"])
def test_batched_views_equal_serial(gpu, deg, N, size, nviews, kind):
    """(the last case, round 5: SH degree 3 x 8 views at 100k Gaussians / 800^2 -- the size the one-launch K6 with staged SH rows and a
    second LDS row per Gaussian exists for; its view 0 is also held to the fp64 oracle)"""
    sc = O.make_scene(N, deg, 0, kind)
    azs = [0.0, 70.0, 160.0, -95.0, 33.0] + [20.0 * i + 5 for i in range(nviews - 5)]   # 19 views: two chunks of the chain
    if N >= 100_000:                                       # configs[3]: the orbit of main.py:219-255, radius 2
        S = [settings_to(O.make_settings(O.orbit_pose(0.0, 45.0 * i, 2.0), size, size, sh_degree=deg), gpu) for i in range(nviews)]
    else:
        S = [settings_to(O.make_settings(O.orbit_pose(-10.0 + 7 * i, az, 2.0 + 0.1 * i), size, size, sh_degree=deg), gpu)
             for i, az in enumerate(azs)]
    B = len(S)
    w = [[x.to(gpu) for x in weights_for(size, size, seed=10 + i)] for i in range(B)]

    def leaves():
        return {k: v.to(gpu).requires_grad_(True) for k, v in sc.items()}

    # serial: the reference's loop (main.py:219-255)
    t = leaves()
    m2 = [torch.zeros(N, 3, device=gpu, requires_grad=True) for _ in range(B)]
    outs = []
    for i in range(B):
        c, r, d, a = D.GaussianRasterizer(raster_settings=S[i])(
            means3D=t["means3D"], means2D=m2[i], shs=t["shs"], colors_precomp=None, opacities=t["opacities"],
            scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
        outs.append((c, r, d, a))
    loss = sum((w[i][0] * outs[i][0]).sum() + (w[i][1] * outs[i][2]).sum() + (w[i][2] * outs[i][3]).sum() for i in range(B))
    loss.backward()
    ref = {k: v.grad.clone() for k, v in t.items()}

    # batched
    t2 = leaves()
    m2b = torch.zeros(B, N, 3, device=gpu, requires_grad=True)
    color, radii, depth, alpha = D.rasterize_views(t2["means3D"], m2b, t2["opacities"], S, shs=t2["shs"],
                                                   scales=t2["scales"], rotations=t2["rotations"])
    assert color.shape == (B, 3, size, size) and radii.shape == (B, N) and depth.shape == (B, 1, size, size)
    bad = []
    for i in range(B):
        for name, got, want in (("color", color[i], outs[i][0]), ("radii", radii[i], outs[i][1]), ("depth", depth[i], outs[i][2]),
                                ("alpha", alpha[i], outs[i][3])):
            if not torch.equal(got, want):
                bad.append((i, name, int((got != want).sum()), float((got.double() - want.double()).abs().max())))
    assert not bad, f"views differing from their single-view render (view, output, elements, max |diff|): {bad}"
    lossb = sum((w[i][0] * color[i]).sum() + (w[i][1] * depth[i]).sum() + (w[i][2] * alpha[i]).sum() for i in range(B))
    lossb.backward()
    floors = grad_floors(sc, {k: v.cpu() for k, v in ref.items()})     # isotropic scene: dL/drotations is rounding noise around 0
    for k in ref:
        scale = max(ref[k].abs().max().item(), floors.get(k, 0.0)) + 1e-12
        assert (t2[k].grad - ref[k]).abs().max().item() <= 2e-5 * scale, k
    for i in range(B):
        scale = m2[i].grad.abs().max().item() + 1e-12
        assert (m2b.grad[i] - m2[i].grad).abs().max().item() <= 2e-5 * scale
    if deg == 3 and N >= 100_000:
        # view 0 of the chain against the fp64 oracle: images, radii, and the one gradient that is per view (means2D)
        import util
        S0 = O.make_settings(O.orbit_pose(0.0, 0.0, 2.0), size, size, sh_degree=deg)
        oo, og, aux = util.run_oracle(sc, S0, [x.cpu() for x in w[0]], torch.float64)
        util.assert_forward_close([color[0].detach().cpu(), radii[0].cpu(), depth[0].detach().cpu(), alpha[0].detach().cpu()], oo, aux)
        util.assert_grads_close({"means2D": m2b.grad[0].cpu()}, {"means2D": og["means2D"]}, aux, row_rel_p999=util.ROW_REL_P999_FULL)


def test_views_chain_precomputed_colours_and_covariances(gpu):
    """The accumulate-into-one-gradient path of the per-Gaussian backward for the other input forms: precomputed
    colours + 3D covariances, and degree-0 SH with K == 1 (rows written straight to HBM, not through LDS)."""
    N, size, B = 1200, 96, 4
    sc = O.make_scene(N, 0, 0, "trained")
    S = [settings_to(O.make_settings(O.orbit_pose(-5.0 + 6 * i, 80.0 * i, 2.1), size, size, sh_degree=0), gpu) for i in range(B)]
    w = [[x.to(gpu) for x in weights_for(size, size, seed=30 + i)] for i in range(B)]
    S33 = O.covariance3d(sc["scales"], 1.0, sc["rotations"])
    cov = torch.stack([S33[:, 0, 0], S33[:, 0, 1], S33[:, 0, 2], S33[:, 1, 1], S33[:, 1, 2], S33[:, 2, 2]], 1).contiguous()
    variants = [dict(shs=sc["shs"][:, :1].contiguous(), scales=sc["scales"], rotations=sc["rotations"]),
                dict(colors_precomp=torch.rand(N, 3, generator=torch.Generator().manual_seed(3)), cov3D_precomp=cov)]
    for kw in variants:
        def leaves():
            d = {k: v.to(gpu).requires_grad_(True) for k, v in kw.items()}
            d["means3D"] = sc["means3D"].to(gpu).requires_grad_(True)
            d["opacities"] = sc["opacities"].to(gpu).requires_grad_(True)
            return d
        t = leaves()
        outs = []
        for i in range(B):
            args = dict(shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None)
            args.update({k: t[k] for k in kw})
            outs.append(D.GaussianRasterizer(raster_settings=S[i])(means3D=t["means3D"], means2D=torch.zeros(N, 3, device=gpu, requires_grad=True),
                                                                   opacities=t["opacities"], **args))
        sum((w[i][0] * outs[i][0]).sum() + (w[i][1] * outs[i][2]).sum() + (w[i][2] * outs[i][3]).sum() for i in range(B)).backward()
        t2 = leaves()
        color, radii, depth, alpha = D.rasterize_views(t2["means3D"], torch.zeros(B, N, 3, device=gpu, requires_grad=True), t2["opacities"], S,
                                                       **{k: t2[k] for k in kw})
        for i in range(B):
            assert torch.equal(color[i], outs[i][0]) and torch.equal(radii[i], outs[i][1]) and torch.equal(alpha[i], outs[i][3])
        sum((w[i][0] * color[i]).sum() + (w[i][1] * depth[i]).sum() + (w[i][2] * alpha[i]).sum() for i in range(B)).backward()
        for k in t:
            scale = t[k].grad.abs().max().item() + 1e-12
            assert (t2[k].grad - t[k].grad).abs().max().item() <= 2e-5 * scale, (sorted(kw), k)


def test_views_chain_without_gaussians_renders_the_backgrounds(gpu):
    bgs = [(1.0, 1.0, 1.0), (0.0, 0.25, 0.5), (0.3, 0.0, 0.0)]
    S = [settings_to(O.make_settings(O.orbit_pose(0, 40.0 * i, 2.0), 40, 24, bg=bg), gpu) for i, bg in enumerate(bgs)]
    e = lambda *s: torch.zeros(*s, device=gpu)
    color, radii, depth, alpha = D.rasterize_views(e(0, 3), e(3, 0, 3), e(0, 1), S, shs=e(0, 1, 3), scales=e(0, 3), rotations=e(0, 4))
    assert radii.shape == (3, 0) and float(depth.abs().max()) == 0.0 and float(alpha.abs().max()) == 0.0
    for i, bg in enumerate(bgs):
        assert torch.equal(color[i], torch.tensor(bg, device=gpu).view(3, 1, 1).expand(3, 24, 40))


def test_batched_views_argument_errors(gpu):
    sc = {k: v.to(gpu) for k, v in O.make_scene(50, 0, 0, "blob").items()}
    S = [settings_to(O.make_settings(O.orbit_pose(0, 0, 2.0), 32, 32), gpu),
         settings_to(O.make_settings(O.orbit_pose(0, 90, 2.0), 48, 32), gpu)]
    with pytest.raises(RuntimeError, match="one image size"):
        D.rasterize_views(sc["means3D"], torch.zeros(2, 50, 3, device=gpu), sc["opacities"], S, shs=sc["shs"],
                          scales=sc["scales"], rotations=sc["rotations"])
    with pytest.raises(RuntimeError, match="num_views"):
        D.rasterize_views(sc["means3D"], torch.zeros(50, 3, device=gpu), sc["opacities"], S[:1], shs=sc["shs"],
                          scales=sc["scales"], rotations=sc["rotations"])


def test_collectives_through_rccl_on_one_rank(gpu):
    """The view-parallel exchange (views.py: gather of the images, scatter of dL/dimage, all-reduce of the gradients) pushed
    through RCCL on this 1-GPU box: a ONE-rank "nccl" group with the single-rank short cuts switched off. Same calls, buffers
    and device tensors as on 8 GPUs; the result must equal the path without collectives, the all-reduce must run in place
    on the rasterizer's gradient allocation."""
    import os
    import socket
    import torch.distributed as dist
    from dreamgaussian_amd import views
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sc = O.make_scene(4000, 1, 0, "trained")
    size = 96
    S = settings_to(O.make_settings(O.orbit_pose(5.0, 40.0, 2.0), size, size, sh_degree=1), gpu)
    wimg = torch.rand(1, 5, size, size, generator=torch.Generator().manual_seed(3)).to(gpu)

    def sds_step():
        t = {k: v.to(gpu).requires_grad_(True) for k, v in sc.items()}
        m2 = torch.zeros(4000, 3, device=gpu, requires_grad=True)
        c, r, d, a = D.GaussianRasterizer(raster_settings=S)(means3D=t["means3D"], means2D=m2, shs=t["shs"], colors_precomp=None,
                                                              opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
        local = torch.cat([c, d, a], 0).unsqueeze(0)
        batch = views.gather_images(local.detach(), dst=0, num_views=1)
        g_local = views.scatter_view_grads((batch - 0.5) * wimg, local, src=0, num_views=1)
        torch.autograd.backward([local], [g_local])
        params = list(t.values())
        ptrs = {p.grad.untyped_storage().data_ptr() for p in params}
        views.allreduce_grads(params)
        assert {p.grad.untyped_storage().data_ptr() for p in params} == ptrs
        return batch.clone(), {k: v.grad.clone() for k, v in t.items()}, m2.grad.clone(), len(ptrs)

    ref_batch, ref_g, ref_m2, _ = sds_step()                 # no process group: no collective runs
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu)
    try:
        views.force_collectives(True)
        batch, g, m2g, nstorage = sds_step()
        torch.cuda.synchronize()
    finally:
        views.force_collectives(False)
        dist.destroy_process_group()
    assert nstorage == 1                                     # all parameter gradients live in the rasterizer's one allocation
    assert torch.equal(batch, ref_batch)
    g["means2D"], ref_g["means2D"] = m2g, ref_m2            # (not reduced: the last tensor of the allocation, outside the span)
    for k in g:
        scale = ref_g[k].abs().max().item() + 1e-30
        assert (g[k] - ref_g[k]).abs().max().item() <= 1e-4 * scale, k      # two backward passes: fp32 atomic order


def test_sharded_adam_and_live_rows_through_rccl_on_one_rank(gpu):
    """views.ShardedAdam (reduce-scatter -> gsr_adam_step on the own slice -> all-gather of the parameters) and
    views.allreduce_live_rows executed through RCCL on this 1-GPU box (ONE-rank "nccl" group, short cuts off), on the gradients the
    rasterizer's backward really leaves (one allocation, span reduced in place): parameters and moments after two steps equal
    FusedAdam's on the same gradients; the live-row exchange returns the gradients unchanged and names the live fraction."""
    import os
    import socket
    import torch.distributed as dist
    from dreamgaussian_amd import views
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    N, size = 5000, 128
    sc = O.make_scene(N, 2, 0, "trained")
    S = settings_to(O.make_settings(O.orbit_pose(5.0, 40.0, 2.0), size, size, sh_degree=2), gpu)
    w = [x.to(gpu) for x in weights_for(size, size)]
    names = list(sc.keys())

    def run(mode):
        t = {k: torch.nn.Parameter(v.to(gpu).clone()) for k, v in sc.items()}
        params = [t[k] for k in names]
        opt = D.FusedAdam([{"params": [p], "lr": 1e-3} for p in params], lr=0.0, eps=1e-15)
        sh = views.ShardedAdam(opt) if mode == "sharded" else None
        frac = None
        for it in range(2):
            for p in params:
                p.grad = None
            m2 = torch.zeros(N, 3, device=gpu, requires_grad=True)
            # (normalised rotations would need an activation in front: the raw tensors ARE the inputs here, as in the fused entry)
            c, r, d, a = D.GaussianRasterizer(raster_settings=S)(means3D=t["means3D"], means2D=m2, shs=t["shs"], colors_precomp=None,
                                                                  opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
            torch.autograd.backward([c, d, a], w)
            if it == 0:
                assert len({p.grad.untyped_storage().data_ptr() for p in params}) == 1
            if mode == "sharded":
                sh.step()
            else:
                if mode == "live":
                    before = [p.grad.clone() for p in params]
                    U, n = views.allreduce_live_rows(params, probe=[names.index("means3D"), names.index("opacities")])
                    frac = U / n
                    for p, b in zip(params, before):
                        assert torch.equal(p.grad, b)          # one rank: the sum over the ranks is the gradient itself, dead rows untouched (zeros)
                opt.step()
        if sh is not None:
            sh.gather_state()
        return {k: t[k].detach().clone() for k in names}, {k: (opt.state[t[k]]["exp_avg"].clone(), opt.state[t[k]]["exp_avg_sq"].clone()) for k in names}, frac

    ref_p, ref_m, _ = run("plain")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu)
    try:
        views.force_collectives(True)
        sh_p, sh_m, _ = run("sharded")
        lv_p, lv_m, frac = run("live")
        torch.cuda.synchronize()
    finally:
        views.force_collectives(False)
        dist.destroy_process_group()
    assert 0.0 < frac < 1.0
    for k in names:
        for got_p, got_m in ((sh_p, sh_m), (lv_p, lv_m)):
            # (two separate backward passes: the float atomics of the compositing backward add in another order, Adam's m / sqrt(v) carries it)
            # m / sqrt(v) is sign-like in Adam's first steps: an element whose gradient is cancellation noise may step the other way)
            dp = (got_p[k] - ref_p[k]).abs()
            assert dp.max().item() <= 2 * 2 * 1e-3 * 1.01 and (dp > 1e-5).float().mean().item() <= 2e-3, (k, dp.max().item(), (dp > 1e-5).float().mean().item())
            for i in range(2):
                scale = ref_m[k][i].abs().max().item() + 1e-30
                assert (got_m[k][i] - ref_m[k][i]).abs().max().item() <= 1e-4 * scale, (k, i)
