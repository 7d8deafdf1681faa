"""View-parallel host logic (dreamgaussian_amd/views.py) on CPU: two and four gloo processes.
Each rank renders its own camera (with the oracle standing in for the GPU rasterizer, as the
checker) and the gathered batch / reduced gradients must equal the serial result."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _np(x):
    """Tensors leave the workers as numpy arrays (pickled by value): a torch tensor travels as a shared-memory file descriptor
    that the parent must fetch while the worker is still alive -- a race with the worker's exit."""
    if torch.is_tensor(x):
        return x.detach().numpy().copy()
    if isinstance(x, dict):
        return {k: _np(v) for k, v in x.items()}
    return x


def _pt(x):
    import numpy as np
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x)
    if isinstance(x, dict):
        return {k: _pt(v) for k, v in x.items()}
    return x


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _render(cam_az, sc, W, H):
    from oracle import gs_oracle as O
    S = O.make_settings(O.orbit_pose(0.0, cam_az, 2.0), W, H, sh_degree=0)
    c, r, d, a = O.rasterize(sc["means3D"], None, sc["opacities"], S, shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
    return torch.cat([c, d, a], 0)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import gs_oracle as O
        from dreamgaussian_amd import views
        torch.manual_seed(0)
        W, H = 32, 24
        azs = [0.0, 90.0, 180.0, 270.0]
        sc = {k: v.clone().requires_grad_(True) for k, v in O.make_scene(120, 0, 0, "trained").items()}
        mine = views.shard_views(azs)
        assert mine == azs[rank::world]
        assert [views.owner_of(i) for i in range(4)] == [0, 1, 0, 1]
        local = torch.stack([_render(az, sc, W, H) for az in mine])            # [2,5,H,W]
        batch = views.gather_images(local.detach(), dst=0)
        everywhere = views.gather_images(local.detach(), dst=None)
        # async gather used by bench.py
        buf = views.make_gather_buffer(world, 5, H, W, "cpu")
        work = views.gather_views_async(local[0, :3], local[0, 3:4], local[0, 4:5], buf, dst=0)
        if work is not None:
            work.wait()
        # loss on the gathered batch lives on rank 0; its image gradient is scattered back
        gw = torch.rand(4, 5, H, W, generator=torch.Generator().manual_seed(1))
        g_local = views.scatter_view_grads(gw if rank == 0 else None, local, src=0)
        torch.autograd.backward([local], [g_local])
        params = list(sc.values())
        views.allreduce_grads(params, bucket_bytes=1 << 10)                    # several buckets
        # gradients carved out of ONE allocation (what the rasterizer's backward hands to autograd): reduced in place, in one
        # collective over their span -- same storage afterwards, the trailing (per-view) tensor of the allocation untouched
        flat = torch.full((256,), float(rank + 1))
        pa, pb, pv = (torch.nn.Parameter(torch.zeros(s)) for s in ((10, 3), (10, 1), (10, 3)))
        pa.grad, pb.grad, pv.grad = flat[0:30].view(10, 3), flat[64:74].view(10, 1), flat[128:158].view(10, 3)
        views.allreduce_grads([pa, pb])
        total = float(sum(range(1, world + 1)))
        assert pa.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()
        assert bool((pa.grad == total).all()) and bool((pb.grad == total).all()) and bool((pv.grad == rank + 1).all())
        # "local" placement of the loss (round 5): every rank differentiates its own views' terms of the same loss -- no image crosses
        # a link -- and the all-reduce is asynchronous (a handle, waited for before the gradients are read)
        sc_l = {k: v.detach().clone().requires_grad_(True) for k, v in O.make_scene(120, 0, 0, "trained").items()}
        local_l = torch.stack([_render(az, sc_l, W, H) for az in mine])
        torch.autograd.backward([local_l], [gw[rank::world]])
        h = views.allreduce_grads(list(sc_l.values()), bucket_bytes=1 << 10, async_op=True)
        assert isinstance(h, views.GradSync)
        h.wait()
        h.wait()                                                                # (idempotent)
        grads_local = {k: v.grad.clone() for k, v in sc_l.items()}
        if rank == 0:
            q.put(_np(dict(batch=batch, everywhere=everywhere, buf=buf.clone(),
                           grads={k: v.grad.clone() for k, v in sc.items()}, grads_local=grads_local)))
        else:
            assert batch is None
            q.put(_np(dict(everywhere=everywhere, grads={k: v.grad.clone() for k, v in sc.items()}, grads_local=grads_local)))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_view_parallel_equals_serial():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import gs_oracle as O
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [_pt(q.get(timeout=120)) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0 = next(r for r in res if "batch" in r)
    r1 = next(r for r in res if "batch" not in r)
    # serial reference
    W, H = 32, 24
    azs = [0.0, 90.0, 180.0, 270.0]
    sc = {k: v.clone().requires_grad_(True) for k, v in O.make_scene(120, 0, 0, "trained").items()}
    serial = torch.stack([_render(az, sc, W, H) for az in azs])
    gw = torch.rand(4, 5, H, W, generator=torch.Generator().manual_seed(1))
    torch.autograd.backward([serial], [gw])
    assert torch.allclose(r0["batch"], serial.detach(), atol=1e-6)
    assert torch.equal(r0["everywhere"], r1["everywhere"]) and torch.allclose(r0["everywhere"], serial.detach(), atol=1e-6)
    assert torch.allclose(r0["buf"][0], serial[0].detach(), atol=1e-6) and torch.allclose(r0["buf"][1], serial[1].detach(), atol=1e-6)
    for k, v in sc.items():
        for r in (r0, r1):
            assert torch.allclose(r["grads"][k], v.grad, rtol=1e-4, atol=1e-6 * v.grad.abs().max().item()), k
            assert torch.allclose(r["grads_local"][k], v.grad, rtol=1e-4, atol=1e-6 * v.grad.abs().max().item()), ("local", k)


def _worker_ragged(rank, world, port, q, azs):
    """The multi-view SDS step of main.py:219-275 over `world` ranks with UNEQUAL view counts: render own views,
    gather to rank 0, loss gradient on rank 0, scatter, local backward, bucketed all-reduce."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import gs_oracle as O
        from dreamgaussian_amd import views
        W, H = 24, 16
        sc = {k: v.clone().requires_grad_(True) for k, v in O.make_scene(80, 0, 0, "trained").items()}
        mine = views.shard_views(azs)
        assert len(mine) == views.views_on_rank(len(azs), rank, world)
        local = torch.stack([_render(az, sc, W, H) for az in mine]) if mine else torch.zeros(0, 5, H, W)
        batch = views.gather_images(local.detach(), dst=0, num_views=len(azs))
        gw = torch.rand(len(azs), 5, H, W, generator=torch.Generator().manual_seed(1))
        g_local = views.scatter_view_grads(gw if rank == 0 else None, local, src=0, num_views=len(azs))
        assert tuple(g_local.shape) == tuple(local.shape)
        if mine:
            torch.autograd.backward([local], [g_local])
        for v in sc.values():                      # a rank without views still takes part in the all-reduce
            if v.grad is None:
                v.grad = torch.zeros_like(v)
        views.allreduce_grads(list(sc.values()), bucket_bytes=1 << 9)
        q.put(_np(dict(rank=rank, batch=batch, grads={k: v.grad.clone() for k, v in sc.items()})))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("nviews", [6, 3], ids=["6_views_on_4_ranks", "3_views_on_4_ranks"])
def test_four_rank_unequal_view_counts_equal_serial(nviews):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import gs_oracle as O
    world, port = 4, _free_port()
    azs = [360.0 * i / nviews for i in range(nviews)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_ragged, args=(r, world, port, q, azs)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([_pt(q.get(timeout=180)) for _ in range(world)], key=lambda r: r["rank"])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    W, H = 24, 16
    sc = {k: v.clone().requires_grad_(True) for k, v in O.make_scene(80, 0, 0, "trained").items()}
    serial = torch.stack([_render(az, sc, W, H) for az in azs])
    gw = torch.rand(nviews, 5, H, W, generator=torch.Generator().manual_seed(1))
    torch.autograd.backward([serial], [gw])
    assert torch.allclose(res[0]["batch"], serial.detach(), atol=1e-6) and all(r["batch"] is None for r in res[1:])
    for k, v in sc.items():
        for r in res:
            assert torch.allclose(r["grads"][k], v.grad, rtol=1e-4, atol=1e-6 * v.grad.abs().max().item()), (k, r["rank"])


def test_single_process_fallthrough():
    from dreamgaussian_amd import views
    x = torch.rand(2, 5, 4, 4)
    assert views.shard_views([1, 2, 3]) == [1, 2, 3]
    assert views.gather_images(x) is x
    assert views.scatter_view_grads(x, x) is x
    views.allreduce_grads([torch.nn.Parameter(torch.zeros(3))])


def _worker_forced(port, q):
    """One rank, collectives forced: the path tests/test_views_gpu.py drives through RCCL on the 1-GPU box, here on gloo."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from dreamgaussian_amd import views
        views.force_collectives(True)
        x = torch.rand(3, 5, 4, 6)
        got = views.gather_images(x, dst=0)
        back = views.scatter_view_grads(2 * x, x, src=0)
        p = torch.nn.Parameter(torch.zeros(7)); p.grad = torch.arange(7.0)
        views.allreduce_grads([p])
        q.put(dict(fresh=got is not x, same=torch.equal(got, x), back=torch.equal(back, 2 * x), grad=torch.equal(p.grad, torch.arange(7.0))))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_forced_collectives_on_one_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_forced, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=60)
    p.join(timeout=30)
    assert p.exitcode == 0 and res == dict(fresh=True, same=True, back=True, grad=True)


def _worker_span(rank, world, port, q):
    """allreduce_grads on gradients laid out exactly as the rasterizer's backward lays them out for the split / raw-activation
    entry at SH degree 3 (rasterizer.carve_gradients: means3D | opacity | features_dc | scales | rotations | features_rest | means2D)."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dreamgaussian_amd import views
        from dreamgaussian_amd.rasterizer import carve_gradients
        N = 37                                             # not a multiple of 64: every gradient is followed by padding
        widths = [3, 1, 3, 0, 3, 4, 0, 45, 3]              # k_sh = 1 (features_dc), k_rest = 15, no colours / covariances
        names = ["means3D", "opacity", "features_dc", None, "scaling", "rotation", None, "features_rest", "means2D"]

        def fresh():
            flat, offs = carve_gradients(N, widths, "cpu")
            flat.fill_(float("nan"))                       # padding must never be read into a result
            params = {}
            for i, nm in enumerate(names):
                if nm is None:
                    continue
                p = torch.nn.Parameter(torch.zeros(N, widths[i]))
                g = flat[offs[i]:offs[i] + N * widths[i]].view(N, widths[i])
                g.copy_(torch.full((N, widths[i]), float(i + 1)) * (rank + 1))
                p.grad = g
                params[nm] = p
            return flat, params

        out = {}
        # (a) every parameter gradient (means2D is the per-view one: not passed): ONE in-place all-reduce over the span
        flat, params = fresh()
        ptr = flat.data_ptr()
        views.allreduce_grads([p for k, p in params.items() if k != "means2D"])
        tot = sum(r + 1 for r in range(world))
        out["span_ok"] = all(torch.equal(params[nm].grad, torch.full((N, widths[i]), float(i + 1) * tot))
                             for i, nm in enumerate(names) if nm not in (None, "means2D"))
        out["span_in_place"] = all(p.grad.untyped_storage().data_ptr() == ptr for p in params.values())
        out["means2D_untouched"] = torch.equal(params["means2D"].grad, torch.full((N, 3), 9.0 * (rank + 1)))
        # (b) the advisor's case: opacity left out of the first call (a live gradient sits in the hole between means3D and
        # features_dc) -> no span reduction; opacity stays local, a second call reduces it exactly once
        flat, params = fresh()
        views.allreduce_grads([p for k, p in params.items() if k not in ("means2D", "opacity")])
        out["hole_not_reduced"] = torch.equal(params["opacity"].grad, torch.full((N, 1), 2.0 * (rank + 1)))
        out["others_reduced"] = torch.equal(params["features_rest"].grad, torch.full((N, 45), 8.0 * tot))
        views.allreduce_grads([params["opacity"]])
        out["hole_reduced_once"] = torch.equal(params["opacity"].grad, torch.full((N, 1), 2.0 * tot))
        q.put(dict(rank=rank, **out))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_span_allreduce_on_the_split_entry_layout_and_its_hole_case():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_span, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert all(v for k, v in r.items() if k != "rank"), r


def test_gather_fallback_is_counted_and_refused_under_nccl(monkeypatch):
    """views.gather_views_async may replace `gather` by `all_gather` (world x the bytes) only on a backend without gather, counts
    and warns when it does, and re-raises under "nccl" (RCCL has gather: an exception there is a failure)."""
    from dreamgaussian_amd import views
    calls = {}
    monkeypatch.setattr(views, "_world", lambda group=None: (0, 2))
    monkeypatch.setattr(views.dist, "gather", lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no gather")))
    monkeypatch.setattr(views.dist, "all_gather_into_tensor", lambda *a, **k: calls.setdefault("all_gather", True))
    c, d, a = torch.rand(3, 4, 4), torch.rand(1, 4, 4), torch.rand(1, 4, 4)
    buf = views.make_gather_buffer(2, 5, 4, 4, "cpu")
    monkeypatch.setattr(views.dist, "get_backend", lambda group=None: "gloo")
    before = views.gather_fallbacks
    with pytest.warns(UserWarning, match="all_gather"):
        monkeypatch.setattr(views, "gather_fallbacks", 0)
        views.gather_views_async(c, d, a, buf)
    assert calls.get("all_gather") and views.gather_fallbacks == 1
    monkeypatch.setattr(views.dist, "get_backend", lambda group=None: "nccl")
    with pytest.raises(RuntimeError, match="no gather"):
        views.gather_views_async(c, d, a, buf)
    monkeypatch.setattr(views, "gather_fallbacks", before)


def _worker_sharded(rank, world, port, q):
    """ShardedAdam (reduce-scatter -> Adam on the own slice -> all-gather of the parameters) and allreduce_live_rows against the
    replicated path (dense all-reduce + torch.optim.Adam on every rank), on the split-entry gradient layout at SH degree 3 and on
    gradients that do not share a storage."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dreamgaussian_amd import views
        from dreamgaussian_amd.rasterizer import carve_gradients
        N = 203
        widths = [3, 1, 3, 0, 3, 4, 0, 45, 3]
        names = ["xyz", "opacity", "f_dc", None, "scaling", "rotation", None, "f_rest", "means2D"]
        lrs = {"xyz": 1e-3, "opacity": 5e-2, "f_dc": 1e-2, "scaling": 5e-3, "rotation": 5e-3, "f_rest": 5e-4}
        out = {}
        for layout in ("span", "separate"):
            g0 = torch.Generator().manual_seed(100)
            init = {nm: torch.randn(N, widths[i], generator=g0) for i, nm in enumerate(names) if nm not in (None, "means2D")}
            pa = {nm: torch.nn.Parameter(v.clone()) for nm, v in init.items()}      # sharded path
            pb = {nm: torch.nn.Parameter(v.clone()) for nm, v in init.items()}      # replicated path
            oa = torch.optim.Adam([{"params": [pa[nm]], "lr": lrs[nm], "name": nm} for nm in pa], lr=0.0, eps=1e-15)
            ob = torch.optim.Adam([{"params": [pb[nm]], "lr": lrs[nm], "name": nm} for nm in pb], lr=0.0, eps=1e-15)
            sh = views.ShardedAdam(oa)
            for it in range(3):
                gr = torch.Generator().manual_seed(1000 * it + rank)           # every rank its own gradients
                vals = {nm: torch.randn(N, widths[names.index(nm)], generator=gr) for nm in pa}
                if layout == "span":
                    flat, offs = carve_gradients(N, widths, "cpu")
                    flat.fill_(float("nan"))                                   # padding: never read into a result
                    for i, nm in enumerate(names):
                        if nm is None:
                            continue
                        g = flat[offs[i]:offs[i] + N * widths[i]].view(N, widths[i])
                        if nm == "means2D":
                            g.fill_(7.0)
                        else:
                            g.copy_(vals[nm]); pa[nm].grad = g
                else:
                    for nm in pa:
                        pa[nm].grad = vals[nm].clone()
                for nm in pb:
                    pb[nm].grad = vals[nm].clone()
                views.allreduce_grads(list(pb.values()))
                ob.step()
                sh.step()
                if layout == "span":
                    out["means2D_untouched"] = bool((flat[offs[8]:offs[8] + 3 * N] == 7.0).all())
            sh.gather_state()
            ok = True
            for nm in pa:
                # (equal up to the order of the fp32 sums: reduce-scatter and all-reduce add the ranks' gradients in different orders)
                ok = ok and torch.allclose(pa[nm].detach(), pb[nm].detach(), rtol=2e-6, atol=2e-6 * lrs[nm] * 3 + 1e-7)
                for key in ("exp_avg", "exp_avg_sq"):
                    ref = ob.state[pb[nm]][key]
                    ok = ok and torch.allclose(oa.state[pa[nm]][key], ref, rtol=2e-6, atol=2e-6 * float(ref.abs().max()))
                ok = ok and float(oa.state[pa[nm]]["step"]) == 3.0
            out["sharded_" + layout] = ok
            out["moved_" + layout] = all(not torch.equal(pa[nm].detach(), init[nm]) for nm in pa)
        # ---- live rows: a quarter of the Gaussians carries a gradient on each rank
        g0 = torch.Generator().manual_seed(7)
        ps = [torch.nn.Parameter(torch.zeros(N, w)) for w in (3, 1, 48)]
        qs = [torch.nn.Parameter(torch.zeros(N, w)) for w in (3, 1, 48)]
        gr = torch.Generator().manual_seed(50 + rank)
        live = torch.rand(N, generator=gr) < 0.25
        for p, qq in zip(ps, qs):
            g = torch.randn(p.shape, generator=gr) * live[:, None]
            p.grad, qq.grad = g.clone(), g.clone()
        U, n = views.allreduce_live_rows(ps, probe=[0, 1])
        views.allreduce_grads(qs)
        out["live_equal"] = all(torch.allclose(p.grad, qq.grad, rtol=1e-6, atol=1e-7) for p, qq in zip(ps, qs))
        out["live_fraction_sane"] = 0 < U < n == N
        q.put(dict(rank=rank, U=U, **out))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_adam_and_live_row_exchange_equal_the_replicated_path(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_sharded, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len({r["U"] for r in res}) == 1                 # every rank exchanged the same union
    for r in res:
        assert all(v for k, v in r.items() if k not in ("rank", "U")), sorted((k, v) for k, v in r.items() if not v)
