"""The host side of the reference's REAL loop (main.py:198-216: a 256^2 known view and a 128^2 / 256^2 / 512^2 novel view every
iteration, depth never differentiated), ABI 6 of include/gsr.h: the shape-keyed table of list-size predictions, NULL incoming
gradients, the forward that does not wait for its counters (GSR_VIEW_ASYNC_STATS), and rendering from two threads / two streams."""
import threading

import pytest
import torch

from oracle import gs_oracle as O
import util
from util import run_hip, run_oracle, weights_for, assert_forward_close, assert_grads_close, settings_to, grad_floors
import dreamgaussian_amd as D
from dreamgaussian_amd import rasterizer as R

pytestmark = pytest.mark.gpu


def _render(sc, S, dev, weights=None, use=(True, True, True)):
    """forward (+ backward through the outputs named by `use` = colour / depth / alpha); everything stays on the device"""
    t = {k: v.detach().to(dev).requires_grad_(True) for k, v in sc.items()}
    m2d = torch.zeros(t["means3D"].shape[0], 3, device=dev, requires_grad=True)
    out = D.GaussianRasterizer(raster_settings=settings_to(S, dev))(
        means3D=t["means3D"], means2D=m2d, shs=t.get("shs"), colors_precomp=t.get("colors_precomp"), opacities=t["opacities"],
        scales=t.get("scales"), rotations=t.get("rotations"), cov3D_precomp=t.get("cov3D_precomp"))
    grads = None
    if weights is not None:
        outs = [out[0], out[2], out[3]]
        sel = [i for i in range(3) if use[i]]
        torch.autograd.backward([outs[i] for i in sel], [weights[i].to(dev) for i in sel])
        grads = {k: v.grad.detach().clone() for k, v in t.items()}
        grads["means2D"] = m2d.grad.detach().clone()
    return [o.detach() for o in out], grads


def test_alternating_resolutions_both_speculate(gpu):
    """main.py:198-216 renders 256^2 and then another resolution EVERY iteration. The predictions are kept per (N, H, W, views): from
    the second pair on both renders are enqueued without waiting for the host (GsrStats.speculated), and what they render does not
    depend on it."""
    sc = O.make_scene(4000, 0, 0, "blob")
    shapes = [(256, 256), (128, 128), (512, 512), (96, 160)]
    first = {}
    for it in range(3):
        for (H, W) in shapes:
            S = O.make_settings(O.orbit_pose(0.0, 20.0 * it, 2.0), W, H, sh_degree=0)
            out, _ = _render(sc, S, gpu)
            st = D.last_stats()
            assert st["pending"] == 0
            assert st["speculated"] == (1 if it > 0 else 0), (it, H, W, st)
            if it == 0:
                first[(H, W)] = st["M"]
    # nine shapes push the first ones out of the 8-entry table (least recently used first): they start over, the recent ones do not
    for k in range(9):
        S = O.make_settings(O.orbit_pose(0.0, 0.0, 2.0), 64 + 16 * k, 64, sh_degree=0)
        _render(sc, S, gpu)
        assert D.last_stats()["speculated"] == 0
    S = O.make_settings(O.orbit_pose(0.0, 0.0, 2.0), 256, 256, sh_degree=0)
    _render(sc, S, gpu)
    assert D.last_stats()["speculated"] == 0
    S = O.make_settings(O.orbit_pose(0.0, 0.0, 2.0), 64 + 16 * 8, 64, sh_degree=0)
    _render(sc, S, gpu)
    assert D.last_stats()["speculated"] == 1


@pytest.mark.parametrize("use", [(True, False, False), (False, False, True), (True, False, True), (False, True, False)],
                         ids=["colour", "alpha", "colour+alpha", "depth"])
def test_absent_incoming_gradients_are_zeros(gpu, use):
    """A loss that does not touch an output hands the backward None for it; the library takes NULL (ABI 6) where the wrapper used to
    build a zero image. Same gradients as explicit zero images, up to the order of the float atomics."""
    N, H, W = 3000, 120, 200
    sc = O.make_scene(N, 2, 0, "trained")
    S = O.make_settings(O.orbit_pose(-10.0, 30.0, 2.0), W, H, sh_degree=2)
    w = weights_for(H, W)
    _, ga = _render(sc, S, gpu, w, use)
    wz = [w[i] if use[i] else torch.zeros_like(w[i]) for i in range(3)]
    _, gb = _render(sc, S, gpu, wz, (True, True, True))
    for k in ga:
        scale = gb[k].abs().max().item() + 1e-30
        assert (ga[k] - gb[k]).abs().max().item() <= 1e-5 * scale, k
        assert torch.isfinite(ga[k]).all()


def test_no_incoming_gradient_at_all_gives_exact_zeros(gpu):
    sc = O.make_scene(1500, 0, 0, "blob")
    S = O.make_settings(O.orbit_pose(0.0, 0.0, 2.0), 96, 96, sh_degree=0)
    t = {k: v.detach().to(gpu).requires_grad_(True) for k, v in sc.items()}
    m2d = torch.zeros(1500, 3, device=gpu, requires_grad=True)
    out = D.GaussianRasterizer(raster_settings=settings_to(S, gpu))(means3D=t["means3D"], means2D=m2d, shs=t["shs"], opacities=t["opacities"],
                                                                  scales=t["scales"], rotations=t["rotations"])
    (out[0].sum() * 0.0).backward()       # a zero colour gradient, nothing for depth / alpha
    for k, v in t.items():
        assert v.grad is not None and (v.grad == 0).all(), k


def test_async_forward_matches_the_blocking_one(gpu):
    """GSR_VIEW_ASYNC_STATS (opt-in): from the second call of a shape on gsr_forward returns without waiting for its counters. Images
    bit-identical to the blocking forward's, gradients equal up to atomic order, the counts arrive through last_stats()."""
    N, H, W = 6000, 160, 160
    sc = O.make_scene(N, 1, 0, "trained")
    w = weights_for(H, W)
    poses = [O.orbit_pose(-10.0, 40.0 * i, 2.0) for i in range(4)]
    ref = []
    for p in poses:
        S = O.make_settings(p, W, H, sh_degree=1)
        o, g = _render(sc, S, gpu, w)
        ref.append((o, g, D.last_stats()))
    old = D.set_async_forward(True)
    try:
        S0 = O.make_settings(poses[0], W, H, sh_degree=1)
        _render(sc, S0, gpu, w)                                   # (the shape is known from the loop above: already asynchronous)
        for p, (o_ref, g_ref, st_ref) in zip(poses, ref):
            S = O.make_settings(p, W, H, sh_degree=1)
            o, g = _render(sc, S, gpu, w)
            pk = R.peek_stats()
            assert pk["pending"] != 0 and pk["M"] == -1                              # returned before the counters were collected
            st = D.last_stats()
            assert st["pending"] == 0 and st["speculated"] == 1
            assert (st["M"], st["M_ref"], st["V"], st["max_tile"]) == (st_ref["M"], st_ref["M_ref"], st_ref["V"], st_ref["max_tile"])
            for a, b in zip(o, o_ref):
                assert torch.equal(a, b)
            for k in g:
                scale = g_ref[k].abs().max().item() + 1e-30
                assert (g[k] - g_ref[k]).abs().max().item() <= 1e-5 * scale, k
    finally:
        D.set_async_forward(old)


def test_async_forward_overflow_is_loud(gpu):
    """An asynchronous forward whose lists outgrow the speculative capacity (the largest recent call of the shape + 50 %) cannot be
    repeated -- the host has returned. Its images are NaN, not stale memory; the thread's next forward raises (-6) and the one after
    that renders correctly (the table has learnt the new size)."""
    W = H = 64
    S = O.make_settings(O.orbit_pose(0, 0, 2.0), W, H, sh_degree=0)
    N = 12000
    g = torch.Generator().manual_seed(3)
    small = O.make_scene(N, 0, 3, "blob")
    big = dict(small); big["scales"] = small["scales"] * 6.0
    old = D.set_async_forward(True)
    try:
        _render(small, S, gpu); _render(small, S, gpu)
        assert D.last_stats()["speculated"] == 1
        out, _ = _render(big, S, gpu)
        assert torch.isnan(out[0]).all() and torch.isnan(out[2]).all() and torch.isnan(out[3]).all()
        with pytest.raises(RuntimeError, match="speculative capacity"):
            _render(big, S, gpu)
        out2, _ = _render(big, S, gpu)
        D.set_async_forward(False)
        out3, _ = _render(big, S, gpu)
        assert torch.isfinite(out2[0]).all()
        for a, b in zip(out2, out3):
            assert torch.equal(a, b)
    finally:
        D.set_async_forward(old)


def test_two_threads_two_streams(gpu):
    """The library's per-call state is thread-local (predictions, the pinned counter block) or inside the caller's scratch; autograd
    runs the backward on its own thread anyway. Two host threads, each with its own stream and scene, rendering at the same time,
    get what they get alone."""
    cases = [(O.make_scene(5000, 1, 0, "trained"), O.make_settings(O.orbit_pose(-10.0, 30.0, 2.0), 200, 136, sh_degree=1), weights_for(136, 200, 1)),
             (O.make_scene(3000, 0, 1, "trained"), O.make_settings(O.orbit_pose(5.0, 200.0, 2.0), 128, 128, sh_degree=0), weights_for(128, 128, 2))]
    alone = [_render(sc, S, gpu, w) for sc, S, w in cases]
    res, errs = [None, None], []
    start = threading.Barrier(2)

    def work(i):
        try:
            sc, S, w = cases[i]
            s = torch.cuda.Stream(device=gpu)
            with torch.cuda.stream(s):
                start.wait()
                for _ in range(6):
                    res[i] = _render(sc, S, gpu, w)
                s.synchronize()
        except Exception as e:                                    # noqa: BLE001 -- reported below
            errs.append((i, repr(e)))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    for i in range(2):
        (o, g), (o_ref, g_ref) = res[i], alone[i]
        for a, b in zip(o, o_ref):
            assert torch.equal(a, b)
        for k in g:
            scale = g_ref[k].abs().max().item() + 1e-30
            assert (g[k] - g_ref[k]).abs().max().item() <= 1e-5 * scale, (i, k)     # (the order of the float atomics differs from run to run)
