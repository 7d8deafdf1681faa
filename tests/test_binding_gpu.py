"""The two bindings of the rasterizer's entry points -- the C++ autograd function of dreamgaussian_amd/csrc/gsr_torch.cpp (the default when
it has been built) and the ctypes / Python autograd.Function of dreamgaussian_amd/rasterizer.py -- drive the same C ABI with the same
arguments: images bit-identical, gradients equal up to the order of the compositing backward's float atomics, the same behaviour at the
edges (no gradient wanted, a second backward of one forward, the fused and split entries, empty inputs, error messages)."""
import pytest
import torch

from oracle import gs_oracle as O
import util
from util import weights_for, settings_to
import dreamgaussian_amd as D
from dreamgaussian_amd import rasterizer as R

pytestmark = pytest.mark.gpu


def _both(fn):
    out = {}
    for name, on in (("cpp", True), ("ctypes", False)):
        old = D.use_cpp_binding(on)
        try:
            out[name] = fn()
        finally:
            D.use_cpp_binding(old)
    return out["cpp"], out["ctypes"]


def _close(ga, gb, tol=1e-5, floors=None):
    for k in ga:
        if ga[k] is None or gb[k] is None:
            assert ga[k] is None and gb[k] is None, k
            continue
        scale = max(gb[k].abs().max().item(), (floors or {}).get(k, 0.0)) + 1e-30     # (an isotropic blob's drotations is cancellation noise: its floor)
        assert ga[k].shape == gb[k].shape and (ga[k] - gb[k]).abs().max().item() <= tol * scale + 1e-12, k


def test_binding_is_built_and_used(gpu):
    assert D.binding_loaded(), "dreamgaussian_amd/_gsr_torch.so is missing: run python -m dreamgaussian_amd.build"
    sc = O.make_scene(300, 0, 0, "blob")
    S = O.make_settings(O.orbit_pose(0, 0, 2.0), 64, 64, sh_degree=0)
    util.run_hip(sc, S, gpu, weights_for(64, 64))
    assert R._last_via_binding and "_gsr_torch.so" in open("/proc/self/maps").read()
    old = D.use_cpp_binding(False)
    try:
        util.run_hip(sc, S, gpu, weights_for(64, 64))
        assert not R._last_via_binding
    finally:
        D.use_cpp_binding(old)


@pytest.mark.parametrize("case", [("trained", 4000, 3, 200, 136), ("blob", 2500, 0, 128, 128), ("trained", 900, 1, 70, 50)], ids=["sh3", "sh0", "small"])
def test_both_bindings_give_the_same_result(gpu, case):
    kind, N, deg, W, H = case
    sc = O.make_scene(N, deg, 0, kind)
    S = O.make_settings(O.orbit_pose(-10.0, 30.0, 2.0), W, H, sh_degree=deg)
    w = weights_for(H, W)
    (oa, ga, sa), (ob, gb, sb) = _both(lambda: util.run_hip(sc, S, gpu, w))
    for a, b in zip(oa, ob):
        assert torch.equal(a, b)
    _close(ga, gb, floors=util.grad_floors(sc, gb))
    for k in ("M", "M_ref", "V", "max_tile", "N", "H", "W", "K", "seg_shift"):
        assert sa[k] == sb[k], k


def test_fused_and_split_entries_through_both_bindings(gpu):
    N, deg, W, H = 3000, 3, 160, 120
    sc = O.make_scene(N, deg, 0, "trained")
    S = settings_to(O.make_settings(O.orbit_pose(5.0, 60.0, 2.0), W, H, sh_degree=deg), gpu)
    w = [x.to(gpu) for x in weights_for(H, W)]
    raw = dict(op=torch.logit(sc["opacities"].clamp(1e-4, 1 - 1e-4)), sc=torch.log(sc["scales"]), rot=sc["rotations"] * 1.7)

    def run(split):
        t = {k: v.to(gpu).requires_grad_(True) for k, v in dict(m=sc["means3D"], dc=sc["shs"][:, :1].contiguous(), rest=sc["shs"][:, 1:].contiguous(),
                                                                sh=sc["shs"], **raw).items()}
        m2 = torch.zeros(N, 3, device=gpu, requires_grad=True)
        if split:
            out = D.rasterize_gaussians_split(t["m"], m2, t["dc"], t["rest"], t["op"], t["sc"], t["rot"], S)
        else:
            out = D.rasterize_gaussians_raw(t["m"], m2, t["sh"], t["op"], t["sc"], t["rot"], S)
        torch.autograd.backward([out[0], out[2], out[3]], w)
        g = {k: (None if v.grad is None else v.grad.detach().clone()) for k, v in t.items()}
        g["m2"] = m2.grad.detach().clone()
        return [o.detach() for o in out], g
    for split in (False, True):
        (oa, ga), (ob, gb) = _both(lambda: run(split))
        for a, b in zip(oa, ob):
            assert torch.equal(a, b)
        _close(ga, gb)


def test_edges_through_both_bindings(gpu):
    sc = O.make_scene(1200, 1, 0, "trained")
    S = settings_to(O.make_settings(O.orbit_pose(0.0, 10.0, 2.0), 96, 80, sh_degree=1), gpu)
    w = [x.to(gpu) for x in weights_for(80, 96)]

    def run():
        res = {}
        t = {k: v.to(gpu).requires_grad_(True) for k, v in sc.items()}
        m2 = torch.zeros(1200, 3, device=gpu, requires_grad=True)
        rast = D.GaussianRasterizer(raster_settings=S)
        kw = dict(means3D=t["means3D"], means2D=m2, shs=t["shs"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
        # (1) a second backward of the same forward (retain_graph): its own accumulators, the same gradients twice over
        out = rast(**kw)
        torch.autograd.backward([out[0], out[2], out[3]], w, retain_graph=True)
        g1 = t["means3D"].grad.detach().clone()
        torch.autograd.backward([out[0], out[2], out[3]], w)
        res["twice"] = (t["means3D"].grad.detach() - 2 * g1).abs().max().item() / (g1.abs().max().item() + 1e-30)
        assert out[1].dtype == torch.int32 and not out[1].requires_grad
        # (2) inference: no input wants a gradient / torch.no_grad(): same images, a backward is refused by autograd
        with torch.no_grad():
            o2 = rast(**kw)
        assert not o2[0].requires_grad
        res["img"] = [o.detach().clone() for o in o2]
        d = {k: v.detach() for k, v in kw.items()}
        o3 = rast(**d)
        assert not o3[0].requires_grad and all(torch.equal(a, b) for a, b in zip(o3, o2))
        # (3) only the colour is differentiated
        for v in t.values():
            v.grad = None
        out = rast(**kw)
        out[0].sum().backward()
        res["col"] = {k: v.grad.detach().clone() for k, v in t.items()}
        # (4) no Gaussians at all: the background, gradients of the right (empty) shapes
        e = {k: v[:0].detach().to(gpu).requires_grad_(True) for k, v in sc.items()}
        oe = rast(means3D=e["means3D"], means2D=torch.zeros(0, 3, device=gpu, requires_grad=True), shs=e["shs"], opacities=e["opacities"],
                  scales=e["scales"], rotations=e["rotations"])
        oe[0].sum().backward()
        res["empty"] = (oe[0].detach().clone(), tuple(e["means3D"].grad.shape), e["shs"].grad is None)     # (an empty tensor counts as absent: no gradient for it)
        # (5) CPU tensors raise
        with pytest.raises(RuntimeError, match="no CPU fallback|There is no CPU"):
            rast(**{k: v.detach().cpu() for k, v in kw.items()})
        return res
    a, b = _both(run)
    assert a["twice"] <= 1e-5 and b["twice"] <= 1e-5
    for x, y in zip(a["img"], b["img"]):
        assert torch.equal(x, y)
    _close(a["col"], b["col"])
    assert torch.equal(a["empty"][0], b["empty"][0]) and a["empty"][1:] == b["empty"][1:] == ((0, 3), True)
