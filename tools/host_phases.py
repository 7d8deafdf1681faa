"""Host time of one fwd+bwd through the drop-in surface at 5k Gaussians / 256^2, phase by phase (perf_counter around the ctypes calls and
around the wrapper's own Python, on whichever thread runs them), and the same step with the host kept from running ahead of / behind
the GPU in different ways. Runs on the GPU box:  python tools/host_phases.py"""
import os, sys, time, threading, statistics as st
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dreamgaussian_amd as D
from dreamgaussian_amd import synthetic as syn, _lib, rasterizer as R

dev = torch.device("cuda:0")
N, W = int(os.environ.get("N", 5000)), int(os.environ.get("W", 256))
sc = syn.make_scene(N, 0, 0, "blob")
rs = syn.make_settings(syn.orbit_pose(0, 0, 2.0), W, W, sh_degree=0, device=dev)
t = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
g = [torch.rand(3, W, W, device=dev), torch.rand(1, W, W, device=dev), torch.rand(1, W, W, device=dev)]
rast = D.GaussianRasterizer(raster_settings=rs)
lib = _lib.load()
acc = {}

def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter_ns()
        r = fn(*a, **k)
        acc.setdefault(name, []).append((time.perf_counter_ns() - t0) / 1e3)
        return r
    return w

class LibProxy:
    def __init__(self, lib): self._l = lib
    def __getattr__(self, n):
        f = getattr(self._l, n)
        return timed("C:" + n, f) if n in ("gsr_forward", "gsr_backward") else f
proxy = LibProxy(lib)
_lib.load = lambda: proxy
R._RasterizeGaussians.forward = staticmethod(timed("py:forward (incl. C)", R._RasterizeGaussians.forward))
R._RasterizeGaussians.backward = staticmethod(timed("py:backward (incl. C)", R._RasterizeGaussians.backward))

def step():
    c, r, d, a = rast(means3D=t["means3D"], means2D=m2d, shs=t["shs"], colors_precomp=None, opacities=t["opacities"],
                      scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
    torch.autograd.backward([c, d, a], g)

def run(n, sync_each=False):
    for _ in range(30): step()
    torch.cuda.synchronize(); acc.clear()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
        if sync_each: torch.cuda.synchronize()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

def host_only(n):
    """host time of one fwd+bwd ENQUEUE (the call returns before the GPU has run it): forward call, backward call"""
    for _ in range(30): step()
    torch.cuda.synchronize()
    tf, tb = [], []
    for _ in range(n):
        t0 = time.perf_counter_ns()
        c, r, d, a = rast(means3D=t["means3D"], means2D=m2d, shs=t["shs"], colors_precomp=None, opacities=t["opacities"],
                          scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
        t1 = time.perf_counter_ns()
        torch.autograd.backward([c, d, a], g)
        t2 = time.perf_counter_ns()
        tf.append((t1 - t0) / 1e3); tb.append((t2 - t1) / 1e3)
        if len(tf) % 8 == 0: torch.cuda.synchronize()       # (keep the queue short: the host must not be throttled by a full one)
    return st.median(tf), st.median(tb)

# round 6: the two bindings side by side (the C++ autograd function of csrc/gsr_torch.cpp is the default when built)
for name, on in (("C++ binding (_gsr_torch.so)", True), ("ctypes / Python autograd.Function", False)):
    if on and not D.binding_loaded():
        print("== C++ binding: not built"); continue
    D.use_cpp_binding(on)
    f_us, b_us = host_only(400)
    ms = run(400)
    print(f"== {name}: {ms:.3f} ms/step free running; host per call: forward {f_us:.1f} us, backward {b_us:.1f} us")
    if not on:
        for k, v in sorted(acc.items()):
            v = sorted(v)
            print(f"   {k:28s} median {st.median(v):7.1f} us  p10 {v[len(v)//10]:7.1f}  p90 {v[9*len(v)//10]:7.1f}")
D.use_cpp_binding(True)
