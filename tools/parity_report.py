"""Print HIP-vs-oracle difference statistics for one synthetic scene (runs on the GPU box).
Test infrastructure: uses oracle/ as the checker only."""
import argparse, sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import gs_oracle as O
import dreamgaussian_amd as D


def run_hip(sc, S, dev, weights=None):
    t = {k: v.detach().to(dev).requires_grad_(True) for k, v in sc.items()}
    N = t["means3D"].shape[0]
    m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
    rs = D.GaussianRasterizationSettings(S.image_height, S.image_width, S.tanfovx, S.tanfovy,
                                         S.bg.to(dev), S.scale_modifier, S.viewmatrix.to(dev),
                                         S.projmatrix.to(dev), S.sh_degree, S.campos.to(dev), False, False)
    out = D.GaussianRasterizer(raster_settings=rs)(
        means3D=t["means3D"], means2D=m2d, shs=t.get("shs"), colors_precomp=t.get("colors_precomp"),
        opacities=t["opacities"], scales=t.get("scales"), rotations=t.get("rotations"),
        cov3D_precomp=t.get("cov3D_precomp"))
    grads = None
    if weights is not None:
        wc, wd, wa = [w.to(dev) for w in weights]
        loss = (wc * out[0]).sum() + (wd * out[2]).sum() + (wa * out[3]).sum()
        loss.backward()
        grads = {k: v.grad.detach().cpu() for k, v in t.items()}
        grads["means2D"] = m2d.grad.detach().cpu()
    return [o.detach().cpu() for o in out], grads, D.last_stats()


def run_oracle(sc, S, weights=None, dtype=torch.float32):
    t = {k: v.detach().to(dtype).requires_grad_(True) for k, v in sc.items()}
    N = t["means3D"].shape[0]
    m2d = torch.zeros(N, 3, dtype=dtype, requires_grad=True)
    S2 = S._replace(bg=S.bg.to(dtype), viewmatrix=S.viewmatrix.to(dtype), projmatrix=S.projmatrix.to(dtype),
                    campos=S.campos.to(dtype))
    c, r, d, a, aux = O.rasterize(t["means3D"], m2d, t["opacities"], S2, shs=t.get("shs"),
                                  colors_precomp=t.get("colors_precomp"), scales=t.get("scales"),
                                  rotations=t.get("rotations"), cov3D_precomp=t.get("cov3D_precomp"),
                                  return_aux=True)
    grads = None
    if weights is not None:
        wc, wd, wa = [w.to(dtype) for w in weights]
        loss = (wc * c).sum() + (wd * d).sum() + (wa * a).sum()
        loss.backward()
        grads = {k: v.grad.detach() for k, v in t.items()}
        grads["means2D"] = m2d.grad.detach()
    return [c.detach(), r, d.detach(), a.detach()], grads, aux


def stats(name, x, ref):
    x, ref = x.double(), ref.double()
    err = (x - ref).abs()
    scale = ref.abs().max().item() + 1e-30
    q = torch.quantile(err.flatten()[:: max(1, err.numel() // 1000000)], 0.999).item()
    print(f"  {name:12s} max|err| {err.max().item():.3e}  p99.9 {q:.3e}  max|ref| {scale:.3e}  "
          f"rel(max) {err.max().item() / scale:.3e}  rel(L2) {(err.norm() / (ref.norm() + 1e-30)).item():.3e}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=5000)
    ap.add_argument("--deg", type=int, default=0)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--kind", default="blob")
    ap.add_argument("--f64", action="store_true")
    a = ap.parse_args()
    sc = O.make_scene(a.n, a.deg, 0, a.kind)
    S = O.make_settings(O.orbit_pose(0, 0, 2.0), a.size, a.size, sh_degree=a.deg)
    g = torch.Generator().manual_seed(1)
    w = (torch.rand(3, a.size, a.size, generator=g), torch.rand(1, a.size, a.size, generator=g),
         torch.rand(1, a.size, a.size, generator=g))
    dev = torch.device("cuda:0")
    t0 = time.time(); ho, hg, st = run_hip(sc, S, dev, w); torch.cuda.synchronize(); t1 = time.time()
    print("hip stats", st, f"{t1 - t0:.3f}s")
    oo, og, aux = run_oracle(sc, S, w, torch.float64 if a.f64 else torch.float32)
    print("oracle M", aux["M"], "V", aux["V"], f"{time.time() - t1:.1f}s")
    print("radii equal:", bool((ho[1] == oo[1]).all()), "mismatch", int((ho[1] != oo[1]).sum()))
    for n, i in (("color", 0), ("depth", 2), ("alpha", 3)):
        stats(n, ho[i], oo[i])
    for k in og:
        stats("d" + k, hg[k], og[k])
