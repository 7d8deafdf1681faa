"""HIP-vs-oracle difference statistics for one synthetic scene at full BASELINE size (runs on the
GPU box; the oracle takes minutes there, so this is a report, not a test).
Test infrastructure: uses oracle/ as the checker only.

  python tests/parity_report.py --n 100000 --deg 3 --size 800 [--kind blob] [--f64]
"""
import argparse, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.set_num_threads(min(16, os.cpu_count() or 1))
from oracle import gs_oracle as O
import util


def stats(name, x, ref, mask=None):
    x, ref = x.double(), ref.double()
    err = (x - ref).abs()
    scale = ref.abs().max().item() + 1e-30
    strict = err if mask is None else err.masked_fill(mask, 0.0)
    print(f"  {name:12s} max|err| {err.max().item():.3e}  (outside fragile set {strict.max().item():.3e})  "
          f"max|ref| {scale:.3e}  rel(max, strict) {strict.max().item() / scale:.3e}  "
          f"rel(L2) {(err.norm() / (ref.norm() + 1e-30)).item():.3e}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100000)
    ap.add_argument("--deg", type=int, default=3)
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--kind", default="blob")
    ap.add_argument("--f64", action="store_true")
    a = ap.parse_args()
    sc = O.make_scene(a.n, a.deg, 0, a.kind)
    S = O.make_settings(O.orbit_pose(0, 0, 2.0), a.size, a.size, sh_degree=a.deg)
    w = util.weights_for(a.size, a.size)
    dev = torch.device("cuda:0")
    ho, hg, st = util.run_hip(sc, S, dev, w)
    print(f"scene: {a.n} Gaussians, SH degree {a.deg}, {a.size}x{a.size}, kind {a.kind}; HIP stats {st}")
    t0 = time.time()
    oo, og, aux = util.run_oracle(sc, S, w, torch.float64 if a.f64 else torch.float32)
    print(f"oracle ({'f64' if a.f64 else 'f32'}, {torch.get_num_threads()} threads): {time.time() - t0:.0f} s; M {aux['M']} V {aux['V']}; "
          f"fragile pixels {int(aux['fragile_pixels'].sum())}, fragile Gaussians {int(aux['fragile_gaussians'].sum())}")
    for n, i in (("color", 0), ("alpha", 3)):
        err = (ho[i].double() - oo[i].double()).abs().amax(0)
        bad = err > util.FWD_ATOL
        print(f"  {n}: pixels above {util.FWD_ATOL:g}: {int(bad.sum())} ({int((bad & ~aux['fragile_pixels']).sum())} outside the fragile set)")
    print("radii mismatches:", int((ho[1].long() != oo[1].long()).sum()), "of", a.n)
    fp = aux["fragile_pixels"]
    for n, i in (("color", 0), ("depth", 2), ("alpha", 3)):
        stats(n, ho[i], oo[i], fp[None].expand_as(oo[i]))
    fg = aux["fragile_gaussians"]
    for k in og:
        m = fg.reshape([-1] + [1] * (og[k].dim() - 1)).expand_as(og[k])
        stats("d" + k, hg[k].reshape(og[k].shape), og[k], m)
    try:
        util.assert_forward_close(ho, oo, aux, atol=util.FWD_ATOL if a.f64 else 5e-5)
        util.assert_grads_close(hg, og, aux, rtol=util.GRAD_RTOL if a.f64 else 5e-4, floors=util.grad_floors(sc, og))
        print("PARITY GATE: pass (tests/util.py tolerances%s)" % ("" if a.f64 else ", fp32-oracle widths"))
    except AssertionError as e:
        print("PARITY GATE: FAIL", e)
