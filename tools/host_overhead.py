"""Where does the host time of one fwd+bwd go at DreamGaussian's real sizes (5k Gaussians, 256^2)?
cProfile of 300 steps through the drop-in surface. Runs on the GPU box."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dreamgaussian_amd as D
from dreamgaussian_amd import synthetic as syn

dev = torch.device("cuda:0")
N, W = int(os.environ.get("N", 5000)), int(os.environ.get("W", 256))
sc = syn.make_scene(N, 0, 0, "blob")
rs = syn.make_settings(syn.orbit_pose(0, 0, 2.0), W, W, sh_degree=0, device=dev)
t = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
m2d = torch.zeros(N, 3, device=dev, requires_grad=True)
g = [torch.rand(3, W, W, device=dev), torch.rand(1, W, W, device=dev), torch.rand(1, W, W, device=dev)]
rast = D.GaussianRasterizer(raster_settings=rs)

def step():
    c, r, d, a = rast(means3D=t["means3D"], means2D=m2d, shs=t["shs"], colors_precomp=None, opacities=t["opacities"],
                      scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
    torch.autograd.backward([c, d, a], g)

for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 300 * 1e3:.3f} ms/step wall (N={N}, {W}x{W})")
# the floor of the drop-in surface on this host: an autograd.Function with the rasterizer's inputs and outputs that launches NOTHING
class _Null(torch.autograd.Function):
    @staticmethod
    def forward(ctx, m3, m2, sh, op, sc, rot):
        ctx.save_for_backward(m3, m2, sh, op, sc, rot)
        ctx.set_materialize_grads(False)
        e = torch.empty
        return e(3, W, W, device=dev), e(N, dtype=torch.int32, device=dev), e(1, W, W, device=dev), e(1, W, W, device=dev)

    @staticmethod
    def backward(ctx, gc, gr, gd, ga):
        m3, m2, sh, op, sc, rot = ctx.saved_tensors
        return tuple(torch.empty_like(x) for x in (m3, m2, sh, op, sc, rot))

def null_step():
    c, r, d, a = _Null.apply(t["means3D"], m2d, t["shs"], t["opacities"], t["scales"], t["rotations"])
    torch.autograd.backward([c, d, a], g)

for _ in range(20): null_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): null_step()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / 300 * 1e3:.3f} ms/step floor: an autograd.Function with the same signature that allocates its outputs and launches nothing")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
