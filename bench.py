#!/usr/bin/env python
"""bench.py -- Mrays/s (fwd+bwd) of the MI355X splat rasterizer on BASELINE.json's workloads.

  python bench.py [--gpus N --steps K --warmup W] [--workload 1M-800-sh3] [--kind blob] [--step render|sds]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
  (`--gpus N` without a launcher's RANK/WORLD_SIZE in the environment spawns the N ranks itself through
  torch.distributed.run on 127.0.0.1, and fails loudly when the box has fewer than N GPUs.)

A step = one forward + one backward of the rasterizer (through the drop-in
`GaussianRasterizer` autograd surface, i.e. the path gs_renderer.py:800-809 / main.py:273
take) for ONE camera over the synthetic scene, inputs and dL/d{color,depth,alpha} already
resident in HBM. With N>1 every rank renders its own orbit camera of the same (replicated)
scene -- the view-parallel mode of SURVEY 8(e) -- and rank 0 gathers the N images with RCCL
inside the timed region; per-GPU work is fixed ("weak" scaling), value = total rays / time.
`--step sds` times the whole exchange of DreamGaussian's multi-view SDS step (main.py:219-275) on
BASELINE configs[3] (250k Gaussians, 512x512, one view per GPU): render, RCCL gather of the images to
rank 0, scatter of dL/dimage back, local backward, bucketed all-reduce of the per-Gaussian gradients.
With N>1 the default run appends that measurement to the same JSON line ("sds_step").

One JSON line on stdout (rank 0). Besides the driver's contract it carries
  roofline      the dominant kernel's algorithmic bytes / its hipEvent-measured duration
  path_roofline the same for the whole fwd+bwd (B_alg of SURVEY 8(d) / step time)
  cpu_baseline  the CPU oracle (oracle/gs_oracle.py, a port: the reference has no CPU render
                path and its CUDA ext is absent) timed on a bounded sample on the host cores
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level table)

WORKLOADS = {                  # BASELINE.json configs (index): N, sh degree, W, H
    "5k-256-sh0": dict(cfg=0, N=5_000, deg=0, W=256, H=256),
    "100k-800-sh3": dict(cfg=1, N=100_000, deg=3, W=800, H=800),
    "1M-800-sh3": dict(cfg=2, N=1_000_000, deg=3, W=800, H=800),
    "250k-512-sh0": dict(cfg=3, N=250_000, deg=0, W=512, H=512),
}


def alg_bytes(N, K, V, M, P, grads_written_by="preprocess_bwd"):
    """Algorithmic (compulsory) HBM bytes of SURVEY 8(d), split per kernel so that the parts
    add up to B_alg(fwd+bwd) = N(3 A_in + 16) + 212 V + 24 M + 56 P.
    `grads_written_by`: the kernel that stores the gradient arrays, N (A_in + 12) bytes of the total. Round 5: for large scenes the
    per-Gaussian backward only visits the Gaussians that carry a gradient (gsr_preprocess_bwd_compact) and a compositing kernel -- the
    forward's, through GsrView.grad_clear, or the backward's -- stores the zeros of all the others on the side: the compulsory write
    of every gradient element is then that kernel's (main() prints both attributions)."""
    A_in = 44 + 12 * K
    per = {
        "preprocess_fwd": N * (A_in + 4) + 44 * V,      # inputs read, radii + 44 B state written
        "scatter": 8 * M,                               # (depth,id) entries written
        "tile_sort": 8 * M,                             # ... read back by the sort
        "render_fwd": 44 * V + 28 * P,                  # state read, 20 B out + 8 B aux per pixel
        "render_bwd": 84 * V + 8 * M + 28 * P,          # state read, 40 B 2D grads, lists, pixel grads
        "preprocess_bwd": N * A_in + 40 * V,            # inputs re-read, 2D grads read
    }
    per[grads_written_by] += N * (A_in + 12)            # ... and every gradient element written once
    per["total"] = sum(per.values())
    assert per["total"] == N * (3 * A_in + 16) + 212 * V + 24 * M + 56 * P
    return per


def kernel_family(name: str) -> str:
    """Kernels that share one line of alg_bytes: the sort's size classes; the forward compositing's two kernels
    (gsr_render_fwd_seg composites the depth segments, gsr_render_fwd_combine chains them per pixel, gsr_render_fwd_fix
    walks the one segment in which a pixel stops)."""
    if name.startswith("tile_sort"):
        return "tile_sort"
    if name == "preprocess_bwd_live":                # gsr_preprocess_bwd_compact: the per-Gaussian backward over the live Gaussians only
        return "preprocess_bwd"
    return "render_fwd" if name in ("render_combine", "render_fix") else name


def apply_order(sc, order):
    """`--order morton`: the scene's rows permuted along a Z-order curve (dreamgaussian_amd.morton_order) -- the same
    Gaussians in another order. Used by BOTH step kinds (round 3 applied it in run_sds only and still stamped
    "order": "morton" into the render line: those lines measured the given order)."""
    if order != "morton":
        return sc
    import dreamgaussian_amd as D
    perm = D.morton_order(sc["means3D"]).long()
    return {k: v[perm].contiguous() for k, v in sc.items()}


def build_inputs(wl, kind, dev, azimuth, order="given"):
    from dreamgaussian_amd import synthetic as syn
    import dreamgaussian_amd as D
    sc = apply_order(syn.make_scene(wl["N"], wl["deg"], 0, kind), order)
    rs_cpu = syn.make_settings(syn.orbit_pose(0.0, azimuth, 2.0), wl["W"], wl["H"], sh_degree=wl["deg"])
    rs = D.GaussianRasterizationSettings(*[x.to(dev) if torch.is_tensor(x) else x for x in rs_cpu])
    g = torch.Generator().manual_seed(1)
    H, W = wl["H"], wl["W"]
    grads = [torch.rand(3, H, W, generator=g), torch.rand(1, H, W, generator=g), torch.rand(1, H, W, generator=g)]
    return sc, rs_cpu, rs, grads


def _cpu_worker(job, threads=1):
    """One process of the CPU baseline: compositing fwd+bwd of a strided subset of the non-empty tiles,
    one torch thread (the oracle's per-tile Python loop does not scale across threads)."""
    wl, kind, azimuth, tiles, seed_w = job
    import torch as _t
    _t.set_num_threads(threads)
    from oracle import gs_oracle as O
    from dreamgaussian_amd import synthetic as syn
    sc = syn.make_scene(wl["N"], wl["deg"], 0, kind)
    rs_cpu = syn.make_settings(syn.orbit_pose(0.0, azimuth, 2.0), wl["W"], wl["H"], sh_degree=wl["deg"])
    g = _t.Generator().manual_seed(seed_w)
    H, W = wl["H"], wl["W"]
    grads = [_t.rand(3, H, W, generator=g), _t.rand(1, H, W, generator=g), _t.rand(1, H, W, generator=g)]
    S = O.Settings(*rs_cpu)
    t = {k: v.detach().clone().requires_grad_(True) for k, v in sc.items()}
    O.TIMERS["composite_fwd"] = O.TIMERS["composite_bwd"] = 0.0
    t0 = time.perf_counter()
    c, r, d, a, aux = O.rasterize(t["means3D"], None, t["opacities"], S, shs=t["shs"], scales=t["scales"],
                                  rotations=t["rotations"], return_aux=True, tiles=tiles)
    _t.autograd.backward([c, d, a], grads)
    total = time.perf_counter() - t0
    comp = O.TIMERS["composite_fwd"] + O.TIMERS["composite_bwd"]
    ranges = aux["ranges"]
    return total - comp, comp, (ranges[1:] - ranges[:-1]).tolist()


def cpu_baseline(wl, kind, azimuth, budget_s=20.0, procs=None):
    """CPU oracle (kind 'port': the reference has no CPU path and its CUDA extension is absent, SURVEY
    0.1/0.4) on the SAME scene/camera/loss. The per-Gaussian stage + binning (fwd+bwd) is timed in this
    process on <= 16 torch threads (torch CPU ops stop scaling there); compositing fwd+bwd runs in `procs`
    single-threaded worker processes (the oracle's per-tile loop), each on a strided share of a tile sample
    sized to ~budget_s per worker. Frame time = t(per-Gaussian stage) + slowest worker's composite time
    scaled to all instance-pixel pairs; cores = the larger of the two process/thread counts."""
    import multiprocessing as mp
    ncpu = os.cpu_count() or 1
    procs = procs or max(1, min(32, ncpu // 2))
    nthreads = max(1, min(16, ncpu))
    torch.set_num_threads(nthreads)
    t_pg, _, cnt = _cpu_worker((wl, kind, azimuth, [], 1), threads=nthreads)
    nonempty = [t for t, c in enumerate(cnt) if c > 0]
    H, W = wl["H"], wl["W"]
    if not nonempty:
        return dict(value=H * W / t_pg / 1e6, unit="Mrays/s", cores=nthreads, kind="port", sample="empty scene")
    total_pairs = float(sum(cnt)) * 256.0
    ctx = mp.get_context("spawn")
    with ctx.Pool(procs) as pool:
        probe = [nonempty[len(nonempty) // 3], nonempty[(2 * len(nonempty)) // 3]]
        _, c_probe, _ = pool.apply(_cpu_worker, ((wl, kind, azimuth, probe, 1),))
        per_pair = max(c_probe, 1e-4) / (float(sum(cnt[t] for t in probe)) * 256.0)
        stride = max(1, int(round(total_pairs * per_pair / (budget_s * procs))))
        sample = nonempty[::stride]
        jobs = [(wl, kind, azimuth, sample[w::procs], 1) for w in range(procs) if sample[w::procs]]
        res = pool.map(_cpu_worker, jobs)
    t_comp = max(r[1] for r in res)
    pairs_s = float(sum(cnt[t] for t in sample)) * 256.0
    t_full = t_pg + t_comp * total_pairs / pairs_s
    return dict(value=H * W / t_full / 1e6, unit="Mrays/s", cores=max(len(jobs), nthreads), kind="port",
                sample=(f"oracle fwd+bwd, same scene/camera/loss; the box has {ncpu} host cores, `cores` = the most that were busy at once: per-Gaussian stage + binning "
                        f"in full on {nthreads} torch threads ({t_pg:.1f} s); compositing in {len(jobs)} single-threaded "
                        f"processes on {len(sample)} of {len(nonempty)} non-empty tiles ({100.0 * pairs_s / total_pairs:.1f}% "
                        f"of instance-pixel pairs, slowest worker {t_comp:.1f} s); frame time scaled by pair count "
                        f"to {t_full:.1f} s"))


def naive_gpu(wl, kind, azimuth, dev, reps=2):
    """BASELINE.md section 3's "naive GPU" middle point: the SAME stock-PyTorch program that serves as the CPU baseline (the oracle:
    torch ops per Gaussian, numpy binning on the host, a Python loop over the tiles with torch ops and torch's autograd), executed on
    the MI355X with PyTorch-ROCm's own kernels. Whole frames, fwd+bwd, 1 warm-up + `reps` timed. A baseline, never a checker."""
    from oracle import gs_oracle as O
    from dreamgaussian_amd import synthetic as syn
    O.ALLOW_DEVICE[0] = True
    try:
        sc = syn.make_scene(wl["N"], wl["deg"], 0, kind)
        rs = syn.make_settings(syn.orbit_pose(0.0, azimuth, 2.0), wl["W"], wl["H"], sh_degree=wl["deg"])
        S = O.Settings(*[x.to(dev) if torch.is_tensor(x) else x for x in rs])
        g = torch.Generator().manual_seed(1)
        H, W = wl["H"], wl["W"]
        grads = [torch.rand(3, H, W, generator=g).to(dev), torch.rand(1, H, W, generator=g).to(dev), torch.rand(1, H, W, generator=g).to(dev)]
        times = []
        for it in range(1 + reps):
            t = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            c, r, d, a = O.rasterize(t["means3D"], None, t["opacities"], S, shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
            torch.autograd.backward([c, d, a], grads)
            torch.cuda.synchronize()
            if it:
                times.append(time.perf_counter() - t0)
        dt = sorted(times)[len(times) // 2]
        return dict(value=float(f"{H * W / dt / 1e6:.4g}"), unit="Mrays/s", ms_per_frame=round(dt * 1e3, 1), kind="torch oracle on the MI355X",
                    sample=f"whole frames fwd+bwd, same scene/camera/loss, stock PyTorch-ROCm kernels + torch autograd, binning in numpy on the host; median of {reps}")
    finally:
        O.ALLOW_DEVICE[0] = False


def spawn_ranks(n):
    """`--gpus n` without a launcher: run torch.distributed.run ourselves (one process per GPU, 127.0.0.1)."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        raise SystemExit(f"bench.py --gpus {n}: this box has {have} GPU(s); refusing to measure fewer ranks than asked")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def run_sds(a, dev, rank, world):
    """The multi-view SDS exchange (main.py:219-275) on BASELINE configs[3]: 250k Gaussians, SH degree 0,
    512x512, ONE orbit view per GPU. Timed region of a step, on every rank: forward of the own view, RCCL
    gather of the [5,H,W] images to rank 0, the image-space loss gradient on rank 0 (an elementwise surrogate of
    the guidance: the real one is a diffusion UNet and is out of scope), scatter of dL/dimage, local backward,
    bucketed all-reduce of the per-Gaussian gradients (every rank then holds the summed gradient and can take
    the identical Adam step: no parameter broadcast). Returns the result dict (rank 0) or None."""
    import torch.distributed as dist
    import dreamgaussian_amd as D
    from dreamgaussian_amd import views
    wl = WORKLOADS[getattr(a, "sds_workload", None) or "250k-512-sh0"]
    azimuth = 360.0 * rank / max(world, 1)
    sc, _, rs, _ = build_inputs(wl, a.kind, dev, azimuth, a.order)
    t = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    m2d = torch.zeros(wl["N"], 3, device=dev, requires_grad=True)
    rast = D.GaussianRasterizer(raster_settings=rs)
    H, W = wl["H"], wl["W"]
    wimg = torch.rand(world, 5, H, W, generator=torch.Generator().manual_seed(7)).to(dev) if rank == 0 else None
    params = list(t.values())
    # --reduce sharded: reduce-scatter of the gradient span, the Adam step on this rank's 1/N slice, all-gather of the parameters
    # (views.ShardedAdam; lr = 0: the scene stays the benchmark's); --reduce live: only the rows some rank has a gradient for
    # (views.allreduce_live_rows); --sds-adam: the dense / live modes also take the (full) Adam step inside the timed region
    reduce_mode = getattr(a, "reduce", "dense")
    opt = sharded = None
    if reduce_mode == "sharded" or getattr(a, "sds_adam", False):
        opt = D.FusedAdam([{"params": [p], "lr": 0.0} for p in params], lr=0.0, eps=1e-15)
        if reduce_mode == "sharded":
            sharded = views.ShardedAdam(opt)
    live_stat = [0, 0]

    wmine = torch.rand(world, 5, H, W, generator=torch.Generator().manual_seed(7))[rank:rank + 1].to(dev)   # (the same weights, this rank's view)
    pending = [None]

    def step():
        if pending[0] is not None:                       # the previous step's all-reduce: its gradients are released right below
            pending[0].wait()
            pending[0] = None
        for v in params:
            v.grad = None
        m2d.grad = None
        color, radii, depth, alpha = rast(means3D=t["means3D"], means2D=m2d, shs=t["shs"], colors_precomp=None,
                                          opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"],
                                          cov3D_precomp=None)
        local = torch.cat([color, depth, alpha], 0).unsqueeze(0)              # [1,5,H,W], differentiable
        if a.sds_mode == "gather":
            batch = views.gather_images(local.detach(), dst=0, num_views=world)
            gw = (batch - 0.5) * wimg if rank == 0 else None                 # d(surrogate loss)/d(images) on rank 0
            g_local = views.scatter_view_grads(gw, local, src=0, num_views=world)
            torch.autograd.backward([local], [g_local])
            views.allreduce_grads(params)
        else:                                            # "local": the same loss, every rank differentiating its own view's term
            torch.autograd.backward([local], [(local.detach() - 0.5) * wmine])
            if sharded is not None:
                sharded.step()
            elif reduce_mode == "live":
                live_stat[0], live_stat[1] = views.allreduce_live_rows(params, probe=[0, 2])     # positions and opacities decide
                if opt is not None:
                    opt.step()
            else:
                pending[0] = views.allreduce_grads(params, async_op=True)
                if opt is not None:
                    pending[0] and pending[0].wait(); pending[0] = None
                    opt.step()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def drain():
        if pending[0] is not None:
            pending[0].wait()
            pending[0] = None

    for _ in range(a.warmup):
        step()
    drain()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain()                                              # the last step's all-reduce belongs to the timed region
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        td = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(td, op=dist.ReduceOp.MAX)
        dt = float(td.item())
    if rank != 0:
        return None
    st = D.last_stats()
    grad_bytes = sum(p.numel() * 4 for p in params)
    return {"workload": f"BASELINE.json configs[3]: {wl['N']} Gaussians, SH degree {wl['deg']}, {W}x{H}, one orbit view per "
                        f"GPU ({world} views), scene '{a.kind}'",
            "timed": ("fwd + RCCL gather(images) + loss grad on rank 0 + scatter(dL/dimage) + bwd + all-reduce(grads)" if a.sds_mode == "gather" else
                      {"dense": "fwd + loss grad of the own view on every rank + bwd + all-reduce(grads), asynchronous: waited for at the top of the next step",
                       "sharded": "fwd + loss grad of the own view + bwd + reduce-scatter(grads) + Adam on the own 1/N slice + all-gather(parameters)",
                       "live": "fwd + loss grad of the own view + bwd + all-reduce(live flags, N bytes) + all-reduce(the union's rows)"}[reduce_mode]
                      + (" + Adam step (all rows, every rank)" if opt is not None and sharded is None else "")),
            "sds_mode": a.sds_mode, "reduce": reduce_mode if a.sds_mode == "local" else "dense",
            **({"live_rows": live_stat[0], "rows": live_stat[1]} if reduce_mode == "live" else {}),
            "value": round(H * W * world * a.steps / dt / 1e6, 3), "unit": "Mrays/s", "ms_per_step": round(dt / a.steps * 1e3, 4),
            "views_per_step": world, "image_bytes_per_view": 5 * H * W * 4, "allreduce_bytes": grad_bytes,
            "M": st.get("M_ref"), "M_emitted": st.get("M")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="1M-800-sh3", choices=sorted(WORKLOADS))
    ap.add_argument("--kind", default="blob", choices=["blob", "trained"])
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU-oracle compositing per worker; 0 disables")
    ap.add_argument("--step", default="render", choices=["render", "sds"],
                    help="render = rasterizer fwd+bwd (+ RCCL gather of the images when N>1): the headline metric; "
                         "sds = the multi-view SDS exchange on BASELINE configs[3] (see the module docstring)")
    ap.add_argument("--sds-mode", default="gather", choices=["local", "gather"],
                    help="--step sds: where the image-space loss is evaluated (dreamgaussian_amd/views.py): local = every rank differentiates its "
                         "own view's term, the all-reduce of the parameter gradients is the only collective; gather = rank 0 evaluates it "
                         "for all views (gather of the images, scatter of dL/dimage, all-reduce)")
    ap.add_argument("--reduce", default="dense", choices=["dense", "sharded", "live"],
                    help="--step sds --sds-mode local: how the parameter gradients meet (dreamgaussian_amd/views.py): dense = one all-reduce over "
                         "their span; sharded = reduce-scatter + Adam on the own slice + all-gather of the parameters (ShardedAdam); live = "
                         "all-reduce of the rows that carry a gradient on some rank only (allreduce_live_rows)")
    ap.add_argument("--sds-adam", action="store_true", help="--step sds: the dense / live modes also take the Adam step inside the timed region")
    ap.add_argument("--sds-workload", default=None, choices=sorted(WORKLOADS), help="--step sds on another scene than BASELINE configs[3] (e.g. 1M-800-sh3: 62 MB of gradients)")
    ap.add_argument("--order", default="given", choices=["given", "morton"],
                    help="given = the Gaussians in the order the scene generator made them (random in space: the reference's init and "
                         "the headline metric); morton = rows permuted along a Z-order curve first (dreamgaussian_amd.reorder_gaussians)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--naive-gpu", action="store_true",
                    help="also time the torch oracle ON the MI355X (BASELINE.md section 3's \"naive GPU\" middle point; whole frames: "
                         "sensible up to 100k Gaussians) and add it to the JSON line as `naive_gpu`")
    ap.add_argument("--force-collectives", action="store_true",
                    help="with --gpus 1: initialise a ONE-rank nccl group and run every collective of the step through RCCL "
                         "(views.force_collectives) instead of short-circuiting them")
    ap.add_argument("--trace-steps", action="store_true", help="print every timed step's wall time to stderr")
    ap.add_argument("--views", type=int, default=1,
                    help="cameras per step on each GPU (B > 1: dreamgaussian_amd.rasterize_views keeps them in flight "
                         "together; --views-serial renders them one after the other like the reference's loop)")
    ap.add_argument("--views-serial", action="store_true")
    ap.add_argument("--activations", default="none", choices=["none", "torch", "fused"],
                    help="what the timed step does about DreamGaussian's parameter activations (gs_renderer.py:134-142): "
                         "none = the rasterizer alone on activated inputs (the headline metric); torch = sigmoid/exp/normalize "
                         "as torch ops + their autograd, as Renderer.render does; fused = rasterize_gaussians_raw")
    ap.add_argument("--binding", default="auto", choices=["auto", "cpp", "ctypes"],
                    help="which torch binding drives the C ABI: the C++ autograd function of csrc/gsr_torch.cpp (auto: when it has been built) or the "
                         "ctypes / Python autograd.Function of rasterizer.py; recorded in the line's config")
    ap.add_argument("--async-forward", action="store_true",
                    help="dreamgaussian_amd.set_async_forward(True): gsr_forward returns without waiting for its instance counters (GSR_VIEW_ASYNC_STATS, "
                         "opt-in; recorded in the line's config -- a line with it is not the headline)")
    ap.add_argument("--hook", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B measurements only: a test hook of the library (dreamgaussian_amd._testing), e.g. fwd_lds_kb=44; "
                         "recorded in the line's config (a line with hooks is not the headline)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    if a.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(a.gpus)                      # does not return
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    elif a.force_collectives:
        import socket
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port_ = s_.getsockname()[1]; s_.close()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(port_))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from dreamgaussian_amd import views as _v
        _v.force_collectives(True)

    import dreamgaussian_amd as D
    from dreamgaussian_amd import _lib, views
    if a.binding == "ctypes":
        D.use_cpp_binding(False)
    elif a.binding == "cpp" and not D.binding_loaded():
        raise SystemExit("--binding cpp: dreamgaussian_amd/_gsr_torch.so has not been built (python -m dreamgaussian_amd.build)")
    if a.async_forward:
        D.set_async_forward(True)
    if a.hook:
        from dreamgaussian_amd import _testing
        for h in a.hook:
            name, _, val = h.partition("=")
            _testing.set(name, int(val))

    if a.step == "sds":
        res = run_sds(a, dev, rank, world)
        if rank == 0:
            out = {"metric": "Mrays/s (multi-view SDS step, fwd+bwd+exchange)", "value": res["value"], "unit": "Mrays/s",
                   "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": res["ms_per_step"],
                   "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                   "config": {"workload": res["workload"], "parallelism": f"view-parallel x{world}" if world > 1 else ("single GPU, collectives through a 1-rank RCCL group" if a.force_collectives else "single GPU"),
                              "timed": res["timed"], "sds_mode": res["sds_mode"], "reduce": res.get("reduce"), "live_rows": res.get("live_rows"), "allreduce_bytes": res["allreduce_bytes"],
                              "image_bytes_per_view": res["image_bytes_per_view"], "M": res["M"], "M_emitted": res["M_emitted"]},
                   "roofline": None, "cpu_baseline": None}
            print(json.dumps(out), flush=True)
        if world > 1 or a.force_collectives:
            dist.destroy_process_group()
        return
    wl = WORKLOADS[a.workload]
    K = (wl["deg"] + 1) ** 2
    azimuth = 360.0 * rank / max(world, 1)          # rank r renders orbit view r
    sc, rs_cpu, rs, grads_cpu = build_inputs(wl, a.kind, dev, azimuth, a.order)
    t = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
    m2d = torch.zeros(wl["N"], 3, device=dev, requires_grad=True)
    gout = [g.to(dev) for g in grads_cpu]
    rast = D.GaussianRasterizer(raster_settings=rs)
    gather_buf = views.make_gather_buffer(world, 5, wl["H"], wl["W"], dev) if world > 1 else None

    if a.views > 1:
        from dreamgaussian_amd import synthetic as syn
        vs = [D.GaussianRasterizationSettings(*[x.to(dev) if torch.is_tensor(x) else x for x in
              syn.make_settings(syn.orbit_pose(0.0, azimuth + 360.0 * i / a.views, 2.0), wl["W"], wl["H"], sh_degree=wl["deg"])])
              for i in range(a.views)]
        m2d_b = torch.zeros(a.views, wl["N"], 3, device=dev, requires_grad=True)
        m2d_l = [torch.zeros(wl["N"], 3, device=dev, requires_grad=True) for _ in range(a.views)]
        rasts = [D.GaussianRasterizer(raster_settings=x) for x in vs]

    gout_views = [g.unsqueeze(0).expand(a.views, *g.shape).contiguous() for g in gout] if a.views > 1 else None

    def step_views():
        for v in t.values():
            v.grad = None
        if a.views_serial:
            for i in range(a.views):
                m2d_l[i].grad = None
                c, r, d, al = rasts[i](means3D=t["means3D"], means2D=m2d_l[i], shs=t["shs"], colors_precomp=None,
                                       opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
                torch.autograd.backward([c, d, al], gout)
        else:
            m2d_b.grad = None
            c, r, d, al = D.rasterize_views(t["means3D"], m2d_b, t["opacities"], vs, shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
            torch.autograd.backward([c, d, al], gout_views)

    if a.activations != "none":      # raw parameters whose activations reproduce the scene
        raw = dict(opacity=torch.logit(sc["opacities"].clamp(1e-4, 1 - 1e-4)), scaling=torch.log(sc["scales"]),
                   rotation=sc["rotations"] * 1.7)
        traw = {k: v.to(dev).requires_grad_(True) for k, v in raw.items()}

    def step_activations():
        for v in list(t.values()) + list(traw.values()):
            v.grad = None
        m2d.grad = None
        if a.activations == "torch":
            out = rast(means3D=t["means3D"], means2D=m2d, shs=t["shs"], colors_precomp=None,
                       opacities=torch.sigmoid(traw["opacity"]), scales=torch.exp(traw["scaling"]),
                       rotations=torch.nn.functional.normalize(traw["rotation"]), cov3D_precomp=None)
        else:
            out = D.rasterize_gaussians_raw(t["means3D"], m2d, t["shs"], traw["opacity"], traw["scaling"], traw["rotation"], rs)
        torch.autograd.backward([out[0], out[2], out[3]], gout)

    def step(gather=True):
        if a.activations != "none":
            return step_activations()
        if a.views > 1:
            return step_views()
        for v in t.values():
            v.grad = None
        m2d.grad = None
        color, radii, depth, alpha = rast(means3D=t["means3D"], means2D=m2d, shs=t["shs"],
                                          colors_precomp=None, opacities=t["opacities"],
                                          scales=t["scales"], rotations=t["rotations"],
                                          cov3D_precomp=None)
        work = None
        if world > 1 and gather:                     # RCCL gather of the rendered views to rank 0
            work = views.gather_views_async(color, depth, alpha, gather_buf, dst=0)
        torch.autograd.backward([color, depth, alpha], gout)
        if work is not None:
            work.wait()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    trace = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        if a.trace_steps:
            torch.cuda.synchronize()
            trace.append(time.perf_counter())
    sync()
    dt = time.perf_counter() - t0
    if a.trace_steps and rank == 0:
        prev = t0
        print("step ms:", " ".join(f"{(t - p) * 1e3:.2f}" for p, t in zip([t0] + trace[:-1], trace)), file=sys.stderr)
    if world > 1:
        td = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(td, op=dist.ReduceOp.MAX)
        dt = float(td.item())
    st = D.last_stats()

    # ---- per-kernel durations: hipEvents around every launch, same workload, K more steps ----
    kern, kern_raw, roof, path_roof, dt_prof = {}, {}, None, None, None
    if not a.no_roofline and rank == 0 and a.views == 1:
        _lib.profile_reset()
        _lib.profile_enable(True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            step(gather=False)                       # rank 0 only: no collective in this pass
        torch.cuda.synchronize()
        dt_prof = time.perf_counter() - t1
        _lib.profile_enable(False)
        raw = _lib.profile_read()
        kern_raw = {name: ms / a.steps for name, (ms, n) in raw.items()}
        fam_launches = {}
        for name, (ms, n) in raw.items():
            fam = kernel_family(name)
            e = kern.setdefault(fam, [0.0, 0])
            e[0] += ms
            e[1] = max(e[1], n)
            fam_launches[fam] = fam_launches.get(fam, 0) + n / a.steps
        P = wl["H"] * wl["W"]
        # who stores the zeros of the gradient arrays: the per-Gaussian backward itself (it streams every Gaussian), or -- its "live
        # Gaussians only" variant ran -- the forward's compositing kernel (GsrStats.bwd_prepared == 2: GsrView.grad_clear) / the
        # backward's. The line's `frac` keeps SURVEY 8(d)'s split (the gradient arrays priced with the per-Gaussian backward) whoever
        # clears them; when the DOMINANT kernel is the one that clears, its figure with those 248 MB counted is printed beside it.
        gw = "preprocess_bwd"
        if "preprocess_bwd_live" in raw:
            gw = "render_fwd" if st.get("bwd_prepared") == 2 else "render_bwd"
        ab = alg_bytes(wl["N"], K, st["V"], st["M_ref"], P)
        ab_gw = alg_bytes(wl["N"], K, st["V"], st["M_ref"], P, gw)
        per_step = {k: v[0] / a.steps for k, v in kern.items()}
        dom = max(per_step, key=per_step.get)
        # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate passes,
        # MI355X_MICROARCH.md HBM section; tools/pmc_summary.py -> profiles/pmc_traffic.json). The file carries
        # the digest of the kernel sources it was collected with: a stale file reads as null, never as a number.
        traffic, traffic_note = None, "not collected"
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf):
            try:
                from dreamgaussian_amd import build as _build
                tj = json.load(open(tf))
                if tj.get("source_digest") == _build.kernel_digest():
                    traffic = tj.get(f"{a.workload}/{a.kind}", {}).get(dom, {}).get("hbm_bytes")
                    traffic_note = tj.get("note", "rocprofv3 PMC, same kernel sources")
                else:
                    traffic_note = "profiles/pmc_traffic.json was collected with other kernel sources"
            except Exception:
                traffic = None
        if dom in ab:
            ach = ab[dom] / (per_step[dom] * 1e-3) / 1e9
            ab_emit = alg_bytes(wl["N"], K, st["V"], st["M"], P)       # priced with the lists the kernels really walk
            roof = dict(bound="hbm", kernel=dom, achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(ach / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=traffic_note,
                        alg_bytes_per_launch=ab[dom], instances_priced="M_ref (reference emission rule, SURVEY 8(d))",
                        alg_bytes_per_launch_emitted=ab_emit[dom],
                        frac_emitted=round(ab_emit[dom] / (per_step[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        gradient_arrays_cleared_by=gw,
                        **({"frac_with_the_gradient_arrays_it_clears": round(ab_gw[dom] / (per_step[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
                           if dom == gw else {}),
                        avg_launch_ms=round(per_step[dom], 4), launches_per_step=round(fam_launches.get(dom, 1)),
                        kernels={k: round(v, 4) for k, v in kern_raw.items() if kernel_family(k) == dom})
        ach_p = ab["total"] / (dt / a.steps) / 1e9
        path_roof = dict(bound="hbm", achieved=round(ach_p, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                         frac=round(ach_p / HBM_PEAK_GBS, 5), alg_bytes_per_step=ab["total"],
                         gpu_kernel_ms_per_step=round(sum(per_step.values()), 4))
    # ---- the same K un-instrumented steps once more, now that W + 2K steps have run: the first ~25 steps of a process run on rising
    # clocks (profiles/r04_warmup_ramp.txt), so the line's `value` (exactly W warm-up + K timed steps, as the contract says) sits on
    # the ramp; this is what a training loop sees from its 25th iteration on. Reported beside `value`, never instead of it.
    dt_after = None
    if rank == 0 and world == 1 and not a.no_roofline:
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        dt_after = time.perf_counter() - t2
    if world > 1:
        dist.barrier()

    sds = sds_local = None
    if world > 1 and a.views == 1 and a.activations == "none":
        for v in list(t.values()) + [m2d]:         # release the headline scene before the second measurement
            v.grad = None
        # both placements of the loss, each under its own name (round-5 advisor: one default that changed between rounds made the
        # rounds' sds_step figures incomparable)
        a.sds_mode = "gather"
        sds = run_sds(a, dev, rank, world)
        a.sds_mode = "local"
        sds_local = run_sds(a, dev, rank, world)

    cpu = None
    if rank == 0 and world == 1 and a.cpu_budget > 0:
        cpu = cpu_baseline(wl, a.kind, azimuth, a.cpu_budget)
        cpu["value"] = float(f"{cpu['value']:.4g}")

    naive = None
    if rank == 0 and world == 1 and a.naive_gpu:
        for v in list(t.values()) + [m2d]:
            v.grad = None
        naive = naive_gpu(wl, a.kind, azimuth, dev)

    if rank == 0:
        rays = wl["H"] * wl["W"] * world * a.steps * a.views
        out = {
            "metric": "Mrays/s (fwd+bwd)", "value": round(rays / dt / 1e6, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[{wl['cfg']}]: {wl['N']} Gaussians, SH degree "
                                   f"{wl['deg']}, {wl['W']}x{wl['H']}, fwd+bwd, scene '{a.kind}' seed 0, "
                                   f"orbit camera r=2 fovy=49.1",
                       "views_per_step": world * a.views, "order": a.order, "activations": a.activations, "views_mode": ("single" if a.views == 1 else ("serial loop" if a.views_serial else "rasterize_views/chain")), "parallelism": f"view-parallel x{world}" if world > 1 else "single GPU",
                       "N": wl["N"], "K": K, "V": st.get("V"), "M": st.get("M_ref"), "M_emitted": st.get("M"),
                       "max_tile_list": st.get("max_tile"), "seg_shift": st.get("seg_shift"),
                       **({"hooks": list(a.hook)} if a.hook else {}), **({"async_forward": True} if a.async_forward else {}),
                       "binding": "cpp" if (a.binding != "ctypes" and D.binding_loaded()) else "ctypes"},
            "roofline": roof, "path_roofline": path_roof, "cpu_baseline": cpu,
            "kernels_ms_per_step": {k: round(v[0] / a.steps, 4) for k, v in sorted(kern.items())},
            "kernels_ms_per_step_raw": {k: round(v, 4) for k, v in sorted(kern_raw.items())} if kern else {},
            "ms_per_step_with_events": None if dt_prof is None else round(dt_prof / a.steps * 1e3, 4),
            "ms_per_step_after_ramp": None if dt_after is None else round(dt_after / a.steps * 1e3, 4),
        }
        if sds is not None:
            out["sds_step"] = sds
            out["sds_step_local"] = sds_local
        if naive is not None:
            out["naive_gpu"] = naive
        print(json.dumps(out), flush=True)
    if views.gather_fallbacks:
        raise SystemExit(f"bench.py: the gather of the rendered views fell back to all_gather {views.gather_fallbacks} times "
                         f"(world x the bytes): the line above does not measure the intended exchange")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
