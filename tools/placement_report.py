"""Where did the workgroups of the two compositing kernels run, and for how long? Needs the probe build (tools/placement_probe.sh):
    GSR_LIB=$PWD/_exp/libgsr_place.so python tools/placement_report.py [--workload 1M-800-sh3] [--kind blob]
Prints, for gsr_render_fwd_serial (one row per wave) and gsr_render_bwd_q2 (one row per work item): the kernel's span, how the busy
waves spread over XCDs / CUs / SIMDs, when SIMDs finish, and the waves that finish last. 100 MHz wall clock (10 ns steps)."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import dreamgaussian_amd as D
from dreamgaussian_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="1M-800-sh3")
ap.add_argument("--kind", default="blob")
ap.add_argument("--hook", action="append", default=[])
a = ap.parse_args()
if a.hook:
    from dreamgaussian_amd import _testing
    for h in a.hook:
        n_, _, v_ = h.partition("="); _testing.set(n_, int(v_))
dev = torch.device("cuda:0")
wl = bench.WORKLOADS[a.workload]
sc, rs_cpu, rs, grads_cpu = bench.build_inputs(wl, a.kind, dev, 0.0, "given")
t = {k: v.to(dev).requires_grad_(True) for k, v in sc.items()}
m2d = torch.zeros(wl["N"], 3, device=dev, requires_grad=True)
gout = [g.to(dev) for g in grads_cpu]
rast = D.GaussianRasterizer(raster_settings=rs)
for _ in range(12):
    for v in t.values(): v.grad = None
    c, r, d, al = rast(means3D=t["means3D"], means2D=m2d, shs=t["shs"], colors_precomp=None, opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"], cov3D_precomp=None)
    torch.autograd.backward([c, d, al], gout)
torch.cuda.synchronize()
lib = _lib.load()
lib.gsr_debug_placement.restype = C.c_int
lib.gsr_debug_placement.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
st = D.last_stats()
T = ((wl["W"] + 15) // 16) * ((wl["H"] + 15) // 16)


def read(which, rows):
    buf = np.zeros(rows * 4, dtype=np.uint64)
    rc = lib.gsr_debug_placement(which, buf.ctypes.data, buf.size)
    assert rc == 0, rc
    return buf.reshape(rows, 4)


def where(w):
    hw = (w & 0xffffffff).astype(np.int64); xcc = (w >> 32).astype(np.int64) & 15
    return dict(xcc=xcc, se=(hw >> 13) & 7, sh=(hw >> 12) & 1, cu=(hw >> 8) & 15, simd=(hw >> 4) & 3, wave=hw & 15)


def report(name, rows, length, extra, per_simd):
    t0, t1 = rows[:, 0].astype(np.int64), rows[:, 1].astype(np.int64)
    ok = t1 > 0
    rows, t0, t1, length, extra = rows[ok], t0[ok], t1[ok], length[ok], extra[ok]
    base = t0.min()
    s_us, e_us = (t0 - base) / 100.0, (t1 - base) / 100.0
    w = where(rows[:, 2])
    cu_key = ((w["xcc"] * 8 + w["se"]) * 2 + w["sh"]) * 16 + w["cu"]
    key = cu_key * 4 + w["simd"] if per_simd else cu_key
    busy = length > 0
    print(f"== {name}: {len(rows)} rows, {int(busy.sum())} with work; span {e_us.max():.1f} us; busy rows start at median {np.median(s_us[busy]):.1f} us (p90 {np.percentile(s_us[busy], 90):.1f}, max {s_us[busy].max():.1f})")
    dur = e_us - s_us
    print(f"   duration of a busy row: median {np.median(dur[busy]):.1f} us, p90 {np.percentile(dur[busy], 90):.1f}, max {dur[busy].max():.1f}; sum over busy rows {dur[busy].sum() / 1e3:.2f} ms")
    print(f"   distinct XCDs {len(set(w['xcc']))}, CUs {len(set(cu_key))}, {'SIMDs' if per_simd else 'CUs'} used {len(set(key[busy]))}")
    # per unit: number of busy rows, work, finish time
    units = {}
    for k_, b_, L, e_, d_ in zip(key, busy, extra, e_us, dur):
        if not b_: continue
        u = units.setdefault(int(k_), [0, 0, 0.0, 0.0]); u[0] += 1; u[1] += int(L); u[2] = max(u[2], e_); u[3] += d_
    arr = np.array(list(units.values()), dtype=np.float64)
    print(f"   per {'SIMD' if per_simd else 'CU'}: busy rows min / median / max = {arr[:, 0].min():.0f} / {np.median(arr[:, 0]):.0f} / {arr[:, 0].max():.0f}; work (entries walked) min / median / max = {arr[:, 1].min():.0f} / {np.median(arr[:, 1]):.0f} / {arr[:, 1].max():.0f}")
    print(f"   finish time per unit: p10 {np.percentile(arr[:, 2], 10):.1f}  median {np.median(arr[:, 2]):.1f}  p90 {np.percentile(arr[:, 2], 90):.1f}  max {arr[:, 2].max():.1f} us; corr(work, finish) = {np.corrcoef(arr[:, 1], arr[:, 2])[0, 1]:.2f}")
    print(f"   ideal (total work spread evenly, same rate as the slowest unit's): {arr[:, 1].sum() / len(arr) / max(arr[:, 1].max(), 1) * arr[:, 2].max():.1f} us")
    # how many rows are alive over time
    for q in (0.25, 0.5, 0.75, 0.9):
        tt = e_us.max() * q
        alive = int(((s_us <= tt) & (e_us > tt) & busy).sum())
        print(f"   at {q:.0%} of the span ({tt:.0f} us): {alive} busy rows alive")
    last = np.argsort(-e_us)[:8]
    for i in last:
        print(f"   late: row {i} (workgroup {i // 4 if per_simd else i}) list {int(length[i])} walked {int(extra[i])} start {s_us[i]:.1f} end {e_us[i]:.1f} us  xcc {w['xcc'][i]} se {w['se'][i]} cu {w['cu'][i]} simd {w['simd'][i]}")
    # placement of the first rows: which CU did launch-order neighbours get?
    first = [f"{w['xcc'][i]}.{w['se'][i]}.{w['cu'][i]}" for i in range(0, min(len(rows), 4 * 24 if per_simd else 24), 4 if per_simd else 1)]
    print("   xcc.se.cu of the first workgroups in launch order:", " ".join(first))


f = read(0, min(T * 4, 16384))
report("gsr_render_fwd_serial (row = wave)", f, (f[:, 3] & 0xffffffff).astype(np.int64), (f[:, 3] >> 32).astype(np.int64), True)
nb = min(int(st["M"] // 64 + T + 1), 65536)
b = read(1, nb)
report("gsr_render_bwd_q2 (row = work item)", b, (b[:, 3] & 0xffffffff).astype(np.int64), (b[:, 3] & 0xffffffff).astype(np.int64), False)
