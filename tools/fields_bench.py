"""Time extract_fields (C ABI gsr_extract_fields) on the MI355X and price it against the vector-ALU
roofline that bounds it. Prints one JSON line per size.

    python tools/fields_bench.py [--sizes 100000,1000000] [--resolution 128] [--reps 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dreamgaussian_amd as D
from dreamgaussian_amd import _lib, synthetic
from dreamgaussian_amd.fields import block_geometry

# ALGORITHMIC fp32 operations per (grid point, Gaussian) pair, as the reference's formula spends them
# (gs_renderer.py:79-83, 281): 3 sub + 6 products + 6 x inverse + 1 x -0.5 + 5 add/sub + compare/select
# + 1 x opacity + 1 add = 24, plus one exp. The kernel shares the (x, y) terms of four z-neighbours, so it
# executes fewer; the roofline prices the algorithmic count.
OPS_PER_PAIR = 25
# vector-ALU peak WITHOUT fma credit (the reference's one-rounding-per-operation arithmetic forbids
# contraction) but with packed fp32: 256 CUs x 4 SIMDs x 16 lanes x 2 (v_pk_*) x 2.4 GHz = 78.6 Tops/s
# (= half of the 157.3 TFLOP/s vector fp32 figure of MI355X_MICROARCH.md, which counts an fma as two).
VALU_PEAK_OPS = 256 * 4 * 16 * 2 * 2.4e9


def pair_count(xyz, op, scale_center, R, nb, relax, dev):
    """sum over blocks of (grid points of the block) x (Gaussians of the block), from the same membership rule."""
    axis, split, lo, hi = block_geometry(R, nb, relax)
    center, scale = scale_center
    keep = op.reshape(-1) > 0.005
    p = (xyz[keep] - center) * scale
    lo, hi = lo.to(dev), hi.to(dev)
    m = [((p[:, a:a + 1] > lo[None]) & (p[:, a:a + 1] < hi[None])).float() for a in range(3)]
    nc = lo.shape[0]
    xy = (m[0][:, :, None] * m[1][:, None, :]).reshape(-1, nc * nc)
    cnt = (xy.t() @ m[2]).reshape(nc, nc, nc)                        # Gaussians per block
    lens = torch.tensor([min(split, R - i * split) for i in range(nc)], device=dev, dtype=torch.float32)
    pts = lens[:, None, None] * lens[None, :, None] * lens[None, None, :]
    return float((cnt.double() * pts.double()).sum().item()), float(cnt.max().item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="100000,1000000")
    ap.add_argument("--resolution", type=int, default=128)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for N in [int(s) for s in a.sizes.split(",")]:
        sc = synthetic.make_scene(N, 0, 0, "trained")
        t = [sc[k].to(dev) for k in ("means3D", "opacities", "scales", "rotations")]
        occ, center, scale = D.extract_fields(*t, resolution=a.resolution)      # warm-up
        torch.cuda.synchronize()
        _lib.profile_reset(); _lib.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(a.reps):
            occ, center, scale = D.extract_fields(*t, resolution=a.resolution)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.reps
        _lib.profile_enable(False)
        rows = {n: ms / max(c, 1) for n, (ms, c) in _lib.profile_read().items()}
        pairs, lmax = pair_count(t[0], t[1], (center, np.float32(scale)), a.resolution, 16, 1.5, dev)
        acc_ms = rows.get("fields_accumulate", float("nan"))
        ops = pairs * OPS_PER_PAIR
        print(json.dumps({
            "op": "extract_fields", "N": N, "resolution": a.resolution, "ms_wall": round(wall * 1e3, 3),
            "kernels_ms": {k: round(v, 4) for k, v in rows.items() if k.startswith("fields")},
            "pairs": pairs, "max_gaussians_per_block": lmax,
            "roofline": {"bound": "valu", "achieved": round(ops / (acc_ms * 1e-3) / 1e12, 3), "peak": round(VALU_PEAK_OPS / 1e12, 2),
                         "unit": "Tops/s fp32", "frac": round(ops / (acc_ms * 1e-3) / VALU_PEAK_OPS, 3)},
            "Gpairs_per_s": round(pairs / (acc_ms * 1e-3) / 1e9, 2), "occ_max": round(float(occ.max()), 4)}))


if __name__ == "__main__":
    main()
