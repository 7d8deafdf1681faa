"""How far the forward compositing walks each tile's list (CPU, oracle; analysis aid, not a test).

    python tests/walk_stats.py --n 1000000 --size 800 --kind blob

For every non-empty tile and each of its four 8x8 blocks (one wave in K5) the walk ends at the
block's deepest last contributor when every pixel of the block has stopped (T(1-alpha) < 1e-4 fired:
T_final within `--stopped-below` of the threshold), else at the end of the list. Prints the share
of listed (tile, Gaussian) entries that are walked and the longest walks: the numbers behind
DESIGN.md section 7 (is the forward bound by one long walk or by the number of independent walks?)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gs_oracle as O


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--size", type=int, default=800)
    ap.add_argument("--deg", type=int, default=0)
    ap.add_argument("--kind", default="blob")
    ap.add_argument("--stopped-below", type=float, default=None,
                    help="a pixel counts as stopped when T_final < this (default 1.2e-4 for blob: alpha <= 0.1; 2e-3 for trained)")
    a = ap.parse_args()
    thr = a.stopped_below if a.stopped_below is not None else (1.2e-4 if a.kind == "blob" else 2e-3)
    sc = O.make_scene(a.n, a.deg, 0, a.kind)
    S = O.make_settings(O.orbit_pose(0.0, 0.0, 2.0), a.size, a.size, sh_degree=a.deg)
    t0 = time.time()
    with torch.no_grad():
        _, _, _, _, aux = O.rasterize(sc["means3D"], None, sc["opacities"], S, shs=sc["shs"], scales=sc["scales"],
                                      rotations=sc["rotations"], return_aux=True)
    print(f"oracle forward {time.time() - t0:.1f} s, M = {aux['M']}")
    ranges = aux["ranges"]
    nc = aux["n_contrib"].reshape(a.size, a.size)
    Tf = aux["T_final"].reshape(a.size, a.size)
    gx = (a.size + 15) // 16
    listed = walked = 0
    per_tile = []
    for tix in range(gx * gx):
        n = int(ranges[tix + 1] - ranges[tix])
        if n == 0:
            continue
        ty, tx = divmod(tix, gx)
        w_tile = 0
        for by in (0, 8):
            for bx in (0, 8):
                nb = nc[ty * 16 + by:ty * 16 + by + 8, tx * 16 + bx:tx * 16 + bx + 8]
                tb = Tf[ty * 16 + by:ty * 16 + by + 8, tx * 16 + bx:tx * 16 + bx + 8]
                if nb.numel() == 0:
                    continue
                w = n if bool((tb >= thr).any()) else min(n, int(nb.max()) + 1)
                w_tile = max(w_tile, w)
                walked += w
                listed += n
        per_tile.append((n, w_tile))
    pt = np.array(per_tile)
    print(f"stopped pixels (T_final < {thr:g}): {float((Tf < thr).float().mean()):.3f} of the image")
    print(f"non-empty tiles {len(pt)}; wave work items {4 * len(pt)}; listed entries x waves {listed}; walked {walked} "
          f"({walked / listed:.3f} of listed)")
    print(f"longest list {pt[:, 0].max()}, longest walk {pt[:, 1].max()}, mean walk {pt[:, 1].mean():.0f}; "
          f"tiles walking > 2000: {(pt[:, 1] > 2000).sum()}, > 1000: {(pt[:, 1] > 1000).sum()}")


if __name__ == "__main__":
    main()
